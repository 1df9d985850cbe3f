"""GPU parity: the HIP path (through the C ABI) vs the committed golden fixtures of the reference and
vs the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): tokeniser tensors bit-exact after the f64->f32 cast, sampled token
descriptors <= 1e-6, line descriptors <= 1e-4 (fp32), match matrices identical by index."""
import numpy as np
import pytest
import torch

from helpers import BASE_CFG, TOK_KEYS, golden_cfg, load, oracle_image, tiny_maps, weights_for
from workloads import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

DESC_TOL = 1e-4     # north_star tolerance on unit-norm fp32 descriptors
TOKDESC_TOL = 1e-6


@pytest.fixture(scope="module")
def engine():
    from linetr_amd.engine import Engine
    return Engine(synth.calibrated_state_dict(), "cuda:0")


def run_native(eng, rows_list, dd, ds, hw, cfg, align_corners=False, valid_masks=None):
    recs, cu_k, cu_n = eng.prefilter(rows_list, hw[0], hw[1], remove_borders=cfg["remove_borders"],
                                     min_length=cfg["min_length"], max_keylines=cfg["max_keylines"],
                                     token_distance=cfg["token_distance"], max_tokens=cfg["max_tokens"],
                                     valid_masks=valid_masks)
    tb = eng.tokenize(recs, cu_k, cu_n, dd, ds, token_distance=cfg["token_distance"], max_tokens=cfg["max_tokens"],
                      align_corners=align_corners)
    ld = eng.forward(tb)
    torch.cuda.synchronize()
    return tb, ld


def A_from(tb, img=0):
    k0, k1 = tb.cu_k[img], tb.cu_k[img + 1]
    n0, n1 = tb.cu_n[img], tb.cu_n[img + 1]
    s2l = tb.sub2line[n0:n1].cpu().numpy()
    cnt = np.bincount(s2l, minlength=k1 - k0)
    A = np.zeros((k1 - k0, n1 - n0), np.float32)
    A[s2l, np.arange(n1 - n0)] = (1.0 / cnt[s2l]).astype(np.float32)
    return A


def check_tokens(tb, g, prefix="", img=0, keys=TOK_KEYS):
    k0, k1 = tb.cu_k[img], tb.cu_k[img + 1]
    n0, n1 = tb.cu_n[img], tb.cu_n[img + 1]
    got = {
        "klines": tb.klines[k0:k1], "length_klines": tb.length[k0:k1], "angles": tb.angles[k0:k1],
        "sublines": tb.sublines[n0:n1], "pnt_sublines": tb.pnt[n0:n1], "mask_sublines": tb.mask[n0:n1][..., None],
        "resp_sublines": tb.resp[n0:n1][..., None], "angle_sublines": tb.angle_sub[n0:n1],
        "score_sublines": tb.score[n0:n1][..., None],
    }
    for k in keys:
        want = g[prefix + k][0]
        if k == "mat_klines2sublines":
            have = A_from(tb, img)
        else:
            have = got[k].cpu().numpy()
        assert have.shape == want.shape, (k, have.shape, want.shape)
        if k == "angles" or k == "angle_sublines":
            # host libm vs numpy cos/sin may differ in the last float64 ulp before the f32 cast
            assert np.abs(have - want).max() <= 1.2e-7, k
        else:
            assert np.array_equal(have, want), (k, np.abs(have - want).max())


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x3"])
def test_cfg2_pair_golden_other_precisions(engine, mode):
    """The stated contract (descriptors within 1e-4 of the reference, line matches identical by index) holds in every
    arithmetic mode, not only in the fp32-faithful default."""
    g = load("cfg2_pair")
    hw = (480, 640)
    maps = [synth.synth_dense_maps(int(g[f"{t}_seed"]), *hw) for t in "ab"]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    engine.set_precision(mode)
    try:
        tb, ld = run_native(engine, [g["a_lines"], g["b_lines"]], dd, ds, hw, BASE_CFG)
        dk, off, m01 = engine.match(ld[:199], np.array([0, 199]), tb.sub2line[:199], np.array([0, 199]), ld[199:],
                                    np.array([0, 199]), tb.sub2line[199:], np.array([0, 199]), 0.8, True)
        ld = ld.cpu().numpy()
    finally:
        engine.set_precision("bf16x6")
    for i, t in enumerate("ab"):
        assert np.abs(ld[199 * i:199 * (i + 1)].T - g[f"{t}_line_desc"][0]).max() < 1e-4
    M = np.zeros((199, 199))
    m = m01.cpu().numpy()
    M[np.nonzero(m >= 0)[0], m[m >= 0]] = 1
    assert np.array_equal(M, g["pair_M"][0])


def test_cfg2_pair_golden(engine):
    g = load("cfg2_pair")
    hw = (480, 640)
    maps = [synth.synth_dense_maps(int(g[f"{t}_seed"]), *hw) for t in "ab"]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    tb, ld = run_native(engine, [g["a_lines"], g["b_lines"]], dd, ds, hw, BASE_CFG)
    assert list(np.diff(tb.cu_n)) == [199, 199]
    ld = ld.cpu().numpy()
    for i, t in enumerate("ab"):
        check_tokens(tb, g, prefix=f"{t}_", img=i)
        n0, n1 = tb.cu_n[i], tb.cu_n[i + 1]
        desc = tb.desc[n0:n1].cpu().numpy()
        ii, jj = g[f"{t}_desc_sample_idx"].T
        assert np.abs(desc[ii, jj] - g[f"{t}_desc_sample"]).max() < TOKDESC_TOL
        assert np.abs(desc.astype(np.float64).sum(-1) - g[f"{t}_desc_checksum"]).max() < 1e-4
        err = np.abs(ld[n0:n1].T - g[f"{t}_line_desc"][0]).max()
        print(f"image {t}: max |line_desc - reference| = {err:.3e}")
        assert err < DESC_TOL
    dk, off, m01 = engine.match(torch.from_numpy(ld[:199]).cuda(), np.array([0, 199]), tb.sub2line[:199],
                                np.array([0, 199]), torch.from_numpy(ld[199:]).cuda(), np.array([0, 199]),
                                tb.sub2line[199:], np.array([0, 199]), 0.8, True)
    Dk = dk.cpu().numpy().reshape(199, 199)
    assert np.abs(Dk - g["pair_Dk"][0]).max() < 1e-4
    M = np.zeros((199, 199))
    m = m01.cpu().numpy()
    M[np.nonzero(m >= 0)[0], m[m >= 0]] = 1
    assert np.array_equal(M, g["pair_M"][0])          # matches bit-exact by index


@pytest.mark.parametrize("name", ["tiny_default", "tiny_float_td", "tiny_align_true", "tiny_max3", "tiny_noborder"])
def test_tiny_cases(engine, name):
    g = load(name)
    dd, ds, hw = tiny_maps(g)
    cfg = golden_cfg(g)
    tb, ld = run_native(engine, [g["lines"].copy()], dd.cuda(), ds.cuda(), hw, cfg, bool(g["align_corners"]))
    check_tokens(tb, g)
    assert np.abs(tb.desc.cpu().numpy() - g["desc_sublines"][0]).max() < TOKDESC_TOL
    assert np.abs(ld.cpu().numpy().T - g["line_desc"][0]).max() < DESC_TOL


def test_tiny_two_layers():
    from linetr_amd.engine import Engine
    g = load("tiny_two_layers")
    dd, ds, hw = tiny_maps(g)
    eng = Engine(weights_for(g), "cuda:0", n_line_descriptive_layers=2)
    tb, ld = run_native(eng, [g["lines"].copy()], dd.cuda(), ds.cuda(), hw, BASE_CFG)
    assert np.abs(ld.cpu().numpy().T - g["line_desc"][0]).max() < DESC_TOL


def test_tiny_validmask(engine):
    g = load("tiny_validmask")
    dd, ds, hw = tiny_maps(g)
    vm = np.ones(hw)
    vm[:, :int(g["valid_mask_cols"])] = 0
    tb, ld = run_native(engine, [g["lines"].copy()], dd.cuda(), ds.cuda(), hw, BASE_CFG, valid_masks=[vm])
    check_tokens(tb, g, keys=["klines", "sublines", "mat_klines2sublines"])
    assert np.abs(ld.cpu().numpy().T - g["line_desc"][0]).max() < DESC_TOL


def test_cfg5_small_long_tokens():
    from linetr_amd.engine import Engine
    g = load("cfg5_small")
    dd, ds, hw = tiny_maps(g)
    eng = Engine(weights_for(g), "cuda:0", image_shape=list(hw))
    cfg = dict(BASE_CFG, max_tokens=int(g["max_tokens"]))
    tb, ld = run_native(eng, [g["lines"]], dd.cuda(), ds.cuda(), hw, cfg)
    check_tokens(tb, g, keys=["klines", "sublines", "pnt_sublines", "mask_sublines", "resp_sublines",
                              "angle_sublines", "score_sublines"])
    desc = tb.desc.cpu().numpy().astype(np.float64)
    assert np.abs(desc.sum(-1) - g["desc_checksum"]).max() < 1e-4
    assert np.abs(ld.cpu().numpy().T - g["line_desc"][0]).max() < DESC_TOL


def test_matcher_known_answers(engine):
    g = load("matcher_cases")
    M, D = g["point_M"][0], g["point_D"][0]
    dist, m01 = engine.match_points(torch.from_numpy(g["point_desc0"]).cuda(), torch.from_numpy(g["point_desc1"]).cuda(),
                                    0.7, True)
    torch.cuda.synchronize()
    assert np.abs(dist.cpu().numpy() - D).max() < 1e-5
    got = np.zeros_like(M)
    m = m01.cpu().numpy()
    got[np.nonzero(m >= 0)[0], m[m >= 0]] = 1
    assert np.array_equal(got, M)


def test_batch_vs_oracle_random(engine):
    """A 6-image var-len batch (different line counts) against the CPU oracle run image by image."""
    hw = (480, 640)
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    rows, dds, dss = [], [], []
    for i, n in enumerate((40, 200, 7, 120, 64, 1)):
        rows.append(synth.synth_lines(300 + i, n, *hw))
        dd, ds = synth.synth_dense_maps(300 + i, *hw)
        dds.append(dd); dss.append(ds)
    tb, ld = run_native(engine, rows, torch.cat(dds).cuda(), torch.cat(dss).cuda(), hw, BASE_CFG)
    ld = ld.cpu().numpy()
    for i in range(len(rows)):
        out = oracle_image(sd, rows[i], dds[i], dss[i], hw, BASE_CFG)
        n0, n1 = tb.cu_n[i], tb.cu_n[i + 1]
        if len(out["klines"]) == 0:
            assert n1 == n0
            continue
        assert out["line_desc"].shape[2] == n1 - n0
        assert np.array_equal(tb.pnt[n0:n1].cpu().numpy(), out["pnt_sublines"][0].numpy())
        assert np.abs(ld[n0:n1].T - out["line_desc"][0].numpy()).max() < DESC_TOL


def run_fused(eng, rows_list, dd, ds, hw, cfg, align_corners=False, want_tokens=False):
    recs, cu_k, cu_n = eng.prefilter(rows_list, hw[0], hw[1], remove_borders=cfg["remove_borders"],
                                     min_length=cfg["min_length"], max_keylines=cfg["max_keylines"],
                                     token_distance=cfg["token_distance"], max_tokens=cfg["max_tokens"])
    tb, ld = eng.describe(recs, cu_k, cu_n, dd, ds, token_distance=cfg["token_distance"], max_tokens=cfg["max_tokens"],
                          align_corners=align_corners, want_tokens=want_tokens)
    torch.cuda.synchronize()
    return tb, ld


def test_fused_describe_matches_golden_and_dense_path(engine):
    """linetr_describe (real tokens only, on-the-fly sampling, padding key with multiplicity) vs the reference
    fixtures and vs the dense linetr_tokenize + linetr_forward path."""
    g = load("cfg2_pair")
    hw = (480, 640)
    maps = [synth.synth_dense_maps(int(g[f"{t}_seed"]), *hw) for t in "ab"]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    rows = [g["a_lines"], g["b_lines"]]
    tb, ld = run_fused(engine, rows, dd, ds, hw, BASE_CFG, want_tokens=True)
    tb_d, ld_d = run_native(engine, rows, dd, ds, hw, BASE_CFG)
    for i, t in enumerate("ab"):
        check_tokens(tb, g, prefix=f"{t}_", img=i)
        n0, n1 = tb.cu_n[i], tb.cu_n[i + 1]
        assert np.abs(ld[n0:n1].cpu().numpy().T - g[f"{t}_line_desc"][0]).max() < DESC_TOL
    assert torch.equal(tb.desc, tb_d.desc) and torch.equal(tb.pnt, tb_d.pnt)
    assert (ld - ld_d).abs().max().item() < 5e-6
    tb2, ld2 = run_fused(engine, rows, dd, ds, hw, BASE_CFG, want_tokens=False)      # lean variant
    assert torch.equal(ld2, ld) and tb2.desc.numel() == 0
    assert torch.equal(tb2.sublines, tb.sublines) and torch.equal(tb2.sub2line, tb.sub2line)


@pytest.mark.parametrize("name", ["tiny_default", "tiny_float_td", "tiny_align_true", "tiny_noborder"])
def test_fused_describe_tiny(engine, name):
    g = load(name)
    dd, ds, hw = tiny_maps(g)
    tb, ld = run_fused(engine, [g["lines"].copy()], dd.cuda(), ds.cuda(), hw, golden_cfg(g), bool(g["align_corners"]))
    assert np.abs(ld.cpu().numpy().T - g["line_desc"][0]).max() < DESC_TOL
    assert np.array_equal(tb.sublines.cpu().numpy(), g["sublines"][0])


def test_fused_describe_long_tokens_and_ragged_batch():
    from linetr_amd.engine import Engine
    g = load("cfg5_small")
    dd, ds, hw = tiny_maps(g)
    eng = Engine(weights_for(g), "cuda:0", image_shape=list(hw))
    cfg = dict(BASE_CFG, max_tokens=int(g["max_tokens"]))
    tb, ld = run_fused(eng, [g["lines"]], dd.cuda(), ds.cuda(), hw, cfg)
    assert np.abs(ld.cpu().numpy().T - g["line_desc"][0]).max() < DESC_TOL
    # ragged batch incl. an image whose only line is dropped by max_keylines=-1 (zero sub-lines)
    rows = [g["lines"][:50], g["lines"][:1], g["lines"][60:]]
    dd3, ds3 = torch.cat([dd] * 3).cuda(), torch.cat([ds] * 3).cuda()
    tb3, ld3 = run_fused(eng, rows, dd3, ds3, hw, cfg)
    tbd, ldd = run_native(eng, rows, dd3, ds3, hw, cfg)
    assert list(np.diff(tb3.cu_n))[1] == 0
    assert (ld3 - ldd).abs().max().item() < 5e-6
