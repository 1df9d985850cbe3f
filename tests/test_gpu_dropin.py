"""GPU: the reference's Python call surface (LineTransformer / Matching / nn_matcher / get_dist_matrix)
served by the HIP library, checked against the golden fixtures of the real reference."""
import numpy as np
import pytest
import torch

from helpers import BASE_CFG, TOK_KEYS, load, tiny_maps, train_mode_batches
from workloads import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
LT_CFG = {"mode": "train", "max_keylines": -1, "min_length": 16, "token_distance": 8, "nn_threshold": 0.8}
REF_KEYS = ["klines", "length_klines", "angles", "sublines", "pnt_sublines", "mask_sublines", "resp_sublines",
            "angle_sublines", "desc_sublines", "score_sublines", "mat_klines2sublines"]


def make_lt(**cfg):
    from models.line_transformer import LineTransformer      # the import path the reference's scripts use
    m = LineTransformer({**LT_CFG, **cfg}).eval()
    m.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
    return m.to("cuda")


def test_linetransformer_preprocess_forward_cfg2():
    g = load("cfg2_pair")
    m = make_lt()
    outs = []
    for t in "ab":
        dd, ds = synth.synth_dense_maps(int(g[f"{t}_seed"]), 480, 640)
        sp = {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}
        kl = synth.array_to_keylines(g[f"{t}_lines"])
        pre = m.preprocess(kl, (1, 1, 480, 640), sp, torch.ones(1, 1, 480, 640).cuda())   # tensor mask: ignored
        assert list(pre.keys()) == REF_KEYS
        assert m.config["image_shape"] == (1, 1, 480, 640)                                  # quirk: config mutated
        for k in TOK_KEYS:
            want = g[f"{t}_{k}"]
            have = pre[k].cpu().numpy()
            assert have.shape == want.shape and have.dtype == np.float32, k
            assert np.abs(have - want).max() <= (1.2e-7 if "angle" in k else 0), k
        out = m(pre)
        assert out is pre and out["line_desc"].shape == (1, 256, 199)
        assert np.abs(out["line_desc"].cpu().numpy() - g[f"{t}_line_desc"]).max() < 1e-4
        outs.append(out)
    # the matching tail exactly as models/matching.py:77-84 writes it, on the drop-in functions
    from models.line_transformer import get_dist_matrix
    from models.nn_matcher import nn_matcher_distmat
    D = get_dist_matrix(outs[0]["line_desc"].cpu().numpy(), outs[1]["line_desc"].cpu().numpy())[0]
    assert D.dtype == np.float32 and np.abs(D - g["pair_D"]).max() < 1e-4
    Dk = m.subline2keyline(D, outs[0]["mat_klines2sublines"][0], outs[1]["mat_klines2sublines"][0])
    assert Dk.shape == (1, 199, 199) and np.abs(Dk - g["pair_Dk"]).max() < 1e-4
    M = nn_matcher_distmat(Dk, 0.8, True)
    assert M.dtype == np.float64 and np.array_equal(M, g["pair_M"])


def test_linetransformer_quirks_and_empty():
    g = load("tiny_validmask")
    dd, ds, hw = tiny_maps(g)
    sp = {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}
    m = make_lt()
    vm = np.ones(hw)
    vm[:, :int(g["valid_mask_cols"])] = 0
    out = m(m.preprocess(synth.array_to_keylines(g["lines"]), (1, 1, *hw), sp, vm))       # ndarray mask honoured
    assert np.array_equal(out["klines"].cpu().numpy(), g["klines"])
    assert np.array_equal(out["mat_klines2sublines"].cpu().numpy(), g["mat_klines2sublines"])
    assert np.abs(out["line_desc"].cpu().numpy() - g["line_desc"]).max() < 1e-4
    g1 = load("tiny_single_line")
    pre = m.preprocess(synth.array_to_keylines(g1["lines"]), (1, 1, *hw), sp)
    assert len(pre["klines"]) == 0
    ret = m(pre)
    for k, v in ret.items():
        assert tuple(v.shape) == tuple(g1[f"ret_{k}_shape"])
    assert len(m.preprocess([], (1, 1, *hw), sp)["klines"]) == 0                          # zero detections: no crash
    bad = [synth.KeyLine(100, 100, 150, 100, length=400.0), synth.KeyLine(100, 200, 300, 200)]   # detector length > geometry
    with pytest.raises(AssertionError):
        m.preprocess(bad, (1, 1, *hw), sp)


def test_module_built_under_inference_mode_and_rebound_weights():
    """Inference tensors keep no version counter (reading `_version` raises): a module built, loaded and run under
    torch.inference_mode() must work, and re-binding a parameter's storage (p.data = ...) must rebuild the native engine."""
    g = load("tiny_validmask")
    dd, ds, hw = tiny_maps(g)
    with torch.inference_mode():
        m = make_lt()
        sp = {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}
        pre = m.preprocess(synth.array_to_keylines(g["lines"]), (1, 1, *hw), sp)
        ref = m(pre)["line_desc"].clone()
    m2 = make_lt()
    pre2 = m2.preprocess(synth.array_to_keylines(g["lines"]), (1, 1, *hw), sp)
    assert torch.equal(m2(pre2)["line_desc"], ref)
    p = m2.final_proj.weight
    p.data = -p.data.clone()                       # new storage, version counter untouched
    flipped = m2(pre2)["line_desc"].clone()
    assert not torch.allclose(flipped, ref, atol=1e-3)
    with torch.no_grad():
        m2.final_proj.weight.mul_(-1.0)            # in place: same storage, version counter bumped
    assert torch.equal(m2(pre2)["line_desc"], ref)
    m2.final_proj.weight = torch.nn.Parameter(-m2.final_proj.weight.detach().clone())   # a replaced Parameter object
    assert torch.equal(m2(pre2)["line_desc"], flipped)
    eng = m2._engine
    m2(pre2)
    assert m2._engine is eng                       # nothing changed: the engine is kept
    # a whole sub-module replaced after the first forward (its old parameter dicts stay intact): the engine must be rebuilt
    conv = torch.nn.Conv1d(256, 256, kernel_size=1).to("cuda")
    with torch.no_grad():
        conv.weight.copy_(-m2.final_proj.weight)
        conv.bias.copy_(-m2.final_proj.bias)
    m2.final_proj = conv
    out = m2(pre2)["line_desc"].clone()
    assert m2._engine is not eng and torch.allclose(out, -flipped, atol=1e-6)
    m2.register_buffer("unrelated", torch.zeros(1, device="cuda"))      # a new entry in a tracked dict: noticed as well
    eng = m2._engine
    m2(pre2)
    assert m2._engine is not eng


def test_valid_mask_of_another_shape_goes_through_numpy_indexing():
    """The native pre-filter addresses an ndarray mask as [height, width]; any other shape must behave like the reference's NumPy
    indexing (models/line_process.py:76-80): IndexError when an index is out of range, never an out-of-bounds read."""
    g = load("tiny_validmask")
    dd, ds, hw = tiny_maps(g)
    sp = {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}
    m = make_lt()
    kl = lambda: synth.array_to_keylines(g["lines"])
    with pytest.raises(IndexError):
        m.preprocess(kl(), (1, 1, *hw), sp, np.ones((1, 1, *hw)))            # 4-D like the tensor masks
    with pytest.raises(IndexError):
        m.preprocess(kl(), (1, 1, *hw), sp, np.ones((hw[0] // 2, hw[1] // 2)))
    big = np.ones((hw[0] + 7, hw[1] + 9))                                      # larger: NumPy indexes it happily; so must we
    big[:, :int(g["valid_mask_cols"])] = 0
    out = m(m.preprocess(kl(), (1, 1, *hw), sp, big))
    assert np.array_equal(out["klines"].cpu().numpy(), g["klines"])
    from linetr_amd.engine import Engine
    with pytest.raises(ValueError):
        m.engine().prefilter([g["lines"]], *hw, remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21,
                             valid_masks=[np.ones((3, 3))])


class FakeSuperPoint(torch.nn.Module):
    """Synthetic SuperPoint stand-in: seeded dense maps + random unit point descriptors."""
    config = {"nn_threshold": 0.7}

    def __init__(self, seeds):
        super().__init__()
        self.seeds = list(seeds)

    def forward(self, data):
        seed = self.seeds.pop(0)
        dd, ds = synth.synth_dense_maps(seed, *data["image"].shape[-2:])
        rs = np.random.RandomState(seed)
        n = 50 + seed % 7
        desc = torch.from_numpy(rs.standard_normal((256, n)).astype(np.float32))
        desc = torch.nn.functional.normalize(desc, dim=0)
        kp = torch.from_numpy(rs.uniform(8, 400, (n, 2)).astype(np.float32))
        dev = data["image"].device
        return {"keypoints": [kp.to(dev)], "scores": [torch.rand(n).to(dev)], "descriptors": [desc.to(dev)],
                "dense_descriptor": dd.to(dev), "dense_score": ds.to(dev)}


class FakeLSD:
    def __init__(self, line_sets):
        self.sets = list(line_sets)

    def detect_torch(self, image):
        return synth.array_to_keylines(self.sets.pop(0))


def test_matching_pipeline_outputs():
    from models.matching import Matching
    g = load("cfg2_pair")
    mt = Matching({"auto_min_length": True, "superpoint": {}, "lsd": {},
                   "linetransformer": {**LT_CFG}},
                  superpoint=FakeSuperPoint([int(g["a_seed"]), int(g["b_seed"])]), lsd=FakeLSD([g["a_lines"], g["b_lines"]]))
    mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
    mt = mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    pred = mt({"image0": img, "image1": img.clone()})
    for k in ("keypoints0", "keypoints1", "klines0", "klines1", "matches_p", "matching_scores_p", "matches_l",
              "matching_scores_l", "line_desc0", "line_desc1", "mat_klines2sublines0", "mat_klines2sublines1"):
        assert k in pred, k
    for k, v in pred.items():
        v[0]                                       # match_line_pairs.py:91-92 indexes every value
    assert mt.linetransformer.config["min_length"] == 16.0 and mt.linetransformer.config["token_distance"] == 8.0
    assert pred["matches_l"].dtype == torch.float64 and pred["matching_scores_l"].dtype == torch.float32
    assert np.array_equal(pred["matches_l"].numpy(), g["pair_M"])
    assert np.abs(pred["matching_scores_l"].numpy() - g["pair_Dk"]).max() < 1e-4
    assert pred["matches_p"].shape[0] == 1 and pred["matches_p"].dtype == torch.float64
    # np.savez payload of match_line_pairs.py:94-104
    out = {k: v[0].cpu().numpy() for k, v in pred.items() if torch.is_tensor(v[0])}
    assert out["klines0"].shape == (199, 2, 2) and out["matches_l"].shape == (199, 199)


def test_matching_forward_under_inference_mode_takes_the_one_call_tail(monkeypatch):
    """tensors made under torch.inference_mode() keep no version counter, so the sub-line maps riding along with the matrices
    could not be stamped and Matching.forward fell back to the separate calls (contents path, two host waits).  The matrices a forward call
    has just made itself are trusted as they are: the one-call tail (linetr_pair_tail) runs, and the results are those of the normal mode."""
    from models.matching import Matching
    from linetr_amd.engine import Engine
    g = load("cfg2_pair")

    def build():
        mt = Matching({"auto_min_length": True, "superpoint": {}, "lsd": {}, "linetransformer": {**LT_CFG}},
                      superpoint=FakeSuperPoint([int(g["a_seed"]), int(g["b_seed"])]), lsd=FakeLSD([g["a_lines"], g["b_lines"]]))
        mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
        return mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    ref = build()({"image0": img, "image1": img.clone()})
    calls = []
    real_tail = Engine.pair_tail
    monkeypatch.setattr(Engine, "pair_tail", lambda self, *a, **k: (calls.append(1), real_tail(self, *a, **k))[1])
    with torch.inference_mode():
        pred = build()({"image0": img, "image1": img.clone()})
    assert len(calls) == 1                                                     # not the fallback
    for k in ("matches_l", "matches_p", "matching_scores_l", "matching_scores_p"):
        assert torch.equal(pred[k], ref[k]), k
    assert torch.equal(pred["line_desc0"], ref["line_desc0"])


def test_matching_pair_of_two_image_sizes_like_the_reference():
    """models/matching.py:29-32 and :45-48: with auto_min_length, min_length / token_distance are recomputed for EACH image from that
    image's own shape.  A 480 x 640 + 960 x 1280 pair against a golden of the real reference (tests/golden/make_golden_mixed.py):
    every token tensor of both images bit for bit, descriptors <= 1e-4, matches_l identical, config left on image1's values."""
    from models.matching import Matching
    g = load("mixed_size_pair")
    mt = Matching({"auto_min_length": True, "linetransformer": {**LT_CFG}},
                  superpoint=FakeSuperPoint([int(g["map_seed0"]), int(g["map_seed1"])]), lsd=FakeLSD([g["lines0"], g["lines1"]]))
    mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
    mt = mt.eval().to("cuda")
    imgs = {s: torch.zeros(1, 1, *(int(v) for v in g["hw" + s]), device="cuda") for s in "01"}
    pred = mt({"image0": imgs["0"], "image1": imgs["1"]})
    assert float(g["min_length1"]) == 32.0 and float(g["token_distance1"]) == 16.0 and float(g["token_distance0"]) == 8.0
    assert mt.linetransformer.config["min_length"] == float(g["final_min_length"])
    assert mt.linetransformer.config["token_distance"] == float(g["final_token_distance"])
    for s in "01":
        for k in TOK_KEYS:
            want, have = g[k + s], pred[k + s].cpu().numpy()
            assert have.shape == want.shape, (k, s)
            assert np.abs(have - want).max() <= (1.2e-7 if "angle" in k else 0), (k, s)
        assert np.abs(pred["line_desc" + s].cpu().numpy() - g["line_desc" + s]).max() < 1e-4
    assert np.array_equal(pred["matches_l"].numpy(), g["matches_l"]) and g["matches_l"].sum() > 0
    assert np.abs(pred["matching_scores_l"].numpy() - g["matching_scores_l"]).max() < 1e-4


class AssetSuperPoint(torch.nn.Module):
    """Hands out the frozen outputs of the reference's SuperPoint on the asset pair (tests/golden/asset_pair.npz)."""
    config = {"nn_threshold": 0.7}

    def __init__(self, g):
        super().__init__()
        self.g, self.i = g, 0

    def forward(self, data):
        s = "01"[self.i % 2]
        self.i += 1
        dev = data["image"].device
        t = lambda k: torch.from_numpy(self.g[k + s]).to(dev)
        return {"keypoints": [t("keypoints")], "scores": [torch.ones(self.g["keypoints" + s].shape[0], device=dev)], "descriptors": [t("descriptors")],
                "dense_descriptor": t("dense_descriptor"), "dense_score": t("dense_score")}


def asset_margins(dk):
    two = np.sort(dk, axis=1)[:, :2]
    twoc = np.sort(dk, axis=0)[:2, :]
    return np.concatenate([two[:, 1] - two[:, 0], twoc[1] - twoc[0]])


def test_matching_on_the_reference_asset_pair_with_real_image_statistics():
    """match_line_pairs.py:80-104 on assets/input_pairs.txt:1 (scannet_0a / scannet_0b) as far as this container can pin it: the
    REFERENCE's SuperPoint (seeded weights) gave the dense maps and key-point descriptors, a deterministic edge-aligned line list stands
    in for cv2's LSD, and the reference's LineTransformer + matching tail gave the expected outputs (tests/golden/make_golden_asset_pair.py).
    Dense maps with real-image statistics (neighbouring descriptor cells at cosine 0.98): every token tensor bit for bit (angles: libm's
    last ulp), sampled token descriptors <= 1e-6, line descriptors <= 1e-4, Dk <= 1e-4, line AND point matches identical by index."""
    from models.matching import Matching
    g = load("asset_pair")
    mt = Matching({"auto_min_length": True, "superpoint": {"nn_threshold": 0.7}, "linetransformer": {**LT_CFG}},
                  superpoint=AssetSuperPoint(g), lsd=FakeLSD([g["lines0"], g["lines1"]]))
    mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
    mt = mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    pred = mt({"image0": img, "image1": img.clone()})
    worst = 0.0
    for s in "01":
        for k in TOK_KEYS:
            want, have = g[k + s], pred[k + s].cpu().numpy()
            assert have.shape == want.shape, (k, s)
            assert np.abs(have - want).max() <= (1.2e-7 if "angle" in k else 0), (k, s)
        err = float(np.abs(pred["line_desc" + s].cpu().numpy() - g["line_desc" + s]).max())
        worst = max(worst, err)
        assert err < 1e-4
    assert np.array_equal(pred["matches_l"].numpy(), g["matches_l"]) and g["matches_l"].sum() >= 40
    dk_err = float(np.abs(pred["matching_scores_l"].numpy() - g["matching_scores_l"]).max())
    assert dk_err < 1e-4
    # Point matches: SuperPoint descriptors of neighbouring key points on a real image are nearly equal, so some argmins of the
    # 1024 x 1024 distance matrix have margins far below fp32 dot-product noise (smallest: 8e-9; 23 rows / columns under 1e-6) -- there the
    # reference's own answer depends on its BLAS's summation order.  The contract (bench.py ARGMIN_CONTRACT): identical wherever every
    # margin involved exceeds 4 x the measured distance error; a differing row must sit on such a near-tie (or on the threshold).
    dp = pred["matching_scores_p"].numpy()[0]
    p_err = max(float(np.abs(dp.min(1) - g["matching_scores_p_rowmin"]).max()), float(np.abs(dp.min(0) - g["matching_scores_p_colmin"]).max()))
    assert p_err < 1e-5
    mp = pred["matches_p"].numpy()[0]
    got_idx, want_idx = np.where(mp.sum(1) > 0, mp.argmax(1), -1), g["matches_p_index"]
    d64 = np.clip(2.0 - 2.0 * g["descriptors0"].astype(np.float64).T @ g["descriptors1"].astype(np.float64), 0, None)
    r2, c2 = np.sort(d64, axis=1)[:, :2], np.sort(d64, axis=0)[:2, :]
    row_m, col_m = r2[:, 1] - r2[:, 0], c2[1] - c2[0]
    tol = 4 * max(p_err, 2.5e-7)
    differing = np.nonzero(got_idx != want_idx)[0]
    for i in differing:
        js = [j for j in (got_idx[i], want_idx[i]) if j >= 0]
        at_risk = row_m[i] < tol or any(col_m[j] < tol for j in js) or any(abs(d64[i, j] - 0.7) < tol for j in js)
        assert at_risk, (i, got_idx[i], want_idx[i], row_m[i], [col_m[j] for j in js])
    n_risk = int((row_m < tol).sum() + (col_m < tol).sum())
    assert len(differing) <= n_risk and abs(int(mp.sum()) - int(g["matches_p_count"])) <= n_risk
    print(f"asset pair, point matcher: max |distance - reference| = {p_err:.2e}; argmin margins below {tol:.1e}: {n_risk} of 2048 "
          f"(smallest {min(row_m.min(), col_m.min()):.1e}); rows whose match differs from the reference's: {len(differing)}, all on such near-ties")
    mg = asset_margins(g["matching_scores_l"][0])
    print(f"asset pair: max |line_desc - reference| = {worst:.3e}, max |Dk - reference| = {dk_err:.3e}, smallest argmin margin {mg.min():.3e}, "
          f"margins below 4 x the Dk error: {int((mg < 4 * dk_err).sum())} of {mg.size}")
    assert int((mg < 4 * max(dk_err, 1e-7)).sum()) == 0          # no argmin of this pair rests on summation order


def test_asset_pair_through_the_raw_tokeniser_and_the_batched_path():
    """The same fixture through linetr_tokenize (the reference's dense desc_sublines: sampled token descriptors against frozen samples,
    <= 1e-6 on correlated maps) and through the fused batched path the benchmark times (linetr_describe, NCHW- and NHWC-fed)."""
    from linetr_amd.engine import Engine
    g = load("asset_pair")
    eng = Engine(synth.calibrated_state_dict(), "cuda:0")
    dd = torch.cat([torch.from_numpy(g["dense_descriptor" + s]) for s in "01"]).cuda()
    ds = torch.cat([torch.from_numpy(g["dense_score" + s]) for s in "01"]).cuda()
    cfg = dict(remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21)
    recs, cu_k, cu_n = eng.prefilter([g["lines0"], g["lines1"]], 480, 640, **cfg)
    tb = eng.tokenize(recs, cu_k, cu_n, dd, ds, token_distance=8, max_tokens=21)
    for i, s in enumerate("01"):
        n0, n1 = tb.cu_n[i], tb.cu_n[i + 1]
        desc = tb.desc[n0:n1].cpu().numpy()
        ii, jj = g["desc_sample_idx" + s].T
        assert np.abs(desc[ii, jj] - g["desc_sample" + s]).max() < 1e-6
        assert np.abs(desc.astype(np.float64).sum(-1) - g["desc_checksum" + s]).max() < 1e-4
        assert np.array_equal(tb.score[n0:n1].cpu().numpy()[..., None], g["score_sublines" + s][0])
    lines = [g["lines0"], g["lines1"]]
    off = np.array([0, len(lines[0]), len(lines[0]) + len(lines[1])], np.int32)
    outs = []
    for layout, feed in (("nchw", dd), ("nhwc", dd.permute(0, 2, 3, 1).contiguous())):
        tb2, ld = eng.describe_lines(np.concatenate(lines), off, feed, ds, dense_layout=layout, **cfg)
        for i, s in enumerate("01"):
            assert np.abs(ld[tb2.cu_n[i]:tb2.cu_n[i + 1]].cpu().numpy().T - g["line_desc" + s][0]).max() < 1e-4
        outs.append(ld)
    assert torch.equal(outs[0], outs[1])


def test_matching_forward_batch_equals_per_pair():
    """forward_batch (one fused describe + one match call for all pairs) vs forward() pair by pair."""
    from models.matching import Matching
    seeds = [31, 32, 33, 34, 35, 36]
    line_sets = [synth.synth_lines(s, 150 + 10 * i, 480, 640) for i, s in enumerate(seeds)]

    def build():
        mt = Matching({"auto_min_length": True, "linetransformer": {**LT_CFG}}, superpoint=FakeSuperPoint(seeds),
                      lsd=FakeLSD(line_sets))
        mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
        return mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    pairs = [{"image0": img, "image1": img.clone()} for _ in range(3)]
    batch = build().forward_batch(pairs)
    single = build()
    assert len(batch) == 3
    for p in range(3):
        ref = single({"image0": img, "image1": img.clone()})
        got = batch[p]
        for k in ("matches_l", "matches_p"):
            assert torch.equal(got[k], ref[k]), (p, k)
        assert (got["matching_scores_l"] - ref["matching_scores_l"]).abs().max().item() < 1e-5
        for k in ("klines0", "klines1", "mat_klines2sublines0", "mat_klines2sublines1", "sublines1", "length_klines0", "resp_sublines1",
                  "angles0", "angles1", "angle_sublines0", "angle_sublines1"):      # r05: the angles too (NumPy's in both surfaces)
            assert torch.equal(got[k].cpu(), ref[k].cpu()), (p, k)
        assert torch.equal(got["matching_scores_p"], ref["matching_scores_p"])          # the point matcher runs on the device in both
        assert hasattr(got["mat_klines2sublines0"], "_linetr_sub2line")
        assert (got["line_desc0"] - ref["line_desc0"]).abs().max().item() < 5e-6
        for v in got.values():
            v[0]


def test_forward_batch_uses_producer_layout():
    """A SuperPoint that also hands out 'dense_descriptor_nhwc' (FusedHeadSuperPoint): forward_batch feeds that map to
    linetr_describe (dense_layout='nhwc') and returns the same line descriptors and matches."""
    from models.matching import Matching
    seeds = [51, 52, 53, 54]
    line_sets = [synth.synth_lines(s, 120 + 15 * i, 480, 640) for i, s in enumerate(seeds)]

    class NhwcSuperPoint(FakeSuperPoint):
        def forward(self, data):
            out = super().forward(data)
            out["dense_descriptor_nhwc"] = out["dense_descriptor"].permute(0, 2, 3, 1).contiguous()
            return out

    def build(cls):
        mt = Matching({"auto_min_length": True, "linetransformer": {**LT_CFG}}, superpoint=cls(seeds), lsd=FakeLSD(line_sets))
        mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
        return mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    pairs = [{"image0": img, "image1": img.clone()} for _ in range(2)]
    a = build(FakeSuperPoint).forward_batch(pairs)
    b = build(NhwcSuperPoint).forward_batch(pairs)
    for pa, pb in zip(a, b):
        assert torch.equal(pa["line_desc0"], pb["line_desc0"]) and torch.equal(pa["line_desc1"], pb["line_desc1"])
        assert torch.equal(pa["matches_l"], pb["matches_l"])
        assert "dense_descriptor_nhwc0" in pb and "dense_descriptor_nhwc0" not in pa


def test_nhwc_dense_layout_matches_nchw():
    from linetr_amd.engine import Engine
    eng = Engine(synth.calibrated_state_dict(), "cuda:0")
    rows = [synth.synth_lines(41, 120, 480, 640), synth.synth_lines(42, 90, 480, 640)]
    maps = [synth.synth_dense_maps(s, 480, 640) for s in (41, 42)]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    off = np.array([0, 120, 210], np.int32)
    kw = dict(remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21)
    _, a = eng.describe_lines(np.concatenate(rows), off, dd, ds, **kw)
    _, b = eng.describe_lines(np.concatenate(rows), off, dd.permute(0, 2, 3, 1).contiguous(), ds, dense_layout="nhwc", **kw)
    assert torch.equal(a, b)


def test_nn_matcher_known_answers():
    from models.nn_matcher import nn_matcher, nn_matcher_distmat
    g = load("matcher_cases")
    for k in ("ties", "big", "empty0", "empty1"):
        assert np.array_equal(nn_matcher_distmat(g[f"{k}_dist"], 0.8, True), g[f"{k}_mutual"]), k
        assert np.array_equal(nn_matcher_distmat(g[f"{k}_dist"], 0.8, False), g[f"{k}_oneway"]), k
    M, D = nn_matcher(g["point_desc0"], g["point_desc1"], 0.7, True)
    assert np.array_equal(M, g["point_M"]) and np.abs(D - g["point_D"]).max() < 1e-5


def _golden_matching(g, **cfg):
    from models.matching import Matching
    mt = Matching({"auto_min_length": True, "superpoint": {}, "lsd": {}, "linetransformer": {**LT_CFG}, **cfg},
                  superpoint=FakeSuperPoint([int(g["a_seed"]), int(g["b_seed"])]), lsd=FakeLSD([g["a_lines"], g["b_lines"]]))
    mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
    return mt.eval().to("cuda")


def test_matching_anchor_caching_like_the_demo():
    """demo_LineTR.py:155,207,250: the anchor frame's outputs are fed back as keypoints0 / scores0 / descriptors0 /
    klines0 / line_desc0 / mat_klines2sublines0 (+ image0) and only image 1 is processed.  With frame b of the golden
    pair as anchor and frame a as the new image the line matches are the golden pair's, transposed."""
    g = load("cfg2_pair")
    mt = _golden_matching(g)
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    pred = mt({"image0": img, "image1": img.clone()})
    keys = ["keypoints", "scores", "descriptors", "klines", "line_desc", "mat_klines2sublines"]     # demo_LineTR.py:155
    last = {k + "0": pred[k + "1"] for k in keys}
    last["image0"] = img
    mt.superpoint.seeds = [int(g["a_seed"])]          # only ONE image may go through SuperPoint / LSD now
    mt.lsd.sets = [g["a_lines"]]
    pred2 = mt({**last, "image1": img.clone()})
    assert mt.superpoint.seeds == [] and mt.lsd.sets == []
    assert "klines0" not in pred2 and "keypoints0" not in pred2 and "klines1" in pred2 and "keypoints1" in pred2
    assert np.array_equal(pred2["matches_l"].numpy()[0], g["pair_M"][0].T)
    assert np.abs(pred2["matching_scores_l"].numpy()[0] - g["pair_Dk"][0].T).max() < 1e-4
    assert torch.equal(pred2["klines1"].cpu(), pred["klines0"].cpu())
    assert (pred2["line_desc1"] - pred["line_desc0"]).abs().max().item() < 1e-6
    # and the demo reads them back exactly like this (demo_LineTR.py:207-222)
    kl0 = last["klines0"][0].cpu().numpy()
    kl1 = pred2["klines1"][0].cpu().numpy()
    ml = np.where(pred2["matches_l"][0].cpu().numpy() > 0)
    assert kl0[ml[0]].shape == kl1[ml[1]].shape == (int(g["pair_M"].sum()), 2, 2)


def test_npz_payload_roundtrip_like_match_line_pairs(tmp_path):
    """match_line_pairs.py:91-104: every value of the returned dict is indexed with [0], tensors go through
    .cpu().numpy(), and eight arrays are written with np.savez.  Same key set, shapes and dtypes here."""
    g = load("cfg2_pair")
    pred_matches = _golden_matching(g)({"image0": torch.zeros(1, 1, 480, 640, device="cuda"),
                                        "image1": torch.zeros(1, 1, 480, 640, device="cuda")})
    pred = {k: v[0].cpu().numpy() for k, v in pred_matches.items() if torch.is_tensor(v[0])}
    pred = {**pred, **{k: v[0] for k, v in pred_matches.items() if not torch.is_tensor(v[0])}}
    out = {"keypoints0": pred["keypoints0"], "keypoints1": pred["keypoints1"], "matches_p": pred["matches_p"],
           "match_confidence_p": pred["matching_scores_p"], "keylines0": pred["klines0"], "keylines1": pred["klines1"],
           "matches_l": pred["matches_l"], "match_confidence_l": pred["matching_scores_l"]}
    path = tmp_path / "a_b_matches.npz"
    np.savez(str(path), **out)
    z = np.load(str(path))
    assert sorted(z.files) == sorted(out)
    n0, n1 = z["keypoints0"].shape[0], z["keypoints1"].shape[0]
    want = {"keypoints0": ((n0, 2), np.float32), "keypoints1": ((n1, 2), np.float32), "matches_p": ((n0, n1), np.float64),
            "match_confidence_p": ((n0, n1), np.float32), "keylines0": ((199, 2, 2), np.float32),
            "keylines1": ((199, 2, 2), np.float32), "matches_l": ((199, 199), np.float64),
            "match_confidence_l": ((199, 199), np.float32)}
    for k, (shape, dt) in want.items():
        assert z[k].shape == shape and z[k].dtype == dt, (k, z[k].shape, z[k].dtype)
    assert np.array_equal(z["matches_l"], g["pair_M"][0]) and np.array_equal(z["keylines0"], g["a_klines"][0])
    assert np.abs(z["match_confidence_l"] - g["pair_Dk"][0]).max() < 1e-4


def test_forward_batch_on_the_golden_pair():
    """forward_batch fed the reference-generated cfg2 pair (twice, as a batch of two pairs): matches identical to the
    golden, descriptors within 1e-4, key-lines bit-exact (the golden has no equal-length lines, so the native
    pre-filter's tie rule does not come into play)."""
    g = load("cfg2_pair")
    from models.matching import Matching
    seeds = [int(g["a_seed"]), int(g["b_seed"])] * 2
    mt = Matching({"auto_min_length": True, "linetransformer": {**LT_CFG}}, superpoint=FakeSuperPoint(seeds),
                  lsd=FakeLSD([g["a_lines"], g["b_lines"]] * 2))
    mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
    mt = mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    out = mt.forward_batch([{"image0": img, "image1": img.clone()} for _ in range(2)])
    for pred in out:
        assert np.array_equal(pred["matches_l"].numpy(), g["pair_M"])
        assert np.abs(pred["matching_scores_l"].numpy() - g["pair_Dk"]).max() < 1e-4
        for s, t in (("0", "a"), ("1", "b")):
            assert np.array_equal(pred["klines" + s].cpu().numpy(), g[f"{t}_klines"])
            assert np.abs(pred["line_desc" + s].cpu().numpy() - g[f"{t}_line_desc"]).max() < 1e-4
            assert np.array_equal(pred["mat_klines2sublines" + s].cpu().numpy(), g[f"{t}_mat_klines2sublines"])


def test_prefilter_tie_order_native_vs_numpy():
    """Equal-length lines: by default the batched pre-filter hands the images that hold ties to NumPy's own argsort
    (Engine.prefilter tie_order="numpy"), so its rows are the per-image path's -- the reference's on this host -- row for row;
    tie_order="stable" keeps the native, machine-independent order (stable argsort reversed), where only tied rows may sit
    elsewhere.  The descriptors follow the rows."""
    from linetr_amd.engine import Engine
    from linetr_amd.line_transformer import change_cv2_T_np, filter_by_length, remove_borders
    eng = Engine(synth.calibrated_state_dict(), "cuda:0")
    rows = synth.synth_lines(77, 40, 480, 640)
    rows[[5, 9, 21, 30, 31, 33], 4] = 20.0                  # six lines of equal detector length (<= their geometric length)
    other = synth.synth_lines(78, 40, 480, 640)             # an image without ties in the same batch
    kw = dict(remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21)
    kl = change_cv2_T_np(synth.array_to_keylines(rows))
    kl = filter_by_length(remove_borders(kl, 8, 480, 640, np.ones((480, 640))), 16, -1)
    recs, cu_k, cu_n = eng.prefilter([other, rows], 480, 640, **kw)
    mine = recs[cu_k[1]:cu_k[2]]
    assert np.array_equal(np.stack([mine["sp"], mine["ep"]], axis=1), kl["klines"])
    assert np.array_equal(mine["length"], kl["length_klines"]) and np.array_equal(mine["angle"], kl["angles"])
    stable, _, _ = eng.prefilter([other, rows], 480, 640, tie_order="stable", **kw)
    nat = np.stack([stable["sp"], stable["ep"]], axis=1)[cu_k[1]:cu_k[2]]
    key = lambda a: sorted(map(tuple, a.reshape(len(a), -1).tolist()))
    assert key(nat) == key(kl["klines"])
    tied = np.isin(kl["length_klines"], [20.0])
    assert np.array_equal(nat[~tied], kl["klines"][~tied])
    assert np.array_equal(stable[:cu_k[1]], recs[:cu_k[1]])                 # the tie-free image is the native pass's either way
    # end to end: forward_batch rows == forward rows on the tied image
    dd, ds = synth.synth_dense_maps(77, 480, 640)
    do, so = synth.synth_dense_maps(78, 480, 640)
    tb, ld = eng.describe_lines(np.concatenate([other, rows]), np.array([0, 40, 80], np.int32), torch.cat([do, dd]).cuda(),
                                torch.cat([so, ds]).cuda(), **kw)
    assert np.array_equal(tb.klines[tb.cu_k[1]:tb.cu_k[2]].cpu().numpy(), kl["klines"].astype(np.float32))


def test_matching_wraps_the_host_superpoint(monkeypatch):
    """Matching() without an injected SuperPoint builds the host project's models.superpoint.SuperPoint and wraps it with
    FusedHeadSuperPoint (heads on linetr_superpoint_heads, channel-last map for the tokeniser); the line branch then
    gives the same descriptors and matches as the un-fused module."""
    import sys
    import types
    from test_gpu_producer import _Helpers, _StandInSuperPoint
    from linetr_amd.superpoint import FusedHeadSuperPoint

    class SuperPoint(_StandInSuperPoint):
        def __init__(self, config):
            super().__init__()
            self.config = {**self.config, "nn_threshold": 0.7, **config}
            self.load_state_dict({k: torch.from_numpy(v) for k, v in synth.superpoint_state_dict(0).items()})

    mod = types.ModuleType("models.superpoint")
    mod.SuperPoint = SuperPoint
    for fn in ("simple_nms", "remove_borders", "top_k_keypoints", "sample_descriptors"):
        setattr(mod, fn, getattr(_Helpers, fn))
    SuperPoint.__module__ = "models.superpoint"
    monkeypatch.setitem(sys.modules, "models.superpoint", mod)
    from models.matching import Matching
    lines = [synth.synth_lines(61, 80, 96, 128, 17.0, 60.0, margin=9.0), synth.synth_lines(62, 80, 96, 128, 17.0, 60.0, margin=9.0)]
    rs = np.random.RandomState(1)
    imgs = [torch.from_numpy(rs.rand(1, 1, 96, 128).astype(np.float32)).cuda() for _ in range(2)]
    preds = []
    for fuse in (True, False):
        mt = Matching({"auto_min_length": True, "linetransformer": {**LT_CFG}, "fuse_superpoint_heads": fuse}, lsd=FakeLSD(lines))
        assert isinstance(mt.superpoint, FusedHeadSuperPoint) == fuse
        mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
        mt = mt.eval().to("cuda")
        if not fuse:                                     # the plain stand-in has no forward of its own: borrow the wrapper's maths
            plain = FusedHeadSuperPoint(mt.superpoint)
            sp_forward = lambda data, plain=plain: {k: v for k, v in plain(data).items() if k != "dense_descriptor_nhwc"}
            mt.superpoint.forward = sp_forward
        preds.append(mt({"image0": imgs[0], "image1": imgs[1]}))
    a, b = preds
    assert "dense_descriptor_nhwc0" in a and "dense_descriptor_nhwc0" not in b
    assert torch.equal(a["matches_l"], b["matches_l"]) and torch.equal(a["klines0"].cpu(), b["klines0"].cpu())
    assert (a["line_desc1"] - b["line_desc1"]).abs().max().item() < 1e-6


def test_training_time_batched_forward_golden():
    """8(f) row 4: LineTransformer.forward on a dict with a batch axis (B = 3, fixed 40 sub-lines per image, 12
    line-descriptive layers) exactly as train.py:163-164 feeds the model with the dataset builder's fixed-size samples
    (util_lines.py:670-766); expected descriptors frozen from the real reference (tests/golden/make_golden_train.py).
    Forward only -- the loss / backward are out of scope."""
    g = load("train_batch")
    hw = tuple(int(v) for v in g["hw"])
    n_fix, nl = int(g["n_fix"]), int(g["n_desc_layers"])
    from models.line_transformer import LineTransformer
    m = LineTransformer({**LT_CFG, "n_line_descriptive_layers": nl}).eval()
    m.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict(nl)), strict=True)
    m = m.to("cuda")
    keys = ("sublines", "pnt_sublines", "desc_sublines", "score_sublines", "mask_sublines", "resp_sublines", "angle_sublines", "klines")
    outs = []
    for b in range(3):
        dd, ds = synth.synth_dense_maps_np(int(g[f"map_seed_{b}"]), *hw)
        sp = {"dense_descriptor": torch.from_numpy(dd).cuda(), "dense_score": torch.from_numpy(ds).cuda()}
        outs.append(m.preprocess(synth.array_to_keylines(g[f"lines_{b}"]), (1, 1, *hw), sp))
    batch = {k: torch.cat([o[k][:, :n_fix] for o in outs], dim=0) for k in keys}
    res = m(batch)
    assert res is batch and res["line_desc"].shape == (3, 256, n_fix)
    assert np.abs(res["line_desc"].cpu().numpy() - g["line_desc"]).max() < 1e-4
    # batch element b equals the same image pushed through alone (the signature attention is per image)
    alone = m({k: v[1:2].clone() for k, v in batch.items() if k != "line_desc"})
    assert (alone["line_desc"][0] - res["line_desc"][1]).abs().max().item() < 2e-6


def _shim_train_batches(g, m):
    hw = tuple(int(v) for v in g["hw"])
    from models.line_process import line_tokenizer
    pre = lambda rows, pred: m.preprocess(synth.array_to_keylines(rows), (1, 1, *hw), pred)
    tok = lambda lines, pred: line_tokenizer(lines, 8, 21, pred, (640, 480))      # conv_fixed_size's image_shape (util_lines.py:682,703)
    return train_mode_batches(g, pre, tok, to_dev=lambda t: t.cuda())


def test_line_tokenizer_with_the_dataset_builders_swapped_image_shape():
    """line_tokenizer's `image_shape` only sets the end-point clip (models/line_process.py:101,115-116); conv_fixed_size passes
    (640, 480) for 480 x 640 maps (dataloaders/utils/util_lines.py:682,703), which bends 29 of the fixture's 111 pseudo lines at
    x = 479.4.  Every tensor against the real reference's output for the same call."""
    from models.line_process import line_tokenizer
    g = load("train_mode")
    dd, ds = synth.synth_dense_maps_np(int(g["map_seed_0_0"]), 480, 640)
    pred = {"dense_descriptor": torch.from_numpy(dd).cuda(), "dense_score": torch.from_numpy(ds).cuda()}
    lines = {k: g[f"pseudo_{k}_0_0"].copy() for k in ("klines", "length_klines", "angles")}
    out = line_tokenizer(lines, 8, 21, pred, (640, 480))
    for k in TOK_KEYS:
        assert np.array_equal(out[k].cpu().numpy(), g[f"pseudo_tok_{k}"]), k
    assert out["klines"][0, :, 1, 0].max().item() <= 479.4 + 1e-4


def test_training_handle_refuses_encoder_widths_beyond_its_statistics_scratch():
    """the train-mode BatchNorm scratch holds 512 channels per layer; a keyline_encoder like [32, 64, 1024, 256] passed
    linetr_create and would have written past it.  Such a handle is refused at creation (inference handles take any width)."""
    from linetr_amd import _native as nat
    from linetr_amd.engine import Engine
    enc = (32, 64, 1024, 256)
    sd = synth.make_state_dict(3, enc=enc)
    with pytest.raises(nat.NativeError, match="training-mode handle"):
        Engine(sd, "cuda:0", keyline_encoder=list(enc), bn_batch_stats=True)
    Engine(sd, "cuda:0", keyline_encoder=list(enc))           # the inference handle of the same widths is fine


def test_train_mode_forward_golden():
    """8(f) row 4, train mode (train.py:127,163-164): the module in .train() called twice on batches of 3 x 250 sub-lines padded /
    truncated by the reference's conv_fixed_size, against a fixture of the real reference in .train() with dropout probability 0
    (tests/golden/make_golden_train_mode.py): BatchNorm on batch statistics, line_desc <= 1e-4, running_mean / running_var /
    num_batches_tracked of all 15 BatchNorm1d layers after each call.  Forward only."""
    from models.line_transformer import LineTransformer
    g = load("train_mode")
    nl = int(g["n_desc_layers"])
    m = LineTransformer({**LT_CFG, "n_line_descriptive_layers": nl})
    m.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict(nl)), strict=True)
    m = m.to("cuda").eval()
    batches = _shim_train_batches(g, m)
    eval_before = m({k: v.clone() for k, v in batches[0].items()})["line_desc"].clone()
    m.train()
    assert m.training and m.dropout == 0.1
    with pytest.raises(RuntimeError, match="dropout"):          # the reference's train mode has dropout 0.1: not silently dropped
        m(dict(batches[0]))
    m.dropout = 0.0
    for c, batch in enumerate(batches):
        res = m(batch)
        assert res is batch and res["line_desc"].shape == (3, 256, 250) and not res["line_desc"].requires_grad
        got = res["line_desc"].cpu().numpy()
        want = g[f"line_desc_{c}"]
        assert np.abs((got if c == 0 else got[:, :, ::5]) - want).max() < 1e-4
        sd = m.state_dict()
        for k, v in sd.items():
            if "running_" in k:
                ref = g[f"bn{c}.{k}"]
                assert np.abs(v.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (c, k)
            elif k.endswith("num_batches_tracked"):
                assert int(v) == int(g[f"bn{c}.{k}"]), (c, k)
    # back in eval mode the folded engine is rebuilt on the MOVED running statistics: the reference's eval forward with the fixture's
    # statistics after the second call (the oracle restates it; pinned by tests/test_oracle_golden.py)
    m.eval()
    eval_after = m({k: v.clone() for k, v in batches[0].items() if k != "line_desc"})["line_desc"]
    assert (eval_after - eval_before).abs().max().item() > 1e-4
    from oracle import linetr_oracle as O
    sd_t = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want = O.forward_batch(sd_t, {k: v.cpu() for k, v in batches[0].items() if k != "line_desc"}, (480, 640))["line_desc"]
    assert (eval_after.cpu() - want).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,n_fix,momentum,wseed", [(1, 7, 0.1, 0), (2, 33, 0.5, 3), (4, 50, 0.1, 0), (3, 1, 0.25, 5)])
def test_train_mode_forward_vs_oracle_random_batches(B, n_fix, momentum, wseed):
    """The train-mode forward against the oracle's restatement (pinned to the real reference by test_oracle_golden.py) on batches of
    other shapes than the fixture's: one sample, a single sub-line per image (BatchNorm over B rows), another momentum, raw seeded
    weights.  Descriptors <= 1e-4 where the batch statistics are well conditioned, running statistics to 2e-5 relative."""
    from models.line_transformer import LineTransformer
    from oracle import linetr_oracle as O
    sdn = synth.calibrated_state_dict() if wseed == 0 else synth.make_state_dict(wseed)
    m = LineTransformer({**LT_CFG}).eval()
    m.load_state_dict(synth.to_torch_state_dict(sdn), strict=True)
    m = m.to("cuda")
    keys = ("sublines", "pnt_sublines", "desc_sublines", "score_sublines", "mask_sublines", "resp_sublines", "angle_sublines", "klines")
    outs = []
    for b in range(B):
        dd, ds = synth.synth_dense_maps_np(700 + b, 480, 640)
        sp = {"dense_descriptor": torch.from_numpy(dd).cuda(), "dense_score": torch.from_numpy(ds).cuda()}
        outs.append(m.preprocess(synth.array_to_keylines(synth.synth_lines(700 + b, 70, 480, 640)), (1, 1, 480, 640), sp))
    batch = {k: torch.cat([o[k][:, :n_fix] for o in outs], dim=0) for k in keys}
    m.train()
    m.dropout = 0.0
    for bn in m._bn_layers():
        bn.momentum = momentum
    sd_t = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    want = O.forward_train(sd_t, {k: v.cpu() for k, v in batch.items()}, (480, 640), momentum=momentum)["line_desc"]
    got = m(batch)["line_desc"]
    assert got.shape == want.shape == (B, 256, n_fix)
    # three rows per BatchNorm: a variance of three samples divides fp32 round-off by a small number in eight places -- looser bound
    assert (got.cpu() - want).abs().max().item() < (1e-4 if B * n_fix >= 7 else 2e-3)
    for k, v in m.state_dict().items():
        if "running_" in k:
            ref = sd_t[k].numpy()
            assert np.abs(v.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), k
        elif k.endswith("num_batches_tracked"):
            assert int(v) == int(sd_t[k])


def test_matching_forward_fused_pair_equals_the_per_image_path(monkeypatch):
    """Matching.forward sends the two images of a pair through ONE fused native call (Matching._describe_fused); every entry of the
    reference's dict must be what the per-image path (preprocess + forward per image) returns: token tensors bit for bit, descriptors
    to fp32 round-off, matches and key order identical.  Also: a pair with a line-less image falls back without detecting twice."""
    from models.matching import Matching
    g = load("cfg2_pair")

    def build(lines):
        mt = Matching({"auto_min_length": False, "linetransformer": {**LT_CFG}},
                      superpoint=FakeSuperPoint([int(g["a_seed"]), int(g["b_seed"])]), lsd=FakeLSD(lines))
        mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
        return mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    fused_calls = []
    orig = Matching._describe_fused
    monkeypatch.setattr(Matching, "_describe_fused", lambda self, *a: fused_calls.append(1) or orig(self, *a))
    fused = build([g["a_lines"], g["b_lines"]])({"image0": img, "image1": img.clone()})
    assert fused_calls == [1]
    monkeypatch.setattr(Matching, "_describe_fused", lambda self, *a: None)
    plain = build([g["a_lines"], g["b_lines"]])({"image0": img, "image1": img.clone()})
    assert list(fused.keys()) == list(plain.keys())
    for k in plain:
        a, b = fused[k], plain[k]
        if not torch.is_tensor(a):
            continue
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if k.startswith("line_desc"):
            assert (a - b).abs().max().item() < 5e-6, k
            assert np.abs(a.cpu().numpy() - g[("a" if k.endswith("0") else "b") + "_line_desc"]).max() < 1e-4
        elif k == "matching_scores_l":
            assert (a - b).abs().max().item() < 1e-5
        else:
            assert torch.equal(a.cpu(), b.cpu()), k
    assert np.array_equal(fused["matches_l"].numpy(), g["pair_M"])
    assert hasattr(fused["mat_klines2sublines0"], "_linetr_sub2line")
    # equal detector lengths (NumPy's argsort decides the rows) and an ndarray valid mask (honoured, line_process.py:76-80): the fused
    # call -- native pre-filter, NumPy only for the tied image and the angles -- still returns the per-image path's tensors
    a_tied = g["a_lines"].copy()
    a_tied[[3, 17, 40, 41, 90, 150], 4] = 21.0
    vm = np.ones((480, 640))
    vm[:, :150] = 0
    monkeypatch.setattr(Matching, "_describe_fused", orig)
    f2 = build([a_tied, g["b_lines"]])({"image0": img, "image1": img.clone(), "valid_mask0": vm})
    monkeypatch.setattr(Matching, "_describe_fused", lambda self, *a: None)
    p2 = build([a_tied, g["b_lines"]])({"image0": img, "image1": img.clone(), "valid_mask0": vm})
    assert 0 < f2["klines0"].shape[1] < 199
    for k in p2:
        if torch.is_tensor(p2[k]) and not k.startswith("line_desc") and k != "matching_scores_l":
            assert torch.equal(f2[k].cpu(), p2[k].cpu()), k
    assert (f2["line_desc0"] - p2["line_desc0"]).abs().max().item() < 5e-6
    # one image without a single usable line: the per-image path takes over, the detector is asked once per image
    monkeypatch.setattr(Matching, "_describe_fused", orig)
    short = np.array([[100.0, 100.0, 105.0, 100.0, 5.0, 0.0]])
    mt = build([g["a_lines"], short])
    out = mt({"image0": img, "image1": img.clone()})
    assert mt.lsd.sets == [] and out["line_desc1"].shape == (1, 256, 0) and out["matches_l"].shape == (1, 199, 0)


def test_forward_many_equals_forward_image_by_image():
    """LineTransformer.forward_many: several pre-processed images through ONE native forward call; every dict gets the 'line_desc'
    forward() gives it (fp32 round-off), dicts without lines get default_ret(), a single live dict takes the plain path."""
    m = make_lt()
    pres, alone, sps = [], [], []
    for seed, n in ((41, 120), (42, 2), (43, 200)):
        dd, ds = synth.synth_dense_maps(seed, 480, 640)
        sp = {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}
        sps.append(sp)
        kl = synth.array_to_keylines(synth.synth_lines(seed, n, 480, 640))
        pres.append(m.preprocess(kl, (1, 1, 480, 640), sp))
        one = m.preprocess(kl, (1, 1, 480, 640), sp)
        alone.append(m(one)["line_desc"].clone())
    empty = m.preprocess([], (1, 1, 480, 640), sp)
    outs = m.forward_many([pres[0], empty, pres[1], pres[2]])
    assert outs[0] is pres[0] and outs[2] is pres[1] and outs[3] is pres[2]
    assert tuple(outs[1]["line_desc"].shape) == (1, 256, 0)
    for out, want in zip((outs[0], outs[2], outs[3]), alone):
        assert out["line_desc"].shape == want.shape
        assert (out["line_desc"] - want).abs().max().item() < 5e-6
    single = m.forward_many([m.preprocess(synth.array_to_keylines(synth.synth_lines(41, 120, 480, 640)), (1, 1, 480, 640), sps[0])])
    assert torch.equal(single[0]["line_desc"], alone[0])          # one live dict: the plain forward() path
