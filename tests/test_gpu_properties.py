"""GPU: size-independent properties of the hot path at BASELINE.json's FULL sizes (cfg3: 64 pairs x 200 lines,
cfg5: 1280x960 / 600 lines / 41 tokens), where the CPU oracle would take minutes:

  * batch invariance      -- an image described inside a 128-image batch == the same image described alone
  * permutation equivariance of the line-signature attention -- re-ordering the detector output of an image
    leaves every line's descriptor unchanged (lines are re-sorted by length, so equal output order)
  * unit-norm descriptors, finite outputs, Dk in [0, 4]
  * matching a batch against a jittered, permuted copy of itself recovers the permutation (>= 97 %)
  * precision modes: bf16x6 (default) == exact-fp32 MFMA to fp32 round-off; bf16x3 within 5e-5
"""
import numpy as np
import pytest
import torch

from helpers import BASE_CFG
from workloads import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
HW = (480, 640)
CFG = dict(remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21)


@pytest.fixture(scope="module")
def eng():
    from linetr_amd.engine import Engine
    return Engine(synth.calibrated_state_dict(), "cuda:0")


def batch_inputs(n_img, seed0=7000, n_lines=200, hw=HW, lo=17.0, hi=167.0):
    lines = [synth.synth_lines(seed0 + i, n_lines, hw[0], hw[1], lo, hi) for i in range(n_img)]
    maps = [synth.synth_dense_maps(seed0 + i, *hw) for i in range(n_img)]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)
    return lines, np.concatenate(lines), off, dd, ds


def describe(eng, cat, off, dd, ds, cfg=CFG):
    tb, ld = eng.describe_lines(cat, off, dd, ds, **cfg)
    torch.cuda.synchronize()
    return tb, ld


def test_cfg3_batch_invariance_and_norms(eng):
    lines, cat, off, dd, ds = batch_inputs(128)
    tb, ld = describe(eng, cat, off, dd, ds)
    assert tb.N == 128 * 199 and torch.isfinite(ld).all()
    assert (ld.norm(dim=1) - 1).abs().max().item() < 1e-5
    for i in (0, 57, 127):                         # same image alone: identical up to fp32 round-off
        tb1, ld1 = describe(eng, lines[i], np.array([0, len(lines[i])], np.int32), dd[i:i + 1], ds[i:i + 1])
        n0, n1 = tb.cu_n[i], tb.cu_n[i + 1]
        assert torch.equal(tb1.sublines, tb.sublines[n0:n1])
        assert (ld1 - ld[n0:n1]).abs().max().item() < 2e-6


def test_ragged_batch_through_the_fused_projection_attention_kernel(eng):
    """48 images with 3 .. 257 detected lines each (2 .. 256 sub-lines: one to eight 32-row wave tiles, partly filled last
    tiles, image boundaries that are not multiples of anything) described as ONE batch -- the batch runs the fused q/k/v
    projection + attention kernel (lt_attn_fused.h: >= 128 (image, head) blocks, <= 256 sub-lines per image) -- against
    every image described alone, which runs the separate projection GEMM and the small attention kernel."""
    counts = [3, 4, 33, 34, 65, 97, 128, 129, 160, 161, 193, 200, 224, 225, 256, 257] * 3
    lines = [synth.synth_lines(7700 + i, n, *HW) for i, n in enumerate(counts)]
    maps = [synth.synth_dense_maps(7700 + i, *HW) for i in range(len(counts))]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)
    tb, ld = describe(eng, np.concatenate(lines), off, dd, ds)
    assert int(np.diff(tb.cu_n).max()) == 256 and int(np.diff(tb.cu_n).min()) == 2
    assert torch.isfinite(ld).all() and (ld.norm(dim=1) - 1).abs().max().item() < 1e-5
    for i in range(len(counts)):
        tb1, ld1 = describe(eng, lines[i], np.array([0, len(lines[i])], np.int32), dd[i:i + 1], ds[i:i + 1])
        n0, n1 = tb.cu_n[i], tb.cu_n[i + 1]
        assert (ld1 - ld[n0:n1]).abs().max().item() < 2e-6, (i, counts[i])


def test_permutation_equivariance(eng):
    lines, cat, off, dd, ds = batch_inputs(4, seed0=7100)
    tb, ld = describe(eng, cat, off, dd, ds)
    rs = np.random.RandomState(3)
    shuffled = [l[rs.permutation(len(l))] for l in lines]
    tb2, ld2 = describe(eng, np.concatenate(shuffled), off, dd, ds)
    assert torch.equal(tb.klines, tb2.klines)      # the length sort makes the output order canonical
    assert (ld - ld2).abs().max().item() < 2e-6


def test_cfg3_matching_recovers_jittered_permutation(eng):
    n_pairs = 64
    lines0 = [synth.synth_lines(7300 + i, 200, *HW) for i in range(n_pairs)]
    jit = [synth.jitter_pair(l, 7400 + i, 0.3) for i, l in enumerate(lines0)]
    maps = [synth.synth_dense_maps(7300 + i, *HW) for i in range(n_pairs)]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    off = np.arange(n_pairs + 1, dtype=np.int32) * 200
    tb0, ld0 = describe(eng, np.concatenate(lines0), off, dd, ds)
    tb1, ld1 = describe(eng, np.concatenate([j[0] for j in jit]), off, dd, ds)
    dk, off_dk, m01 = eng.match(ld0, tb0.cu_n, tb0.sub2line, tb0.cu_k, ld1, tb1.cu_n, tb1.sub2line, tb1.cu_k, 0.8, True)
    dkc = dk.cpu()
    assert dkc.min().item() >= 0 and dkc.max().item() <= 4.0 + 1e-5
    m = m01.cpu().numpy()
    k0, k1 = tb0.klines.cpu().numpy(), tb1.klines.cpu().numpy()
    good = total = 0
    for p in range(n_pairs):
        a0, a1 = tb0.cu_k[p], tb0.cu_k[p + 1]
        b0 = tb1.cu_k[p]
        mm = m[a0:a1]
        idx = np.nonzero(mm >= 0)[0]
        d = np.abs(k0[a0:a1][idx] - k1[b0 + mm[idx]]).reshape(len(idx), -1).max(1)
        d2 = np.abs(k0[a0:a1][idx] - k1[b0 + mm[idx]][:, ::-1]).reshape(len(idx), -1).max(1)
        good += int((np.minimum(d, d2) < 2.0).sum())
        total += a1 - a0
    assert good / total > 0.97, (good, total)


def test_cfg5_full_size_long_lines():
    from linetr_amd.engine import Engine
    hw = (960, 1280)
    eng5 = Engine(synth.calibrated_state_dict(), "cuda:0", image_shape=list(hw))
    cfg = dict(CFG, max_tokens=41)
    lines, cat, off, dd, ds = batch_inputs(4, seed0=7500, n_lines=600, hw=hw, lo=40.0, hi=327.0)
    tb, ld = describe(eng5, cat, off, dd, ds, cfg)
    assert tb.N == 4 * 599 and torch.isfinite(ld).all()
    assert (ld.norm(dim=1) - 1).abs().max().item() < 1e-5
    tb1, ld1 = describe(eng5, lines[2], np.array([0, 600], np.int32), dd[2:3], ds[2:3], cfg)
    assert (ld1 - ld[tb.cu_n[2]:tb.cu_n[3]]).abs().max().item() < 2e-6
    # fused path == dense reference-layout path at full size
    recs, cu_k, cu_n = eng5.prefilter(cat, hw[0], hw[1], offsets=off, **cfg)
    tbd = eng5.tokenize(recs, cu_k, cu_n, dd, ds, token_distance=8, max_tokens=41)
    ldd = eng5.forward(tbd)
    assert (ld - ldd).abs().max().item() < 5e-6
    assert tbd.mask.sum().item() == tbd.N + int(recs["n_tok"].sum())     # tokeniser invariant: 1 CLS + real tokens


def test_precision_modes_agree(eng):
    lines, cat, off, dd, ds = batch_inputs(8, seed0=7600)
    out = {}
    for mode in ("f32", "bf16x6", "bf16x3", "f16x3"):
        eng.set_precision(mode)
        out[mode] = describe(eng, cat, off, dd, ds)[1].clone()
    eng.set_precision("bf16x6")
    assert (out["bf16x6"] - out["f32"]).abs().max().item() < 3e-6      # fp32-faithful
    assert (out["bf16x3"] - out["f32"]).abs().max().item() < 5e-5      # inside the 1e-4 budget
    assert (out["f16x3"] - out["f32"]).abs().max().item() < 6e-6       # fp32-class (2^-22 per product)


def test_cfg3_step_runs_on_the_intended_kernels(eng):
    """Dispatcher guard (HIP-event profile classes of one cfg3 forward): the 17 K >= 256 / N = 256 GEMMs on the pipelined
    128x256 tile with the three row normalisations fused into their epilogues, the FFN's w_1 on the eight-wave 128x128 tile,
    the word and the line encoder's four MLP layers in one launch, the seven q/k/v projections inside the fused projection + attention kernel, pooling on its
    one-pass kernel, nothing on a fallback tile.  (A
    dispatcher rule lost in an edit once moved the 18 launches to the 64x256 tile: correct results, 12 % slower step.)"""
    _, cat, off, dd, ds = batch_inputs(128)
    describe(eng, cat, off, dd, ds)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    try:
        describe(eng, cat, off, dd, ds)
        torch.cuda.synchronize()
        prof = {e["name"]: e["calls"] for e in eng.get_profile()}
    finally:
        eng.set_profiling(False)
    assert prof.get("gemm_bf16x6_128x256") == 17, prof
    assert prof.get("gemm_bf16x6_128x128s") == 1, prof                                              # the FFN's w_1
    # both positional encoders: layers 1-4 of both in ONE launch (every persistent block walks word tiles, then line tiles)
    assert prof.get("pos_mlp_dual_bf16x6") == 1 and not [k for k in prof if k.startswith("mlp123") or k in ("tok_mlp_bf16x6", "line_mlp_bf16x6")], prof
    assert prof.get("sig_qkv_attn_bf16x6") == 7 and "sig_attn_bf16x6" not in prof and prof.get("cls_pool_online") == 1, prof
    assert "row_norm" not in prof, prof
    assert not [k for k in prof if k.startswith("gemm_") and k not in ("gemm_bf16x6_128x256", "gemm_bf16x6_128x128s", "gemm_bf16x6_128x64",
                                                                "gemm_bf16x6_ws64x256")], prof


def test_matcher_beyond_the_lds_segment_table():
    """An image with more key-points than fit the matcher's LDS segment table (12 000): the reference has no limit
    (max_keypoints = -1, models/superpoint.py; nn_matcher.py:33-42), the kernels switch to a table in the workspace.
    Both entry points against the oracle's NumPy restatement."""
    from linetr_amd import nn_matcher as NM
    from oracle import linetr_oracle as O
    rs = np.random.RandomState(11)
    n0, n1 = 70, 12001
    d0 = rs.standard_normal((256, n0)).astype(np.float32)
    d1 = rs.standard_normal((256, n1)).astype(np.float32)
    d0 /= np.linalg.norm(d0, axis=0, keepdims=True)
    d1 /= np.linalg.norm(d1, axis=0, keepdims=True)
    d1[:, 11990:12001] = d0[:, 5:16]                      # planted mutual matches in the last columns
    mat, dist = NM.nn_matcher(d0, d1, 0.8, True)
    _, want_d = O.point_nn(d0, d1, 0.8, True)
    assert np.abs(dist - want_d).max() < 1e-5
    assert np.array_equal(mat, O.mutual_nn(dist, 0.8, True))      # same argmin rules on the SAME float32 distances
    assert mat[0][5:16, 11990:12001].trace() == 11
    assert np.array_equal(NM.nn_matcher_distmat(dist, 0.8, True), O.mutual_nn(dist, 0.8, True))


def test_matcher_segment_table_in_workspace_with_several_pairs_and_pooling(eng):
    """Three pairs in ONE linetr_match call whose side-1 images have more key-lines than the LDS segment table holds (one pair below
    the limit rides along): the table of every pair is built once by pair_seg1_kernel in that pair's own scratch region and only
    read by the pooling blocks.  Sub-lines are pooled in runs of 1-3 on both sides; against the oracle's matcher, pair by pair."""
    from oracle import linetr_oracle as O
    rs = np.random.RandomState(5)
    dims, d0s, d1s, s0s, s1s = [], [], [], [], []
    for n0, k1 in ((40, 12003), (25, 300), (33, 12500)):
        runs1 = rs.randint(1, 3, k1)                              # 1-2 sub-lines per key-line on side 1
        n1 = int(runs1.sum())
        runs0 = rs.randint(1, 4, n0 // 2)
        k0, n0 = len(runs0), int(runs0.sum())
        d0 = rs.standard_normal((n0, 256)).astype(np.float32)
        d1 = rs.standard_normal((n1, 256)).astype(np.float32)
        d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
        d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
        dims.append((n0, k0, n1, k1))
        d0s.append(d0); d1s.append(d1)
        s0s.append(np.repeat(np.arange(k0), runs0).astype(np.int32)); s1s.append(np.repeat(np.arange(k1), runs1).astype(np.int32))
    cu = lambda v: np.concatenate([[0], np.cumsum(v)]).astype(np.int64)
    n0s, k0s, n1s, k1s = (np.array([d[i] for d in dims]) for i in range(4))
    dk, off_dk, m01 = eng.match(torch.from_numpy(np.concatenate(d0s)).cuda(), cu(n0s), torch.from_numpy(np.concatenate(s0s)).cuda(), cu(k0s),
                                torch.from_numpy(np.concatenate(d1s)).cuda(), cu(n1s), torch.from_numpy(np.concatenate(s1s)).cuda(), cu(k1s), 0.8, True)
    torch.cuda.synchronize()
    dk, m01, ck0 = dk.cpu().numpy(), m01.cpu().numpy(), cu(k0s)

    for p, (n0, k0, n1, k1) in enumerate(dims):
        D = np.clip(2.0 - 2.0 * (d0s[p].astype(np.float64) @ d1s[p].astype(np.float64).T), 0, None)     # models/line_process.py:198-201
        st0, st1 = np.flatnonzero(np.r_[1, np.diff(s0s[p])]), np.flatnonzero(np.r_[1, np.diff(s1s[p])])
        Dk = np.add.reduceat(np.add.reduceat(D, st0, axis=0), st1, axis=1) / np.bincount(s0s[p])[:, None] / np.bincount(s1s[p])[None]
        got_dk = dk[off_dk[p]:off_dk[p + 1]].reshape(k0, k1)
        assert np.abs(got_dk - Dk).max() < 1e-5                  # subline2keyline with 1 / num_sublines rows (models/line_transformer.py:277-282)
        want = O.mutual_nn(got_dk[None], 0.8, True)[0]           # same argmin rules on the SAME float32 distances
        got = np.zeros_like(want)
        mm = m01[ck0[p]:ck0[p + 1]]
        got[np.nonzero(mm >= 0)[0], mm[mm >= 0]] = 1
        assert np.array_equal(got, want), p


def test_matcher_fuzz_ties_and_threshold_edges():
    """nn_matcher_distmat (models/nn_matcher.py:3-31) on 300 random distance matrices drawn from a handful of values, so that ties on
    rows AND columns, entries exactly at the threshold and all-rejected rows are the norm: first-index argmin, strict `<`, mutual
    check -- identical by index to the oracle's NumPy restatement; mutual and one-sided."""
    from linetr_amd import nn_matcher as NM
    from oracle import linetr_oracle as O
    rs = np.random.RandomState(2024)
    vals = np.array([0.0, 0.25, 0.5, 0.79999995, 0.8, 0.80000007, 1.0, 1.5, 4.0], np.float32)
    for case in range(300):
        n0, n1 = rs.randint(1, 41), rs.randint(1, 41)
        d = vals[rs.randint(0, len(vals), (1, n0, n1))]
        if case % 7 == 0:
            d[0, rs.randint(0, n0)] = 4.0                        # a row nothing matches
        mutual = bool(case % 2)
        thr = float(np.float32(0.8))
        got = NM.nn_matcher_distmat(d, thr, mutual)
        want = O.mutual_nn(d, thr, mutual)
        assert got.dtype == np.float64 and np.array_equal(got, want), (case, n0, n1, mutual)
    # a float64 matrix is compared in float64, like NumPy compares it (linetr_match_distmat_f64): distances that differ below
    # float32 resolution, and a threshold between two float64 neighbours of 0.8, must come out as the reference says
    for case in range(100):
        n0, n1 = rs.randint(1, 41), rs.randint(1, 41)
        d64 = vals.astype(np.float64)[rs.randint(0, len(vals), (1, n0, n1))] + rs.randint(-2, 3, (1, n0, n1)) * 1e-12
        d64[d64 < 0] = -1e-12                                      # negatives clip to 0: ties at zero, first index
        mutual = bool(case % 2)
        for thr in (0.8, 0.8 + 1e-12):
            got = NM.nn_matcher_distmat(d64, thr, mutual)
            want = O.mutual_nn(d64, thr, mutual)
            assert np.array_equal(got, want), (case, n0, n1, mutual, thr)


def test_async_host_tickets_deliver_their_own_data_and_refuse_stale_ones():
    """Engine.to_host_async / collect (the drop-in path's result transport): tickets collected out of order give their own
    tensors; one whose staging buffer has been handed out again raises instead of returning another call's data."""
    from linetr_amd.engine import Engine
    eng = Engine.heads_only("cuda:0")
    ts = [torch.full((257, 3), float(i), device="cuda") for i in range(4)]
    idx = [torch.arange(5, device="cuda", dtype=torch.int32) + i for i in range(4)]
    tickets = [eng.to_host_async(ts[i], idx[i]) for i in range(4)]
    for i in (2, 0, 3, 1):
        a, b = eng.collect(tickets[i])
        assert a.dtype == np.float32 and (a == i).all() and np.array_equal(b, np.arange(5) + i)
    old = eng.to_host_async(ts[0])
    for _ in range(4):
        eng.to_host_async(ts[1])
    with pytest.raises(RuntimeError, match="staging buffer has been reused"):
        eng.collect(old)
    (h,) = eng.to_host(ts[3])
    assert (h == 3).all()


def test_near_tie_contract(eng):
    """"Line matches bit-exact by index" and its limit (bench.py ARGMIN_CONTRACT): the matcher's argmin equals the float64 answer
    wherever the best-vs-second-best margin exceeds 4 x the measured distance error; a tie built ON PURPOSE below that error (two
    candidate lines whose descriptors differ by ~1e-7) may come out as either index -- both are correct fp32 results, the
    reference's own BLAS could flip it as well -- and never as anything else."""
    rs = np.random.RandomState(11)
    n0, n1 = 150, 151
    d0 = rs.standard_normal((n0, 256)); d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    d1 = d0[rs.permutation(n0)] + 0.15 * rs.standard_normal((n0, 256))
    d1 = np.concatenate([d1, d1[:1]])                       # candidate 150 = candidate 0 ...
    d1[150] += 2e-7 * rs.standard_normal(256)               # ... moved by less than an fp32 ulp of a unit vector's entries
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    d0f, d1f = d0.astype(np.float32), d1.astype(np.float32)
    iota0 = torch.arange(n0, dtype=torch.int32, device="cuda")
    iota1 = torch.arange(n1, dtype=torch.int32, device="cuda")
    dk, _, m01 = eng.match(torch.from_numpy(d0f).cuda(), np.array([0, n0]), iota0, np.array([0, n0]), torch.from_numpy(d1f).cuda(),
                           np.array([0, n1]), iota1, np.array([0, n1]), 10.0, False)      # one-way, no threshold: the raw row argmin
    got = m01.cpu().numpy()
    exact = np.clip(2.0 - 2.0 * d0f.astype(np.float64) @ d1f.astype(np.float64).T, 0, None)
    err = float(np.abs(dk.cpu().numpy().reshape(n0, n1) - exact).max())
    assert err < 2e-6
    srt = np.sort(exact, axis=1)
    margin = srt[:, 1] - srt[:, 0]
    best = exact.argmin(1)
    safe = margin > 4 * err
    assert safe.sum() >= n0 - 2 and np.array_equal(got[safe], best[safe])          # by arithmetic
    tied = np.nonzero(~safe)[0]
    assert len(tied) >= 1                                                           # the constructed tie is in there
    for i in tied:                                                                  # either index of the tie, nothing else
        assert exact[i, got[i]] - exact[i, best[i]] <= 4 * err, (i, got[i], best[i])
    i_tie = int(np.nonzero(best == 0)[0][0]) if (best == 0).any() else int(np.nonzero(best == 150)[0][0])
    assert got[i_tie] in (0, 150)
    # deterministic: the same call gives the same index every time
    for _ in range(5):
        assert np.array_equal(eng.match(torch.from_numpy(d0f).cuda(), np.array([0, n0]), iota0, np.array([0, n0]),
                                        torch.from_numpy(d1f).cuda(), np.array([0, n1]), iota1, np.array([0, n1]), 10.0, False)[2].cpu().numpy(), got)
