"""Descriptor / match parity soak: P random pairs (random line counts, seeds, both dense layouts) described as ONE batch on the GPU and
matched, against the CPU oracle pair by pair: max |descriptor - oracle|, line matches identical by index, smallest argmin margin.
    python tools/parity_soak.py [pairs] [cfg3|cfg5]      (on the GPU box; cfg5 = the long-line workload's shape: 1280 x 960, 300-600 lines of 40-327 px, 41 tokens)
    python tools/parity_soak.py asset                    the fixture with real-image statistics (tests/golden/asset_pair.npz, made by the real reference):
                                                         descriptor / Dk errors, line and point argmin margins, margins at risk"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linetr_amd.engine import Engine  # noqa: E402
from oracle import linetr_oracle as O  # noqa: E402
from workloads import synth  # noqa: E402


def asset_report():
    g = np.load(os.path.join(ROOT, "tests", "golden", "asset_pair.npz"))
    torch.set_grad_enabled(False)
    eng = Engine(synth.calibrated_state_dict(), "cuda:0")
    dd = torch.cat([torch.from_numpy(g["dense_descriptor" + s]) for s in "01"]).cuda()
    ds = torch.cat([torch.from_numpy(g["dense_score" + s]) for s in "01"]).cuda()
    lines = [g["lines0"], g["lines1"]]
    off = np.array([0, len(lines[0]), len(lines[0]) + len(lines[1])], np.int32)
    tb, ld = eng.describe_lines(np.concatenate(lines), off, dd, ds, remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21)
    n, k = np.diff(tb.cu_n), np.diff(tb.cu_k)
    dk, _off, m01 = eng.match(ld[:n[0]], np.array([0, n[0]]), tb.sub2line[:n[0]], np.array([0, k[0]]), ld[n[0]:], np.array([0, n[1]]),
                              tb.sub2line[n[0]:], np.array([0, k[1]]), 0.8, True)
    dist, pm01 = eng.match_points(torch.from_numpy(g["descriptors0"]).cuda(), torch.from_numpy(g["descriptors1"]).cuda(), 0.7, True)
    torch.cuda.synchronize()
    ld_c = ld.cpu().numpy()
    e_desc = max(float(np.abs(ld_c[tb.cu_n[i]:tb.cu_n[i + 1]].T - g["line_desc" + s][0]).max()) for i, s in enumerate("01"))
    Dk = g["matching_scores_l"][0]
    e_dk = float(np.abs(dk.cpu().numpy().reshape(Dk.shape) - Dk).max())
    got = np.zeros_like(Dk, dtype=np.float64)
    mm = m01.cpu().numpy()
    got[np.nonzero(mm >= 0)[0], mm[mm >= 0]] = 1
    marg = lambda d: np.concatenate([np.diff(np.sort(d, axis=1)[:, :2], axis=1)[:, 0], np.diff(np.sort(d, axis=0)[:2, :], axis=0)[0]])
    ml = marg(Dk)
    d64 = np.clip(2.0 - 2.0 * g["descriptors0"].astype(np.float64).T @ g["descriptors1"].astype(np.float64), 0, None)
    mp = marg(d64)
    dist_c = dist.cpu().numpy()
    e_p = max(float(np.abs(dist_c.min(1) - g["matching_scores_p_rowmin"]).max()), float(np.abs(dist_c.min(0) - g["matching_scores_p_colmin"]).max()))
    p_got, p_want = pm01.cpu().numpy(), g["matches_p_index"]
    dd0 = g["dense_descriptor0"][0]
    print(f"asset pair (scannet_0a / 0b through the reference's SuperPoint, seeded weights; neighbouring descriptor cells at cosine "
          f"{float((dd0[:, :, 1:] * dd0[:, :, :-1]).sum(0).mean()):.3f}): {int(tb.K)} key-lines, {int(tb.N)} sub-lines")
    print(f"  lines : max |desc - reference| = {e_desc:.2e}, max |Dk - reference| = {e_dk:.2e}, match matrix identical: {np.array_equal(got, g['matches_l'][0])} "
          f"({int(got.sum())} matches); argmin margins: smallest {ml.min():.2e}, below 1e-5: {int((ml < 1e-5).sum())}, at risk (< 4 x Dk error): {int((ml < 4 * e_dk).sum())} of {ml.size}")
    print(f"  points: max |distance - reference| = {e_p:.2e}, rows whose match differs from the reference's: {int((p_got != p_want).sum())} of {len(p_want)}; "
          f"argmin margins: smallest {mp.min():.1e}, below 1e-6: {int((mp < 1e-6).sum())}, at risk (< 4 x max(error, 2.5e-7)): {int((mp < 4 * max(e_p, 2.5e-7)).sum())} of {mp.size}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "asset":
        return asset_report()
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    big = len(sys.argv) > 2 and sys.argv[2] == "cfg5"
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    hw = (960, 1280) if big else (480, 640)
    cfg = dict(min_length=16, token_distance=8, max_tokens=41 if big else 21, remove_borders=8, max_keylines=-1)
    sdn = synth.calibrated_state_dict()
    sd = synth.to_torch_state_dict(sdn)
    eng = Engine(sdn, "cuda:0", image_shape=list(hw))
    rs = np.random.RandomState(2026)
    lines, maps = [], []
    for i in range(2 * P):
        seed = 70000 + i
        lines.append(synth.synth_lines(seed, int(rs.randint(300, 601)), *hw, 40.0, 327.0) if big else synth.synth_lines(seed, int(rs.randint(12, 320)), *hw))
        maps.append(synth.synth_dense_maps(seed, *hw))
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)
    kw = dict(remove_borders=8, min_length=cfg["min_length"], max_keylines=-1, token_distance=cfg["token_distance"], max_tokens=cfg["max_tokens"])
    tb, ld = eng.describe_lines(np.concatenate(lines), off, dd, ds, **kw)
    tb2, ld2 = eng.describe_lines(np.concatenate(lines), off, dd.permute(0, 2, 3, 1).contiguous(), ds, dense_layout="nhwc", **kw)
    n, k = np.diff(tb.cu_n), np.diff(tb.cu_k)
    cs = lambda v: np.concatenate([[0], np.cumsum(v)]).astype(np.int64)
    i0 = torch.cat([torch.arange(tb.cu_n[i], tb.cu_n[i + 1]) for i in range(0, 2 * P, 2)]).cuda()
    i1 = torch.cat([torch.arange(tb.cu_n[i], tb.cu_n[i + 1]) for i in range(1, 2 * P, 2)]).cuda()
    dk, off_dk, m01 = eng.match(ld[i0], cs(n[0::2]), tb.sub2line[i0], cs(k[0::2]), ld[i1], cs(n[1::2]), tb.sub2line[i1], cs(k[1::2]), 0.8, True)
    torch.cuda.synchronize()
    ld_c, m01_c, dk_c, ck0 = ld.cpu().numpy(), m01.cpu().numpy(), dk.cpu().numpy(), cs(k[0::2])
    worst = worst_dk = 0.0
    bad_pairs, margin, n_match, t0 = 0, np.inf, 0, time.time()
    # r05: every pair once more on its own -- the ONE-launch single-pair matcher (pair_match_fused_kernel) where it applies --, which must
    # give the batched launches' matches
    bad_single = fused = 0
    cn, ck = tb.cu_n.astype(np.int64), tb.cu_k.astype(np.int64)
    for p in range(P):
        a, b = 2 * p, 2 * p + 1
        one = eng.match(ld[cn[a]:cn[a + 1]], np.array([0, n[a]]), tb.sub2line[cn[a]:cn[a + 1]], np.array([0, k[a]]),
                        ld[cn[b]:cn[b + 1]], np.array([0, n[b]]), tb.sub2line[cn[b]:cn[b + 1]], np.array([0, k[b]]), 0.8, True)
        fused += int(n[b] <= 1024 and k[a] <= 4096 and k[b] <= 4096 and k[a] > 0 and k[b] > 0)
        bad_single += int(not np.array_equal(one[2].cpu().numpy(), m01_c[ck0[p]:ck0[p] + int(k[a])]))
    for p in range(P):
        outs = []
        for s in range(2):
            i = 2 * p + s
            o = O.preprocess(synth.array_to_keylines(lines[i]), (1, 1, *hw), maps[i][0], maps[i][1], cfg)
            o = O.forward(sd, o, hw)
            worst = max(worst, float(np.abs(ld_c[tb.cu_n[i]:tb.cu_n[i + 1]].T - o["line_desc"][0].numpy()).max()))
            outs.append(o)
        M, Dk = O.match_lines(outs[0]["line_desc"], outs[1]["line_desc"], outs[0]["mat_klines2sublines"][0], outs[1]["mat_klines2sublines"][0], 0.8)
        K0, K1 = M.shape[1], M.shape[2]
        got = np.zeros_like(M[0])
        mm = m01_c[ck0[p]:ck0[p] + K0]
        got[np.nonzero(mm >= 0)[0], mm[mm >= 0]] = 1
        bad_pairs += int(not np.array_equal(got, M[0]))
        n_match += int(got.sum())
        worst_dk = max(worst_dk, float(np.abs(dk_c[off_dk[p]:off_dk[p + 1]].reshape(K0, K1) - Dk[0]).max()))
        for d in (Dk[0], Dk[0].T):
            if d.shape[1] >= 2:
                two = np.partition(d, 1, axis=1)[:, :2]
                margin = min(margin, float((two[:, 1] - two[:, 0]).min()))
    print(f"{P} pairs, {int(tb.N)} descriptors, {n_match} line matches: max |desc - oracle| = {worst:.2e}, max |Dk - oracle| = {worst_dk:.2e}, "
          f"pairs with a differing match matrix = {bad_pairs}, smallest argmin margin = {margin:.2e}, NHWC-fed vs NCHW-fed descriptors "
          f"max diff = {(ld - ld2).abs().max().item():.1e}; matched pair by pair ({fused} of {P} through the one-launch matcher): "
          f"{bad_single} pairs differ from the batched matcher  (oracle time {time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
