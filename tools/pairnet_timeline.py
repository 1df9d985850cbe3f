"""Where a single-pair persistent signature network launch spends its time (lt_pairnet.h): per stage, when the first block
picked a unit up, when producers were seen, when bodies were done and published -- from device wall-clock stamps (100 MHz).
    python tools/pairnet_timeline.py            (experiments build; on the GPU box)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from linetr_amd import _native as nat  # noqa: E402
from linetr_amd.engine import Engine  # noqa: E402
from workloads import synth  # noqa: E402


def main():
    os.environ["LINETR_PAIRNET"] = "1"
    dev = torch.device("cuda:0")
    eng = Engine(synth.calibrated_state_dict(), dev, lib_path=nat.EXPERIMENTS_LIB_PATH)
    lines, dd, nhwc, ds, hw, T = bench.make_inputs("cfg2", 1, 0, dev, eng)
    pipe = bench.Pipeline(eng, lines, nhwc, ds, hw, T, 1, 1)
    for _ in range(20):
        pipe.describe()
    torch.cuda.synchronize()
    L, G = 7, 256
    S = 4 * L + 1
    buf = torch.zeros((G, S, 8), dtype=torch.int64, device=dev)
    nat.check(eng._L.linetr_debug_pairnet_stamps(eng._h, C.c_void_p(buf.data_ptr())), eng._L)
    pipe.describe()
    torch.cuda.synchronize()
    nat.check(eng._L.linetr_debug_pairnet_stamps(eng._h, None), eng._L)
    st = buf.cpu().numpy().astype(np.float64)
    # is the clock the same on every XCD?  first stamp of every block (block b runs on XCD b % 8; the grid starts within ~0.5 us)
    first = np.where(st[:, :, 0] > 0, st[:, :, 0], np.inf).min(axis=1)
    ok = np.isfinite(first)
    print("first stamp per XCD, us relative to the earliest: " +
          "  ".join(f"{(np.median(first[ok & (np.arange(G) % 8 == x)]) - first[ok].min()) / 100:.2f}" for x in range(8)))
    used = st[:, :, 0] > 0
    t0 = st[:, :, 0][used].min()
    us = lambda v: (v - t0) / 100.0
    names = ["QKV0"] + [f"{k}{l}" for l in range(L) for k in ("ATTN", "W1", "W2" if l < L - 1 else "FINAL", "QKV" if l < L - 1 else "NORM")]
    names = ["QKV0"]
    for l in range(L):
        names += [f"ATTN{l}", f"W1_{l}", f"W2_{l}" if l < L - 1 else "FINAL", f"QKV{l + 1}" if l < L - 1 else "NORM"]
    print(f"{'stage':8s} blocks  first_pick  dep_seen(min/med/max)   body_done(min/med/max)   published(max)   body(med) wait(med)")
    for s in range(S):
        m = used[:, s]
        if not m.any():
            continue
        a, b, c, d = (st[m, s, i] for i in range(4))
        print(f"{names[s]:8s} {int(m.sum()):5d}  {us(a.min()):9.2f}   {us(b.min()):7.2f} {us(np.median(b)):7.2f} {us(b.max()):7.2f}   "
              f"{us(c.min()):7.2f} {us(np.median(c)):7.2f} {us(c.max()):7.2f}   {us(d.max()):9.2f}     {np.median(c - b) / 100:6.2f}  {np.median(b - a) / 100:6.2f}")
    print(f"total {us(st[:, :, 3].max()):.2f} us")
    s = 2
    late = np.argsort(-st[:, s, 1])[:8]
    print("latest dep_seen in W1_0: " + "  ".join(f"b{b}(xcd{b % 8}) {us(st[b, s, 1]):.1f} pick {us(st[b, s, 0]):.1f} prevATTN {'y' if used[b, 1] else 'n'} prevQKV_done {us(st[b, 0, 3]) if used[b, 0] else -1:.1f}" for b in late))
    # inside a unit (median over blocks, us after the dependency was seen): GEMM [4] wave 0's MFMAs done, [5] all partials in LDS;
    # attention [4] V staged, [5] scores, [6] P V done, [7] all waves arrived at the merge
    for s in (1, 2, 3, 4):
        m = used[:, s]
        rel = lambda i: np.median(st[m, s, i] - st[m, s, 1]) / 100
        print(f"{names[s]:8s} after dep_seen: " + "  ".join(f"[{i}] {rel(i):6.2f}" for i in (4, 5, 6, 7, 2, 3) if st[m, s, i].min() > 0))


if __name__ == "__main__":
    main()
