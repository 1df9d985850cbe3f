"""Soak of the describe pipeline (linetr_describe_submit / _join): a long stream of batches of RANDOM sizes (2 .. 128 images, 3 .. 320 lines
each, both dense layouts, an empty batch now and then) through DescribePipeline at depth 2 / 3 / 4, every result compared bit for bit with the
plain call on the same inputs, plus a stall watch on the steady cfg3 stream (largest completion interval against the median).
    python tools/pipeline_soak.py [seconds per depth]      (GPU box)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linetr_amd.engine import DescribePipeline, Engine  # noqa: E402
from workloads import synth  # noqa: E402

HW = (480, 640)
CFG = dict(remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    torch.set_grad_enabled(False)
    eng = Engine(synth.calibrated_state_dict(), "cuda:0")
    rs = np.random.RandomState(606)
    # a pool of images to draw batches from (maps on the device in both layouts)
    n_pool = 160
    lines = [synth.synth_lines(80000 + i, int(rs.randint(3, 321)), *HW) for i in range(n_pool)]
    maps = [synth.synth_dense_maps(80000 + i, *HW) for i in range(n_pool)]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    nhwc = dd.permute(0, 2, 3, 1).contiguous()

    def draw():
        if rs.rand() < 0.03:
            b = int(rs.randint(1, 9))
            return np.zeros((0, 6)), np.zeros(b + 1, np.int32), dd[:b], ds[:b], "nchw"
        b = int(rs.choice([2, 2, 4, 8, 16, 32, 64, 128]))
        i0 = int(rs.randint(0, n_pool - b + 1))
        ls = lines[i0:i0 + b]
        off = np.concatenate([[0], np.cumsum([len(l) for l in ls])]).astype(np.int32)
        layout = "nhwc" if rs.rand() < 0.5 else "nchw"
        return np.concatenate(ls), off, (nhwc if layout == "nhwc" else dd)[i0:i0 + b], ds[i0:i0 + b], layout

    for depth in ((2, 3, 4) if secs > 0 else ()):
        pipe = DescribePipeline(eng, depth)
        pending, n_batches, n_desc, bad = [], 0, 0, 0
        t0 = time.time()

        def check(done):
            nonlocal bad, n_desc
            cat, off, d_, s_, layout = pending.pop(0)
            tb_ref, ld_ref = eng.describe_lines(cat, off, d_, s_, dense_layout=layout, **CFG)
            tb, ld = done
            ok = tb.N == tb_ref.N and torch.equal(ld, ld_ref) and torch.equal(tb.sublines, tb_ref.sublines) and torch.equal(tb.sub2line, tb_ref.sub2line)
            bad += int(not ok)
            n_desc += int(tb.N)
        while time.time() - t0 < secs:
            item = draw()
            pending.append(item)
            done = pipe.submit(item[0], item[1], item[2], item[3], dense_layout=item[4], **CFG)
            n_batches += 1
            if done is not None:
                check(done)
        for done in pipe.drain():
            check(done)
        torch.cuda.synchronize()
        print(f"depth {depth}: {n_batches} batches of random size ({n_desc} descriptors) in {time.time() - t0:.0f} s, every batch also described by the "
              f"plain call in between: results differing from linetr_describe's: {bad}", flush=True)

    # stall watch: the steady cfg3 stream, 2000 steps, a completion event per step
    sys.argv = ["bench.py"]
    import bench
    H, W, n_lines, lo, hi, T, pairs = bench.WORKLOADS["cfg3"]
    ls, d3, _n3, s3, hw, T = bench.make_inputs("cfg3", pairs, 0, torch.device("cuda:0"), eng)
    import gc
    for depth, frozen in ((0, False), (3, False), (3, True)):
        if frozen:
            gc.collect()
            gc.freeze()        # (last run: the start-up heap parked in the permanent generation, as bench.py does before it times anything)
        p = bench.Pipeline(eng, ls, d3, s3, hw, T, 1, pairs, "nchw", pipelined=depth)
        for _ in range(100):
            p.step()
        p.drain(); torch.cuda.synchronize()
        n = 2000
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        host = np.zeros(n)
        evs[0].record()
        for i in range(n):
            h0 = time.perf_counter()
            p.step()
            host[i] = (time.perf_counter() - h0) * 1e3
            evs[i + 1].record()
        p.drain(); torch.cuda.synchronize()
        iv = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(max(depth, 1), n)])
        print(f"cfg3 stream, {'one stream' if not depth else str(depth) + ' batches in flight'}{', gc.freeze() after set-up' if frozen else ''}: {n} steps, completion interval median {np.median(iv):.3f} ms, "
              f"p1 {np.percentile(iv, 1):.3f}, p99 {np.percentile(iv, 99):.3f}, max {iv.max():.3f} ms; intervals above 1.5 x the median: {int((iv > 1.5 * np.median(iv)).sum())}", flush=True)
        if depth:      # where do the long intervals come from?  the host time of the step calls around each of the five longest
            for j in np.argsort(-iv)[:5]:
                k = j + max(depth, 1)
                print(f"    interval {iv[j]:.2f} ms at step {k}: host ms of steps {k - 3} .. {k}: " + ", ".join(f"{host[q]:.2f}" for q in range(max(k - 3, 0), k + 1))
                      + f"; next intervals {', '.join(f'{v:.2f}' for v in iv[j + 1:j + 4])}")
            print(f"    host ms per step call: median {np.median(host):.3f}, p99 {np.percentile(host, 99):.3f}, max {host.max():.3f}")


if __name__ == "__main__":
    main()
