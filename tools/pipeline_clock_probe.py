"""Shader clock and package power while the cfg3 step runs on ONE stream and as the describe pipeline (rocm-smi sampled from a side
thread, as tools/gemm_clock_probe.py does for a single GEMM).  Answers: does the pipeline pay for its occupancy in clock?
usage (GPU box): python tools/pipeline_clock_probe.py [seconds]"""
import os
import re
import statistics
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402
from workloads import synth  # noqa: E402
from linetr_amd.engine import Engine  # noqa: E402

secs = float(argv[1]) if len(argv) > 1 else 5.0
dev = torch.device("cuda:0")
H, W, n_lines, lo, hi, T, pairs = bench.WORKLOADS["cfg3"]
eng = Engine(synth.calibrated_state_dict(), dev, image_shape=[H, W])
lines, dd, nhwc, ds, hw, T = bench.make_inputs("cfg3", pairs, 0, dev, eng)


def probe(label, depth):
    pipe = bench.Pipeline(eng, lines, dd, ds, hw, T, 1, pairs, "nchw", pipelined=depth)
    samples, stop = [], [False]

    def poll():
        while not stop[0]:
            try:
                o = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
                p = re.search(r"Power \(W\): ([\d.]+)", o)
                samples.append((int(s.group(1)) if s else None, float(p.group(1)) if p else None))
            except Exception:
                samples.append((None, None))
    for _ in range(100):
        pipe.step()
    pipe.drain(); torch.cuda.synchronize()
    th = threading.Thread(target=poll)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50):
            pipe.step()
        n += 50
    pipe.drain(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    stop[0] = True
    th.join()
    clk = [c for c, _ in samples[1:] if c]
    pw = [p for _, p in samples[1:] if p]
    tc, tp = clk[len(clk) // 2:] or [0], pw[len(pw) // 2:] or [0]
    print(f"{label:28s} {dt * 1e3:.4f} ms/step | sclk median {statistics.median(tc):.0f} MHz (min {min(tc)}, max {max(tc)}) | power median "
          f"{statistics.median(tp):.0f} W (max {max(tp):.0f}) | {len(clk)} samples", flush=True)


for rep in range(2):
    probe("one stream", 0)
    probe("pipeline, 2 in flight", 2)
    probe("pipeline, 3 in flight", 3)
