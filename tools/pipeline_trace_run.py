#!/usr/bin/env python3
"""Nothing but N steps of the headline step (bench.Pipeline, NCHW-fed like the driver's line) so that a rocprofv3 --kernel-trace of
this process ends in the steady state of the schedule under test:   python tools/pipeline_trace_run.py cfg3 3 [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402
from workloads import synth  # noqa: E402
from linetr_amd.engine import Engine  # noqa: E402

wl = argv[1] if len(argv) > 1 else "cfg3"
depth = int(argv[2]) if len(argv) > 2 else 2
steps = int(argv[3]) if len(argv) > 3 else 60
H, W, n_lines, lo, hi, T, pairs = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
eng = Engine(synth.calibrated_state_dict(), dev, image_shape=[H, W])
lines, dd, nhwc, ds, hw, T = bench.make_inputs(wl, pairs, 0, dev, eng)
pipe = bench.Pipeline(eng, lines, dd if wl == "cfg3" else nhwc, ds, hw, T, 1, pairs, "nchw" if wl == "cfg3" else "nhwc", pipelined=depth)
for _ in range(steps):
    pipe.step()
pipe.drain()
torch.cuda.synchronize()
print("done", wl, depth, steps)
