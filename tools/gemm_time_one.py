"""Time one GEMM shape through linetr_debug_gemm with HIP events (used by tools/gemm_skip_sweep.sh, which points
LINETR_LIB at its ablation builds):   python tools/gemm_time_one.py bf16x6 8192 4096 4096     (LABEL=... names the line)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from workloads import synth
from linetr_amd.engine import Engine
mode = sys.argv[1]; M, N, K = map(int, sys.argv[2:5])
eng = Engine(synth.make_state_dict(0), 'cuda:0'); eng.set_precision(mode)
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda')
for _ in range(5): eng.debug_gemm(A, W, cache_weights=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): eng.debug_gemm(A, W, cache_weights=True)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"{os.environ.get('LABEL', 'GEMM')}: {us:.1f} us/GEMM, {2 * M * N * K / us / 1e6:.1f} TF")
