"""Where does a wave of the pipelined split GEMM spend a K tile?  Needs the debug build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DLT_GEMM_TIMING -Iinclude \\
          -o /tmp/liblinetr_timing.so linetr_amd/csrc/linetr_hip.hip      (tools/build_timing_lib.sh)
    LINETR_LIB=.../liblinetr_timing.so python tools/gemm_phase_timing.py bf16x6 8192 4096 4096
s_memtime stamps of block 0, all waves, K tiles 8..23; printed as the mean share of each phase."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from linetr_amd import synth, _native
from linetr_amd.engine import Engine
mode = sys.argv[1]; M, N, K = map(int, sys.argv[2:5])
eng = Engine(synth.make_state_dict(0), 'cuda:0'); eng.set_precision(mode)
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda')
L = _native.lib()
def timed(label):
    for _ in range(5): eng.debug_gemm(A, W, cache_weights=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.debug_gemm(A, W, cache_weights=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{label}: {us:.1f} us/GEMM, {2 * M * N * K / us / 1e6:.1f} TF")
timed(os.environ.get("LABEL", "GEMM"))

if not hasattr(L, "linetr_debug_read_stamps"): sys.exit(0)
buf = np.zeros(8 * 16 * 8, dtype=np.uint64)
L.linetr_debug_read_stamps.argtypes = [C.c_void_p]
assert L.linetr_debug_read_stamps(buf.ctypes.data) == 0
t = buf.reshape(8, 16, 8)[:, :, :6].astype(np.int64)
names = ["-", "first half: MFMA step 0 | ds_read step 1 | split + ds_write", "barrier", "-", "second half: MFMA step 1 | global loads | ds_read step 0"]
period = (t[:, 1:, 0] - t[:, :-1, 0]).mean()
print(f"{mode} M={M} N={N} K={K}: K-tile period {period:.0f} ticks of s_memrealtime (100 MHz: 10 ns each)")
for i, n in enumerate(names):
    d = (t[:, :, i + 1] - t[:, :, i]).mean()
    print(f"  {n:40s} {d:8.1f} ticks  {100 * d / period:5.1f} %")
d = (t[:, 1:, 0] - t[:, :-1, 5]).mean()
print(f"  {'loop back-edge':40s} {d:8.1f} ticks  {100 * d / period:5.1f} %")
for w in range(8):
    print("  wave", w, "iteration 12 stamps:", (t[w, 4] - t[0, 4, 0]).tolist())
