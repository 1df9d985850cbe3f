"""Phase timing of mlp123_kernel (debug build with -DLT_MLP_STAMPS): 100 MHz wall-clock stamps of wave 0 of each block.
   hipcc ... -DLT_MLP_STAMPS -o tools/liblinetr_var_stamps.so ; LINETR_LIB=... python tools/mlp_stamps.py"""
import ctypes as C, os, sys, json, subprocess
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from linetr_amd import _native
from workloads import synth
from linetr_amd.engine import Engine
eng = Engine(synth.calibrated_state_dict(), 'cuda:0')
lines, dd, ds, (H, W), T = bench.make_inputs('cfg3', 64, 0, eng.device)
pipe = bench.Pipeline(eng, lines, dd, ds, (H, W), T, 1, 64)
for _ in range(5): pipe.describe()
torch.cuda.synchronize()
L = _native.lib()
buf = np.zeros(1024 * 16, dtype=np.uint64)
L.linetr_debug_mlp_stamps.argtypes = [C.c_void_p]
assert L.linetr_debug_mlp_stamps(buf.ctypes.data) == 0
t = buf.reshape(1024, 16).astype(np.int64)
names = ["start", "staged (barrier)", "-", "weights gathered / loop entry", "layer 1 done", "layer 2 done", "layer 3 done", "iteration 0 done", "iteration 1 start", "kernel end"]
ok = t[:, 0] > 0
t0 = t[ok, 0].min()
st = np.sort((t[ok, 0] - t0) / 100.0); en = (t[ok, 9] - t0) / 100.0
print("blocks stamped", ok.sum(), " kernel span (us)", en.max(), " block start times (us): p25/p50/p75/max", np.percentile(st, 25), np.percentile(st, 50), np.percentile(st, 75), st.max())
for us in (1, 5, 10, 20, 30, 50, 100, 150, 200):
    print(f"   running at t={us:4d} us: {int(((t[ok,0]-t0)/100.0 <= us).sum() - (en <= us).sum())} blocks")
for a, b, lab in ((0, 1, "stage weights in LDS + barrier"), (1, 3, "gather resident weights + first feature load"), (3, 4, "layer 1 (VALU)"),
                  (4, 5, "layer 2 (32 MFMA)"), (5, 6, "layer 3 (128 MFMA)"), (6, 7, "staging tile + stores"), (0, 9, "whole wave")):
    d = (t[ok, b] - t[ok, a]) / 100.0
    print(f"{lab:48s} median {np.median(d):8.2f} us   p10 {np.percentile(d,10):8.2f}   p90 {np.percentile(d,90):8.2f}")
