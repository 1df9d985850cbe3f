"""Time linetr_superpoint_heads on a cfg3-sized batch (128 images of 480x640): python tools/producer_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from linetr_amd.engine import Engine
eng = Engine.heads_only('cuda:0')
B, Hc, Wc = 128, 60, 80
sl = torch.randn(B, 65, Hc, Wc, device='cuda') * 2; dr = torch.randn(B, 256, Hc, Wc, device='cuda')
def timed(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
HW = Hc * Wc
for label, fn, nbytes in (("desc -> nhwc", lambda: eng.superpoint_heads(None, dr, nhwc=True), B * HW * 256 * 2 * 4),
                          ("desc -> nhwc + nchw", lambda: eng.superpoint_heads(None, dr, nhwc=True, nchw=True), B * HW * 256 * 3 * 4),
                          ("score", lambda: eng.superpoint_heads(sl, None), B * HW * 129 * 4),
                          ("both heads -> score + nhwc", lambda: eng.superpoint_heads(sl, dr, nhwc=True), B * HW * (512 + 129) * 4)):
    ms = timed(fn)
    print(f"{label:28s} {ms*1e3:8.1f} us  {nbytes/ms/1e9:7.2f} TB/s")
