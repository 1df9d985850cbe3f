"""Split-tile GEMM (lt_gemm_st.h): numerics against float64 and timing against the register-staged split GEMM on the
signature network's shapes.      python tools/st_probe.py [--no-check]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
# tuning switches and the non-shipped kernels live in the experiments build of the library
os.environ.setdefault("LINETR_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments", "liblinetr_hip_experiments.so"))
from workloads import synth
from linetr_amd.engine import Engine

eng = Engine(synth.make_state_dict(0), "cuda:0")
g = torch.Generator(device="cuda").manual_seed(3)


def rnd(*shape):
    return torch.randn(*shape, device="cuda", generator=g)


def check(M, N, K1, K2=0, bias=False, res=False, act=0, st_out=True):
    A1, A2 = rnd(M, K1), (rnd(M, K2) if K2 else None)
    W = rnd(N, K1 + K2) / (K1 + K2) ** 0.5
    b = rnd(N) if bias else None
    R = rnd(M, N) if res else None
    A = torch.cat([A1, A2], 1) if K2 else A1
    want = A.double() @ W.double().T
    if bias:
        want += b.double()
    if act == 1:
        want = want.clamp_min(0)
    if res:
        want += R.double()
    a1s, a2s, ws = eng.to_st(A1), (eng.to_st(A2) if K2 else None), eng.to_st(W)
    rs = eng.to_st(R) if res else None
    if st_out:
        out = torch.zeros(int(eng._L.linetr_st_bytes(M, N)), dtype=torch.uint8, device="cuda")
        eng.gemm_st(a1s, K1, ws, M, N, A2=a2s, K2=K2, bias=b, residual=rs, act=act, out_st=out)
        got = eng.from_st(out, M, N)
    else:
        got = torch.full((M, N), float("nan"), device="cuda")
        eng.gemm_st(a1s, K1, ws, M, N, A2=a2s, K2=K2, bias=b, residual=rs, act=act, out=got)
    torch.cuda.synchronize()
    err = (got.double() - want).abs().max().item()
    scale = want.abs().max().item()
    print(f"check M={M} N={N} K={K1}+{K2} bias={bias} res={res} act={act} st_out={st_out}: max err {err:.2e} (scale {scale:.2f})",
          flush=True)
    assert err < 2e-6 * max(scale, 1.0), err


def time_it(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


if "--no-check" not in sys.argv:
    X = rnd(300, 96)
    st = eng.to_st(X)
    assert torch.equal(eng.from_st(st, 300, 96), X), "ST round trip is not exact"
    print("ST round trip exact", flush=True)
    check(128, 256, 32, st_out=False)
    check(300, 256, 64, st_out=False)
    check(300, 256, 64, bias=True, st_out=True)
    check(1000, 512, 256, K2=256, bias=True, act=1)
    check(777, 256, 512, bias=True, res=True)
    check(2000, 768, 256, bias=True)
    check(515, 256, 256, K2=512, bias=True, st_out=False)

keep = []
for (M, N, K1, K2, res, act, st_out, label) in [
        (25472, 768, 256, 0, False, 0, True, "qkv"), (25472, 512, 256, 256, False, 1, True, "W1"),
        (25472, 256, 512, 0, True, 0, True, "W2+res"), (25472, 256, 256, 512, False, 0, False, "final(f32 out)"),
        (9584, 768, 256, 0, False, 0, True, "cfg5 qkv"), (9584, 512, 256, 256, False, 1, True, "cfg5 W1"),
        (76672, 768, 256, 0, False, 0, True, "cfg5x8 qkv"), (32768, 512, 256, 256, False, 1, True, "2 rounds W1")]:
    K = K1 + K2
    A1, A2 = rnd(M, K1), (rnd(M, K2) if K2 else None)
    W, b = rnd(N, K) / K ** 0.5, rnd(N)
    R = rnd(M, N) if res else None
    keep.append(W)
    a1s, a2s, ws = eng.to_st(A1), (eng.to_st(A2) if K2 else None), eng.to_st(W)
    rs = eng.to_st(R) if res else None
    out_st = torch.zeros(int(eng._L.linetr_st_bytes(M, N)), dtype=torch.uint8, device="cuda") if st_out else None
    out = None if st_out else torch.empty((M, N), device="cuda")
    us = time_it(lambda: eng.gemm_st(a1s, K1, ws, M, N, A2=a2s, K2=K2, bias=b, residual=rs, act=act, out_st=out_st, out=out))
    A = torch.cat([A1, A2], 1) if K2 else A1
    Y = torch.empty((M, N), device="cuda")
    us_old = time_it(lambda: eng.debug_gemm(A, W, b, R, act, cache_weights=True, out=Y))
    fl = 2.0 * M * N * K
    print(f"{label:16s} {M:6d} x {N:4d} x {K:4d}: ST {us:7.1f} us {fl / us / 1e6:6.1f} TF-eq | register-staged {us_old:7.1f} us "
          f"{fl / us_old / 1e6:6.1f} TF-eq", flush=True)
