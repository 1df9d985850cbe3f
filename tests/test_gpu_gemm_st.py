"""GPU: the split-tile (ST) operand format and the LDS-DMA GEMM on it (csrc/lt_gemm_st.h) against float64: exact
round trip of the format, ragged M, two-source K (the [x ; message] concatenation of models/line_transformer.py:166),
bias / ReLU / residual epilogues, ST and fp32 outputs, and agreement with the register-staged split GEMM."""
import pytest
import torch

from linetr_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from linetr_amd.engine import Engine
    return Engine(synth.make_state_dict(0), "cuda:0")


def _rnd(g, *shape):
    return torch.randn(*shape, device="cuda", generator=g)


@pytest.mark.parametrize("rows,K", [(1, 16), (300, 96), (2049, 256)])
def test_st_round_trip_is_exact(eng, rows, K):
    g = torch.Generator(device="cuda").manual_seed(rows + K)
    X = _rnd(g, rows, K) * torch.exp(_rnd(g, rows, K) * 4)      # wide dynamic range: the three planes must carry all 24 bits
    X[0, 0] = 0.0
    st = eng.to_st(X)
    assert st.numel() == (rows + 127) // 128 * 8 * (K // 16) * 1536
    assert torch.equal(eng.from_st(st, rows, K), X)


@pytest.mark.parametrize("M,N,K1,K2,bias,res,act,st_out", [
    (128, 256, 32, 0, False, False, 0, False),     # one tile, two K steps (shorter than the DMA ring)
    (300, 256, 64, 0, True, False, 0, True),       # ragged M: rows 300..383 of the last tile are padding
    (1000, 512, 256, 256, True, False, 1, True),   # W1: [x ; message], BatchNorm-folded bias, ReLU
    (777, 256, 512, 0, True, True, 0, True),       # W2: residual read from an ST image
    (2000, 768, 256, 0, True, False, 0, True),     # q/k/v projection
    (515, 256, 256, 512, True, False, 0, False),   # final projection: [z ; hidden] -> fp32 rows
    (25472, 256, 512, 0, True, True, 0, True),     # cfg3 size
])
def test_gemm_st_vs_float64(eng, M, N, K1, K2, bias, res, act, st_out):
    g = torch.Generator(device="cuda").manual_seed(M + N + K1 + K2)
    A1, A2 = _rnd(g, M, K1), (_rnd(g, M, K2) if K2 else None)
    K = K1 + K2
    W = _rnd(g, N, K) / K ** 0.5
    b = _rnd(g, N) if bias else None
    R = _rnd(g, M, N) if res else None
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
    scale = max(1.0, want.abs().max().item())
    assert (got.double() - want).abs().max().item() < 2e-6 * scale
    # same six products, same fp32 accumulation as the register-staged kernel: the two agree to fp32 summation noise
    old = eng.debug_gemm(A, W, b, R, act)
    assert (got - old).abs().max().item() < 2e-6 * scale
