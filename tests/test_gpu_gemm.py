"""GPU: the fp32-MFMA GEMM kernel against a plain PyTorch fp32 reference of the same op (all tile
configurations, ragged M, every fused epilogue)."""
import pytest
import torch

from linetr_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from linetr_amd.engine import Engine
    return Engine(synth.make_state_dict(0), "cuda:0")


@pytest.mark.parametrize("M,N,K", [(1, 64, 32), (199, 256, 256), (4179, 128, 64), (25472, 256, 512), (70000, 64, 32),
                                   (333, 768, 256), (130, 1024, 256), (257, 256, 1024), (50000, 256, 128)])
def test_gemm_shapes(eng, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    Y = eng.debug_gemm(A, W, b)
    ref = (A.double() @ W.double().t() + b.double()).float()
    assert (Y - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("mode,tol", [("f32", 2e-6), ("bf16x6", 2e-6), ("f16x3", 4e-6), ("bf16x3", 3e-5)])
def test_gemm_precision_modes(eng, mode, tol):
    """every MFMA path against a float64 reference (relative to the largest output magnitude)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn(3000, 512, device="cuda", generator=g)
    W = torch.randn(512, 512, device="cuda", generator=g) / 512 ** 0.5
    eng.set_precision(mode)
    try:
        Y = eng.debug_gemm(A, W)
    finally:
        eng.set_precision("bf16x6")
    ref = (A.double() @ W.double().t()).float()
    assert ((Y - ref).abs().max() / ref.abs().max()).item() < tol


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_epilogues(eng, act):
    g = torch.Generator(device="cuda").manual_seed(act)
    A = torch.randn(777, 256, device="cuda", generator=g)
    W = torch.randn(256, 256, device="cuda", generator=g) / 16
    b = torch.randn(256, device="cuda", generator=g)
    R = torch.randn(777, 256, device="cuda", generator=g)
    Y = eng.debug_gemm(A, W, b, R, act)
    x = A @ W.t() + b
    x = [x, torch.relu(x), torch.nn.functional.gelu(x), (2 - 2 * x).clamp(min=0)][act] + R
    assert (Y - x).abs().max().item() < 5e-5
