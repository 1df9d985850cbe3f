"""GPU: the fp32-MFMA GEMM kernel against a plain PyTorch fp32 reference of the same op (all tile
configurations, ragged M, every fused epilogue)."""
import pytest
import torch

from workloads import synth

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


@pytest.mark.parametrize("M", [16384, 16449, 291208])
@pytest.mark.parametrize("act,strided", [(0, False), (1, True)])
def test_gemm_weight_stationary_kernel(eng, M, act, strided):
    """the K = 128 -> 256 token layer at token-count sizes takes the weight-stationary kernel (lt_gemm_ws.h): whole tiles, a
    ragged last tile (16 449 = 257 x 64 + 1), the cfg3 size; bias + ReLU; row-strided input and output views."""
    g = torch.Generator(device="cuda").manual_seed(M + act)
    Abuf = torch.randn(M, 192 if strided else 128, device="cuda", generator=g)
    A = Abuf[:, :128]
    W = torch.randn(256, 128, device="cuda", generator=g) / 128 ** 0.5
    b = torch.randn(256, device="cuda", generator=g)
    Ybuf = torch.full((M + 1, 320 if strided else 256), -7.0, device="cuda")
    Y = eng.debug_gemm(A, W, b, act=act, out=Ybuf[:M, :256])
    ref = A.double() @ W.double().t() + b.double()
    if act:
        ref = ref.clamp_min(0)
    assert (Y - ref.float()).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())
    assert (Ybuf[M] == -7.0).all() and (not strided or (Ybuf[:, 256:] == -7.0).all())     # nothing written outside the view
    eng.set_profiling(True)
    eng.debug_gemm(A, W, b, act=act, out=Ybuf[:M, :256])
    torch.cuda.synchronize()
    names = {e["name"] for e in eng.get_profile()}
    eng.set_profiling(False)
    assert "gemm_bf16x6_ws64x256" in names, names


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


@pytest.mark.parametrize("mode,tol", [("bf16x3", 3e-5), ("f16x3", 4e-6)])
@pytest.mark.parametrize("M,N,K,act", [(25472, 512, 512, 1), (25473, 256, 256, 2), (25361, 768, 256, 0)])
def test_gemm_112_row_tiles(eng, mode, tol, M, N, K, act):
    """Shapes for which the dispatcher picks the 112 x 256 kernel (v_mfma_f32_16x16x32, lt_gemm_split16.h) in the
    3-product modes: full tiles, a ragged last tile (25 473 = 227 x 112 + 49), every epilogue piece (bias, activation,
    residual) against float64."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    R = torch.randn(M, N, device="cuda", generator=g)
    eng.set_precision(mode)
    try:
        Y = eng.debug_gemm(A, W, b, R, act)
    finally:
        eng.set_precision("bf16x6")
    x = A.double() @ W.double().t() + b.double()
    x = [x, torch.relu(x), torch.nn.functional.gelu(x)][act] + R.double()
    assert ((Y.double() - x).abs().max() / x.abs().max()).item() < tol


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


def _folded_mlp_reference(sd, prefix, x, n_layers=3):
    """Plain PyTorch fp32 reference of the first `n_layers` Conv1d(k=1) + BatchNorm(eval) + ReLU blocks."""
    import torch.nn.functional as F
    for i in range(n_layers):
        w = torch.from_numpy(sd[f"{prefix}.{3 * i}.weight"])[:, :, 0]
        x = F.linear(x, w, torch.from_numpy(sd[f"{prefix}.{3 * i}.bias"]))
        x = F.batch_norm(x, torch.from_numpy(sd[f"{prefix}.{3 * i + 1}.running_mean"]),
                         torch.from_numpy(sd[f"{prefix}.{3 * i + 1}.running_var"]),
                         torch.from_numpy(sd[f"{prefix}.{3 * i + 1}.weight"]),
                         torch.from_numpy(sd[f"{prefix}.{3 * i + 1}.bias"]), False, 0.0, 1e-5)
        x = F.relu(x)
    return x


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 1000, 70001])
def test_fused_posenc_layers_vs_torch(rows):
    """mlp123_kernel (exact-fp32 MFMA, layers 1-3 of both positional encoders) against stock PyTorch fp32 on the same
    rows: ragged row counts around the 32-row MFMA step and the per-wave row split."""
    from workloads import synth
    from linetr_amd.engine import Engine
    sd = synth.calibrated_state_dict()
    eng = Engine(sd, "cuda:0")
    g = torch.Generator().manual_seed(rows)
    H, W = 480, 640
    cx, cy, scale = W / 2.0, H / 2.0, 0.7 * max(H, W)
    # word encoder: [x, y, score]
    pnt = torch.rand(rows, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    score = torch.rand(rows, generator=g)
    got = eng.debug_posenc("word", pnt.cuda(), score.cuda()).cpu()
    xin = torch.cat([(pnt - torch.tensor([cx, cy])) / scale, score[:, None]], dim=1)
    want = _folded_mlp_reference(sd, "klenc.word_position_enc.encoder", xin)
    assert got.shape == (rows, 128)
    assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
    # line encoder: [mid_x, mid_y, resp, cos2t, sin2t]
    sl = torch.rand(rows, 2, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    resp = torch.rand(rows, generator=g)
    ang = torch.rand(rows, 2, generator=g) * 2 - 1
    got = eng.debug_posenc("line", sl.cuda(), resp.cuda(), ang.cuda()).cpu()
    sn = (sl - torch.tensor([cx, cy])) / scale
    xin = torch.cat([(sn[:, 0] + sn[:, 1]) / 2, resp[:, None], ang], dim=1)
    want = _folded_mlp_reference(sd, "klenc.line_position_enc.encoder", xin)
    assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("mode,tol", [("f32", 2e-6), ("bf16x6", 2e-6), ("f16x3", 4e-6), ("bf16x3", 3e-5)])
@pytest.mark.parametrize("M,N,K,act", [(398, 512, 512, 1), (33, 256, 1024, 2), (1, 768, 256, 0), (400, 64, 544, 3),
                                       (398, 768, 256, 0), (500, 768, 512, 1)])
def test_gemm_small_m_kernel(eng, mode, tol, M, N, K, act):
    """Single-pair sizes: the barrier-free K-split kernel (lt_gemm_small.h: 32 x 32 tiles, the block's 4 waves split K,
    fragments straight from global memory) with every epilogue piece, ragged M, K = 17 tiles (uneven split), against
    float64.  The last two shapes have 312 / 384 blocks: the 4-wave, one-staging-buffer variant of which two blocks share a CU.
    In f32 mode the same shapes run on the tiled fp32 kernel."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    R = torch.randn(M, N, device="cuda", generator=g)
    eng.set_precision(mode)
    try:
        Y = eng.debug_gemm(A, W, b, R, act)
    finally:
        eng.set_precision("bf16x6")
    x = A.double() @ W.double().t() + b.double()
    x = [x, torch.relu(x), torch.nn.functional.gelu(x), (2 - 2 * x).clamp(min=0)][act] + R.double()
    assert ((Y.double() - x).abs().max() / x.abs().max()).item() < tol
