"""GPU: the kernels and switches that were measured and NOT shipped, in the experiments build of the library
(experiments/liblinetr_hip_experiments.so = the product sources with -DLINETR_EXPERIMENTS + experiments/csrc; `python -m linetr_amd.build --experiments`):
the stream-K tail, the fused signature MLP, the separate row-norm path, the split-tile (ST) operand format with its
LDS-DMA GEMM and attention, and the row-tile-local GEMM chains.  They stay correct (these tests) so that their numbers in
DESIGN.md can be reproduced; the product library contains none of them."""
import os

import numpy as np
import pytest
import torch

from linetr_amd import _native as nat
from workloads import synth
from test_gpu_properties import batch_inputs, describe

# NOT part of `pytest tests -m gpu` (which loads the product library only): run on the GPU box with
#   python -m linetr_amd.build --experiments && python -m pytest experiments -q -m experiments
pytestmark = [pytest.mark.experiments,
              pytest.mark.skipif(not os.path.exists(nat.EXPERIMENTS_LIB_PATH), reason="experiments library not built"),
              pytest.mark.skipif(not torch.cuda.is_available(), reason="no HIP device")]
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def eng():
    from linetr_amd.engine import Engine
    return Engine(synth.calibrated_state_dict(), "cuda:0", lib_path=nat.EXPERIMENTS_LIB_PATH)


def test_fused_row_norm_epilogue_equals_separate_kernel(eng, monkeypatch):
    """cfg3 batch (25 472 rows: the 128x256 GEMM tile owns complete rows): LayerNorm x 2 and the final L2 normalisation
    fused into the GEMM epilogues (lt_gemm_split.h) against the same GEMMs followed by row_norm_kernel
    (LINETR_NO_FUSED_NORM=1).  Same arithmetic in the same order, so the descriptors agree to the last bits."""
    _, cat, off, dd, ds = batch_inputs(128)
    monkeypatch.delenv("LINETR_NO_FUSED_NORM", raising=False)
    _, fused = describe(eng, cat, off, dd, ds)
    monkeypatch.setenv("LINETR_NO_FUSED_NORM", "1")
    _, plain = describe(eng, cat, off, dd, ds)
    assert fused.shape == plain.shape and fused.shape[0] == 128 * 199
    assert (fused - plain).abs().max().item() <= 2e-7
    assert ((fused.norm(dim=1) - 1).abs() < 1e-5).all()


@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3", "f16x3"])
def test_fused_signature_mlp_equals_two_gemms(eng, monkeypatch, mode):
    """W1 -> ReLU -> W2 + residual of a signature layer in ONE kernel (lt_mlp_fused.h: transposed products, hidden
    activations in registers, W2's K index permuted to the MFMA C/D register order) against the two tiled GEMMs.  Same
    split planes and cross terms; only the fp32 summation order inside a product differs."""
    _, cat, off, dd, ds = batch_inputs(128)
    eng.set_precision(mode)
    try:
        monkeypatch.delenv("LINETR_FUSED_SIG_MLP", raising=False)
        monkeypatch.setenv("LINETR_NO_FUSED_SIG_MLP", "1")
        _, plain = describe(eng, cat, off, dd, ds)
        plain = plain.clone()
        monkeypatch.delenv("LINETR_NO_FUSED_SIG_MLP", raising=False)
        monkeypatch.setenv("LINETR_FUSED_SIG_MLP", "1")
        _, fused = describe(eng, cat, off, dd, ds)
    finally:
        eng.set_precision("bf16x6")
    assert fused.shape == plain.shape and fused.shape[0] == 128 * 199
    tol = {"bf16x6": 1e-6, "bf16x3": 5e-5, "f16x3": 5e-6}[mode]
    assert (fused - plain).abs().max().item() <= tol
    assert ((fused.norm(dim=1) - 1).abs() < 1e-5).all()


@pytest.mark.parametrize("M,N,K,act", [(25472, 512, 512, 1), (25472, 768, 256, 0), (9584, 512, 512, 2), (25473, 512, 256, 0),
                                       (16500, 256, 1024, 0)])
def test_gemm_stream_k_tail(eng, M, N, K, act, monkeypatch):
    """(stream-K is opt-in, LINETR_STREAMK=1: measured slower than the plain launch as built, see lt_gemm_split.h.)
    Shapes whose last round of 128x256 tiles would leave part of the chip idle: the tail tiles are shared by one block
    per CU in (tile, K-tile) runs, partial accumulator tiles travel through the workspace (lt_gemm_split.h).  Against
    float64 with every epilogue piece, a ragged last row tile, and twice in a row: the partition and the summation order
    are fixed, so the result is bit-reproducible."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    R = torch.randn(M, N, device="cuda", generator=g)
    monkeypatch.setenv("LINETR_STREAMK", "1")
    Y1 = eng.debug_gemm(A, W, b, R, act).clone()
    Y2 = eng.debug_gemm(A, W, b, R, act)
    monkeypatch.delenv("LINETR_STREAMK")
    Y0 = eng.debug_gemm(A, W, b, R, act)                 # the plain launch: same products, different summation tree
    assert torch.equal(Y1, Y2)
    assert (Y1 - Y0).abs().max().item() < 1e-4 * max(1.0, Y0.abs().max().item())
    x = A.double() @ W.double().t() + b.double()
    x = [x, torch.relu(x), torch.nn.functional.gelu(x)][act] + R.double()
    assert ((Y1.double() - x).abs().max() / x.abs().max()).item() < 2e-6


# ---- split-tile operands and the LDS-DMA GEMM (csrc/lt_gemm_st.h) --------------------------------------------------

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


@pytest.mark.parametrize("path_env", [{"LINETR_SIG_PATH": "st"}, {"LINETR_SIG_PATH": "st", "LINETR_ATTN_ST_OCC1": "1"},
                                      {"LINETR_GEMM_CHAIN": "1"}, {"LINETR_GEMM_CHAIN": "1", "LINETR_CHAIN_W1_ALONE": "1"},
                                      {"LINETR_NO_FUSED_QKV_ATTN": "1"}, {"LINETR_NO_GEMM_WS": "1"}, {"LINETR_NO_TOKMLP": "1"},
                                      {"LINETR_NO_TOKMLP": "1", "LINETR_NO_GEMM_WS": "1"}])
def test_alternative_signature_paths_equal_the_shipped_one(eng, monkeypatch, path_env):
    """cfg3 batch through the split-tile path (ST GEMMs + ST attention, activations as bf16 planes in HBM) and through the
    row-tile-local GEMM chains, against the shipped path of the same library: same products, same accumulation type."""
    _, cat, off, dd, ds = batch_inputs(128)
    for k in ("LINETR_SIG_PATH", "LINETR_ATTN_ST_OCC1", "LINETR_GEMM_CHAIN", "LINETR_CHAIN_W1_ALONE", "LINETR_NO_FUSED_QKV_ATTN", "LINETR_NO_GEMM_WS", "LINETR_NO_TOKMLP"):
        monkeypatch.delenv(k, raising=False)
    _, plain = describe(eng, cat, off, dd, ds)
    plain = plain.clone()
    for k, v in path_env.items():
        monkeypatch.setenv(k, v)
    _, alt = describe(eng, cat, off, dd, ds)
    assert alt.shape == plain.shape and alt.shape[0] == 128 * 199
    assert (alt - plain).abs().max().item() <= 2e-6
    assert ((alt.norm(dim=1) - 1).abs() < 1e-5).all()


def test_persistent_pair_network_equals_the_launch_chain(eng, monkeypatch):
    """lt_pairnet.h: the signature network of a few small images as ONE persistent launch (arrival counters, write-through
    hand-offs; LINETR_PAIRNET=1, experiments build: measured and not shipped) against the ~30-launch chain.  Same tile arithmetic
    (K split over 8 waves in every GEMM, the attention's keys over 8 waves: summation order only), so the descriptors agree to fp32 round-off;
    and the persistent launch is deterministic: repeated runs are bit-identical (a missed hand-off would show as a flicker)."""
    hw = (480, 640)
    cases = {"pair": [200, 200], "ragged8": [2, 3, 32, 33, 34, 98, 200, 257], "one_line": [2], "two_tiles": [40]}
    for name, counts in cases.items():
        lines = [synth.synth_lines(8100 + i, n, *hw) for i, n in enumerate(counts)]
        maps = [synth.synth_dense_maps(8100 + i, *hw) for i in range(len(counts))]
        dd = torch.cat([m[0] for m in maps]).cuda()
        ds = torch.cat([m[1] for m in maps]).cuda()
        off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)
        cat = np.concatenate(lines)
        monkeypatch.delenv("LINETR_PAIRNET", raising=False)
        tb0, chain = describe(eng, cat, off, dd, ds)
        chain = chain.clone()
        monkeypatch.setenv("LINETR_PAIRNET", "1")
        eng.set_profiling(True)
        tb1, pn = describe(eng, cat, off, dd, ds)
        prof = {e["name"]: e["calls"] for e in eng.get_profile()}
        eng.set_profiling(False)
        assert prof.get("pair_net_bf16x6") == 1, (name, prof)
        assert not any(k.startswith("sig_attn") for k in prof), prof
        assert tb1.N == tb0.N == chain.shape[0] and torch.isfinite(pn).all()
        assert (pn - chain).abs().max().item() <= 2e-6, (name, (pn - chain).abs().max().item())
        first = pn.clone()
        for _ in range(25):
            _, again = describe(eng, cat, off, dd, ds)
            assert torch.equal(again, first), name


def test_persistent_pair_network_is_not_taken_for_large_batches(eng, monkeypatch):
    monkeypatch.setenv("LINETR_PAIRNET", "1")
    _, cat, off, dd, ds = batch_inputs(16)            # 16 x 199 rows > PN_MAX_ROWS and > 8 images
    eng.set_profiling(True)
    describe(eng, cat, off, dd, ds)
    prof = {e["name"] for e in eng.get_profile()}
    eng.set_profiling(False)
    assert "pair_net_bf16x6" not in prof
