"""GPU: the two-stream software pipeline over consecutive describe calls (linetr_describe_submit / linetr_describe_join,
Engine.describe(pipeline_slot=...), DescribePipeline) gives describe_lines' results BIT FOR BIT -- it runs the same kernels on the same
data, only on two library-owned streams -- for batches of different sizes following each other (workspaces of a slot re-used by a
larger / smaller batch), for NCHW- and NHWC-fed maps, with an empty batch in the sequence, and when the caller reads a batch late.
No reference counterpart: the reference is serial per pair (models/matching.py:34-60); SURVEY.md section 7 step 5 asks for the pipelining."""
import numpy as np
import pytest
import torch

from workloads import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
HW = (480, 640)
CFG = dict(remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21)


@pytest.fixture(scope="module")
def eng():
    from linetr_amd.engine import Engine
    return Engine(synth.calibrated_state_dict(), "cuda:0")


def batch(n_img, seed0, n_lines=200):
    lines = [synth.synth_lines(seed0 + i, n_lines if not callable(n_lines) else n_lines(i), HW[0], HW[1], 17.0, 167.0) for i in range(n_img)]
    maps = [synth.synth_dense_maps(seed0 + i, *HW) for i in range(n_img)]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)
    return np.concatenate(lines), off, dd, ds


SMALL = ("klines", "length", "angles", "sublines", "resp", "angle_sub", "sub2line")


def same(a, b):
    (tba, lda), (tbb, ldb) = a, b
    assert tba.N == tbb.N and tba.K == tbb.K
    assert np.array_equal(tba.cu_n, tbb.cu_n) and np.array_equal(tba.cu_k, tbb.cu_k)
    assert torch.equal(lda, ldb)
    for k in SMALL:
        assert torch.equal(getattr(tba, k), getattr(tbb, k)), k


def test_pipelined_batches_equal_the_serial_calls(eng):
    from linetr_amd.engine import DescribePipeline
    # sizes chosen so that a slot's workspace is re-used by a larger and by a smaller batch, the fused projection + attention kernel
    # (>= 32 images) and the GEMM + attention pair (few images) both occur, and a single pair takes the latency kernels
    specs = [(64, 9000, 200), (8, 9100, 200), (128, 9200, 200), (2, 9300, 200), (48, 9400, lambda i: 3 + (i * 37) % 255), (64, 9000, 200)]
    batches = [batch(*s) for s in specs]
    serial = []
    for cat, off, dd, ds in batches:
        serial.append(eng.describe_lines(cat, off, dd, ds, **CFG))
    torch.cuda.synchronize()
    for layout, depth in (("nchw", 2), ("nhwc", 2), ("nchw", 3), ("nhwc", 4)):
        pipe = DescribePipeline(eng, depth)
        got = []
        for cat, off, dd, ds in batches:
            feed = dd if layout == "nchw" else dd.permute(0, 2, 3, 1).contiguous()
            done = pipe.submit(cat, off, feed, ds, dense_layout=layout, **CFG)
            if done is not None:
                got.append(done)
        got += pipe.drain()
        assert pipe.drain() == []
        torch.cuda.synchronize()
        assert len(got) == len(serial)
        for a, b in zip(got, serial):
            same(a, b)
    # the first and the last batch are the same inputs: run to run determinism across slots and neighbours
    same(serial[0], serial[-1])


def test_pipelined_long_line_batches_equal_the_serial_calls():
    """cfg5's shape (960 x 1280, 600 lines of 40-327 px, 41 tokens: images above 256 sub-lines take the q/k/v GEMM + flash attention pair
    instead of the fused kernel, and the cut points fall between other launches) through the pipeline: bit-identical to the plain call."""
    from linetr_amd.engine import DescribePipeline, Engine
    hw = (960, 1280)
    e = Engine(synth.calibrated_state_dict(), "cuda:0", image_shape=list(hw))
    cfg = dict(remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=41)
    sets = []
    for b, seed0 in ((4, 9700), (2, 9710), (6, 9720)):
        lines = [synth.synth_lines(seed0 + i, 600 - 40 * i, hw[0], hw[1], 40.0, 327.0) for i in range(b)]
        maps = [synth.synth_dense_maps(seed0 + i, *hw) for i in range(b)]
        off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)
        sets.append((np.concatenate(lines), off, torch.cat([m[0] for m in maps]).cuda().permute(0, 2, 3, 1).contiguous(), torch.cat([m[1] for m in maps]).cuda()))
    serial = [e.describe_lines(*s, dense_layout="nhwc", **cfg) for s in sets]
    torch.cuda.synchronize()
    assert max(int(np.diff(tb.cu_n).max()) for tb, _ in serial) > 256
    pipe = DescribePipeline(e, 3)
    got = [r for r in (pipe.submit(*s, dense_layout="nhwc", **cfg) for s in sets + sets) if r is not None] + pipe.drain()
    torch.cuda.synchronize()
    assert len(got) == 6
    for a, b in zip(got, serial + serial):
        same(a, b)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x3"])
def test_pipelined_batches_in_the_other_precisions(mode):
    """The cut points sit in forward_core, whatever kernels the arithmetic mode dispatches: exact-fp32 MFMA GEMMs + sig_attn_kernel (f32),
    the two-plane tiles (bf16x3 / f16x3) -- pipelined results equal the plain call's bit for bit there too."""
    from linetr_amd.engine import DescribePipeline, Engine
    e = Engine(synth.calibrated_state_dict(), "cuda:0")
    e.set_precision(mode)
    sets = [batch(24, 9800), batch(2, 9810), batch(40, 9820)]
    serial = [e.describe_lines(*s, **CFG) for s in sets]
    torch.cuda.synchronize()
    pipe = DescribePipeline(e, 3)
    got = [r for r in (pipe.submit(*s, **CFG) for s in sets) if r is not None] + pipe.drain()
    torch.cuda.synchronize()
    for a, b in zip(got, serial):
        same(a, b)


def test_pipeline_with_an_empty_batch_and_a_late_reader(eng):
    from linetr_amd.engine import DescribePipeline
    full = batch(16, 9500)
    empty = (np.zeros((0, 6)), np.zeros(17, np.int32), full[2], full[3])       # 16 images without a detected line
    ref = eng.describe_lines(*full, **CFG)
    torch.cuda.synchronize()
    pipe = DescribePipeline(eng)
    assert pipe.submit(*full, **CFG) is None
    a = pipe.submit(*empty, **CFG)            # joins the first batch
    b = pipe.submit(*full, **CFG)             # "joins" the empty one
    assert b[0].N == 0 and b[1].shape[0] == 0
    c = pipe.submit(*full, **CFG)
    (d,) = pipe.drain()
    # nothing was synchronised on the host so far: the reads below are ordered by the joins alone
    for got in (a, c, d):
        same(got, ref)


def test_submit_without_alternating_slots_is_safe(eng):
    """Two submits to the SAME slot: the second waits (on the host) for the first, so the shared workspace is never overwritten under a
    running batch; results stay exact."""
    full = batch(32, 9600)
    ref = eng.describe_lines(*full, **CFG)
    torch.cuda.synchronize()
    r0 = eng.describe_lines(*full, pipeline_slot=(0, 2), **CFG)
    r1 = eng.describe_lines(*full, pipeline_slot=(0, 2), **CFG)
    eng.describe_join(0)
    same(r1, ref)
    same(r0, ref)      # r0's outputs are its own tensors; batch 0 completed before batch 1 was queued


def test_join_of_an_unused_slot_is_refused():
    from linetr_amd import _native as nat
    from linetr_amd.engine import Engine
    e = Engine(synth.calibrated_state_dict(), "cuda:0")
    with pytest.raises(nat.NativeError):
        e.describe_join(1)
    with pytest.raises(nat.NativeError):
        e.describe_join(7)
    with pytest.raises(ValueError):
        from linetr_amd.engine import DescribePipeline
        DescribePipeline(e, 9)
