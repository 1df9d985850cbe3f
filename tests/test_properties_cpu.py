"""CPU property tests (hypothesis): invariants of the tokeniser restated by the oracle, and the native host
pre-filter against the oracle on random detector outputs (incl. vertical lines, octaves, border cases)."""
import ctypes as C

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from linetr_amd import _native as nat
from workloads import synth
from oracle import linetr_oracle as O

torch.set_grad_enabled(False)
H, W = 480, 640
DD, DS = (torch.from_numpy(a) for a in synth.synth_dense_maps_np(3, H, W))


@st.composite
def line_sets(draw):
    n = draw(st.integers(2, 40))
    rs = np.random.RandomState(draw(st.integers(0, 2 ** 31 - 1)))
    rows = []
    for _ in range(n):
        sx, sy = rs.uniform(0, W), rs.uniform(0, H)
        kind = rs.randint(4)
        ln = rs.uniform(5, 400)
        th = [rs.uniform(0, 2 * np.pi), np.pi / 2, -np.pi / 2, 0.0][kind]       # random / vertical up / down / horizontal
        ex, ey = sx + ln * np.cos(th), sy + ln * np.sin(th)
        octave = int(rs.randint(0, 2))
        sx, sy, ex, ey = (float(np.float32(v)) for v in (sx, sy, ex, ey))
        length = float(np.float32(np.hypot(ex - sx, ey - sy))) / (2 ** octave)   # cv2 reports length at the octave's scale
        rows.append([sx, sy, ex, ey, length * 0.999, float(octave)])
    return np.asarray(rows)


@settings(max_examples=40, deadline=None)
@given(rows=line_sets(), td=st.sampled_from([8.0, 12.8, 16.0]), T=st.sampled_from([5, 21, 41]),
       border=st.sampled_from([0, 8]), max_k=st.sampled_from([-1, 7, 256]))
def test_native_prefilter_equals_oracle(rows, td, T, border, max_k):
    lines = O.keep_long_lines(O.drop_border_lines(O.cv2_to_arrays(synth.array_to_keylines(rows)), border, H, W, None),
                              16, max_k)
    recs = np.zeros(len(rows), dtype=nat.REC_DTYPE)
    k, n = C.c_int32(), C.c_int32()
    code = nat.lib().linetr_prefilter(nat.np_ptr(np.ascontiguousarray(rows)), len(rows), H, W, border, 16.0, max_k, None,
                                      td, T, 0, 0, 0, nat.np_ptr(recs), len(recs), C.byref(k), C.byref(n))
    if code == nat.E_ASSERT:       # the reference would raise AssertionError inside point_on_line for the same input
        return
    nat.check(code)
    recs = recs[:k.value]
    assert len(recs) == len(lines["klines"])
    if len(recs) == 0:
        return
    order_free = len(np.unique(lines["length_klines"])) == len(recs)      # ties: order is implementation-defined
    if order_free:
        assert np.array_equal(recs["sp"], lines["klines"][:, 0]) and np.array_equal(recs["ep"], lines["klines"][:, 1])
        assert np.abs(recs["angle"] - lines["angles"]).max() < 1e-14
    ntok = np.ceil(np.sort(lines["length_klines"])[::-1] / td).astype(int)
    assert np.array_equal(recs["n_tok"], ntok)
    assert np.array_equal(recs["n_sub"], -(-ntok // T)) and n.value == recs["n_sub"].sum()


@settings(max_examples=15, deadline=None)
@given(rows=line_sets(), T=st.sampled_from([5, 21]))
def test_tokenizer_invariants(rows, T):
    cfg = dict(min_length=16, token_distance=8, max_tokens=T, remove_borders=8, max_keylines=-1)
    try:
        out = O.preprocess(synth.array_to_keylines(rows), (1, 1, H, W), DD, DS, cfg)
    except AssertionError:
        return
    if len(out["klines"]) == 0:
        return
    A = out["mat_klines2sublines"][0].numpy()
    mask = out["mask_sublines"][0, :, :, 0].numpy()
    pnt = out["pnt_sublines"][0].numpy()
    sub = out["sublines"][0].numpy()
    ntok = np.ceil(out["length_klines"][0].numpy().astype(np.float64) / 8).astype(int)
    assert np.allclose(A.sum(1), 1) and ((A > 0).sum(0) == 1).all()                 # every sub-line has one key-line
    assert mask[:, 0].all() and mask.sum() - len(mask) == np.ceil(out["length_klines"][0].numpy() / np.float32(8)).sum()
    assert (out["resp_sublines"].numpy() > 0).all()   # (not <= 1: the last sub-line absorbs any geometric slack)
    # sub-lines of a key-line chain end-to-start, padded slots sit at (0,0)
    owner = A.argmax(0)
    for i in range(1, len(sub)):
        if owner[i] == owner[i - 1]:
            assert np.array_equal(sub[i, 0], sub[i - 1, 1])
    assert (pnt[mask[:, 1:] == 0] == 0).all()
    d = np.linalg.norm(out["desc_sublines"][0].numpy(), axis=-1)
    assert np.abs(d - 1).max() < 1e-5
