"""CPU (no GPU needed): the C-ABI library loads, exports every symbol include/linetr_hip.h declares,
and its host-side pre-filter (a1-a3) agrees with the oracle.  No device compute is invoked."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import BASE_CFG, load
from linetr_amd import _native as nat
from workloads import synth
from oracle import linetr_oracle as O

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    L = nat.lib()
    hdr = open(os.path.join(ROOT, "include", "linetr_hip.h")).read()
    # the section behind LINETR_EXPERIMENTS is declared for (and exported by) the experiments build only
    exp = re.search(r"#ifdef LINETR_EXPERIMENTS(.*?)#endif\s*/\* LINETR_EXPERIMENTS \*/", hdr, re.S)
    assert exp, "experiments section of the header not found"
    product_hdr = hdr.replace(exp.group(0), "")
    declared = set(re.findall(r"\b(linetr_[a-z0-9_]+)\s*\(", product_hdr))
    assert declared == set(nat.EXPORTS), declared ^ set(nat.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    declared_x = set(re.findall(r"\b(linetr_[a-z0-9_]+)\s*\(", exp.group(1)))
    assert declared_x == set(nat.EXPERIMENT_EXPORTS), declared_x ^ set(nat.EXPERIMENT_EXPORTS)
    for name in declared_x:
        assert not hasattr(L, name), f"{name} must not be in the product library"
    if os.path.exists(nat.EXPERIMENTS_LIB_PATH):
        LX = nat.lib(nat.EXPERIMENTS_LIB_PATH)
        for name in declared | declared_x:
            assert hasattr(LX, name), name
    assert L.linetr_abi_version() == 5
    assert C.sizeof(nat.LineRec) == 80


def prefilter(rows, hw, cfg, vm=None):
    L = nat.lib()
    recs = np.zeros(max(len(rows), 1), dtype=nat.REC_DTYPE)
    k, n = C.c_int32(), C.c_int32()
    rows = np.ascontiguousarray(rows, np.float64)
    code = L.linetr_prefilter(nat.np_ptr(rows), len(rows), hw[0], hw[1], cfg["remove_borders"], float(cfg["min_length"]),
                              cfg["max_keylines"], nat.np_ptr(vm) if vm is not None else None,
                              float(cfg["token_distance"]), cfg["max_tokens"], 0, 0, 0, nat.np_ptr(recs), len(recs),
                              C.byref(k), C.byref(n))
    nat.check(code)
    return recs[:k.value], n.value


def oracle_prefilter(rows, hw, cfg, vm=None):
    lines = O.cv2_to_arrays(synth.array_to_keylines(rows))
    lines = O.drop_border_lines(lines, cfg["remove_borders"], hw[0], hw[1], vm)
    return O.keep_long_lines(lines, cfg["min_length"], cfg["max_keylines"])


@pytest.mark.parametrize("seed", [11, 12, 77])
def test_prefilter_matches_oracle_random(seed):
    hw = (480, 640)
    rows = synth.synth_lines(seed, 200, *hw, len_lo=5.0)
    recs, n = prefilter(rows, hw, BASE_CFG)
    ref = oracle_prefilter(rows, hw, BASE_CFG)
    assert len(recs) == len(ref["klines"])
    assert np.array_equal(recs["sp"], ref["klines"][:, 0]) and np.array_equal(recs["ep"], ref["klines"][:, 1])
    assert np.array_equal(recs["length"], ref["length_klines"])
    assert np.abs(recs["angle"] - ref["angles"]).max() < 1e-15
    ntok = np.ceil(ref["length_klines"] / 8).astype(int)
    assert np.array_equal(recs["n_tok"], ntok)
    assert np.array_equal(recs["n_sub"], -(-ntok // 21))
    assert n == recs["n_sub"].sum()
    assert np.array_equal(recs["first_sub"], np.cumsum(recs["n_sub"]) - recs["n_sub"])
    assert np.array_equal(recs["first_tok"], np.cumsum(recs["n_tok"]) - recs["n_tok"])


@pytest.mark.parametrize("name", ["tiny_default", "tiny_max3", "tiny_noborder", "tiny_float_td"])
def test_prefilter_special_lines(name):
    g = load(name)
    cfg = dict(BASE_CFG)
    for k in g.files:
        if k.startswith("cfg_"):
            cfg[k[4:]] = g[k].item()
    hw = tuple(int(v) for v in g["hw"])
    recs, n = prefilter(g["lines"], hw, cfg)
    assert len(recs) == g["klines"].shape[1] and n == g["sublines"].shape[1]
    assert np.array_equal(recs["sp"].astype(np.float32), g["klines"][0][:, 0])
    assert np.array_equal(recs["length"].astype(np.float32), g["length_klines"][0])


def test_prefilter_valid_mask_and_errors():
    g = load("tiny_validmask")
    hw = tuple(int(v) for v in g["hw"])
    vm = np.ones(hw)
    vm[:, :int(g["valid_mask_cols"])] = 0
    recs, n = prefilter(g["lines"], hw, BASE_CFG, vm)
    assert np.array_equal(recs["sp"].astype(np.float32), g["klines"][0][:, 0])
    # lineLength larger than the geometric length -> the reference's AssertionError (line_process.py:44-45)
    bad = np.array([[100.0, 100.0, 150.0, 100.0, 200.0, 0.0], [10.0, 10.0, 300.0, 10.0, 20.0, 0.0]])
    with pytest.raises(AssertionError):
        prefilter(bad, hw, BASE_CFG)
    # zero lines is fine (the reference would raise IndexError; strict superset)
    recs, n = prefilter(np.zeros((0, 6)), hw, BASE_CFG)
    assert len(recs) == 0 and n == 0


def test_pack_lines_matches_prefilter():
    hw = (480, 640)
    rows = synth.synth_lines(5, 50, *hw)
    recs, n = prefilter(rows, hw, BASE_CFG)
    L = nat.lib()
    out = np.zeros(len(recs), dtype=nat.REC_DTYPE)
    kl = np.ascontiguousarray(np.stack([recs["sp"], recs["ep"]], 1))
    n2 = C.c_int32()
    nat.check(L.linetr_pack_lines(nat.np_ptr(kl), nat.np_ptr(np.ascontiguousarray(recs["length"])),
                                  nat.np_ptr(np.ascontiguousarray(recs["angle"])), len(recs), 8.0, 21, 0, 0, 0,
                                  nat.np_ptr(out), C.byref(n2)))
    assert n2.value == n
    for f in nat.REC_DTYPE.names:
        assert np.array_equal(out[f], recs[f]), f


def test_equal_lengths_are_reported_and_reordered_like_numpy():
    """Equal detector lengths: linetr_prefilter_tied_images names the images, and engine.repack_like_numpy gives them the
    order (and, across the [:max_keylines] cut, the set) NumPy's argsort gives the reference on this host
    (models/line_process.py:15-18); images without ties are untouched."""
    from linetr_amd import engine, line_process as lp
    L = nat.lib()
    H, W, td, T = 480, 640, 8.0, 21
    imgs = [synth.synth_lines(300 + i, 60, H, W) for i in range(6)]
    for i, idx in ((1, (3, 11, 17, 40)), (4, (0, 59))):          # images 1 and 4: groups of equal lengths
        imgs[i][list(idx), 4] = 18.0 + i
    imgs[4][30:40, 4] = 25.0                                      # a larger group: unstable sorts scramble these
    cat = np.ascontiguousarray(np.concatenate(imgs))
    off = np.arange(7, dtype=np.int32) * 60
    for max_k in (-1, 50):                                        # 50: image 4's group of ten straddles nothing, -1 drops the shortest
        recs = np.zeros(len(cat), dtype=nat.REC_DTYPE)
        cu_k, cu_n = np.zeros(7, np.int32), np.zeros(7, np.int32)
        nat.check(L.linetr_prefilter_batch(nat.np_ptr(cat), nat.np_ptr(off), 6, H, W, 8, 16.0, max_k, None, td, T, 1,
                                           nat.np_ptr(recs), len(recs), nat.np_ptr(cu_k), nat.np_ptr(cu_n)))
        tied = np.full(6, -1, np.int32)
        assert L.linetr_prefilter_tied_images(nat.np_ptr(tied), 6) == 2 and tied[:2].tolist() == [1, 4]
        assert L.linetr_prefilter_tied_images(None, 0) == 2       # a query for the count alone
        before = recs.copy()
        for i in (1, 4):
            engine.repack_like_numpy(L, recs, cu_k, cu_n, i, cat[off[i]:off[i + 1]], H, W, 8, 16.0, max_k, td, T)
        for i in range(6):
            want = lp.filter_by_length(lp.remove_borders(lp.lines_from_rows(imgs[i].copy()), 8, H, W, None), 16.0, max_k)
            r = recs[cu_k[i]:cu_k[i + 1]]
            assert np.array_equal(r["sp"], want["klines"][:, 0]) and np.array_equal(r["ep"], want["klines"][:, 1]), i
            assert np.array_equal(r["length"], want["length_klines"])
            assert np.array_equal(r["first_sub"], cu_n[i] + np.concatenate([[0], np.cumsum(r["n_sub"])[:-1]]))
            assert (r["image"] == i).all() and np.array_equal(r["line_local"], np.arange(len(r)))
            if i not in (1, 4):
                assert np.array_equal(r, before[cu_k[i]:cu_k[i + 1]])
        assert np.array_equal(recs["first_tok"][:cu_k[-1]], np.concatenate([[0], np.cumsum(recs["n_tok"][:cu_k[-1]])[:-1]]))
    # a single image through linetr_prefilter: index 0; a tie-free call clears the report
    k, n = C.c_int32(), C.c_int32()
    one = np.zeros(60, dtype=nat.REC_DTYPE)
    nat.check(L.linetr_prefilter(nat.np_ptr(np.ascontiguousarray(imgs[4])), 60, H, W, 8, 16.0, -1, None, td, T, 0, 0, 0,
                                 nat.np_ptr(one), 60, C.byref(k), C.byref(n)))
    assert L.linetr_prefilter_tied_images(nat.np_ptr(tied), 6) == 1 and tied[0] == 0
    nat.check(L.linetr_prefilter(nat.np_ptr(np.ascontiguousarray(imgs[0])), 60, H, W, 8, 16.0, -1, None, td, T, 0, 0, 0,
                                 nat.np_ptr(one), 60, C.byref(k), C.byref(n)))
    assert L.linetr_prefilter_tied_images(nat.np_ptr(tied), 6) == 0
