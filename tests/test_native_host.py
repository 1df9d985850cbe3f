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
    declared = set(re.findall(r"\b(linetr_[a-z_]+)\s*\(", product_hdr))
    assert declared == set(nat.EXPORTS), declared ^ set(nat.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    declared_x = set(re.findall(r"\b(linetr_[a-z_]+)\s*\(", exp.group(1)))
    assert declared_x == set(nat.EXPERIMENT_EXPORTS), declared_x ^ set(nat.EXPERIMENT_EXPORTS)
    for name in declared_x:
        assert not hasattr(L, name), f"{name} must not be in the product library"
    if os.path.exists(nat.EXPERIMENTS_LIB_PATH):
        LX = nat.lib(nat.EXPERIMENTS_LIB_PATH)
        for name in declared | declared_x:
            assert hasattr(LX, name), name
    assert L.linetr_abi_version() == 3
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
