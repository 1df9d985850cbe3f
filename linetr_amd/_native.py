"""ctypes binding of liblinetr_hip.so (the C ABI of include/linetr_hip.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblinetr_hip.so")
# the product sources + experiments/csrc built with -DLINETR_EXPERIMENTS (`python -m linetr_amd.build --experiments`): tuning
# switches and the kernels that were measured and lost.  Lives OUTSIDE the package; only tools/ and `pytest -m experiments` load it.
EXPERIMENTS_LIB_PATH = os.path.join(os.path.dirname(_HERE), "experiments", "liblinetr_hip_experiments.so")

E_ASSERT = -4


class ModelConfig(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("n_heads", C.c_int32), ("d_inner", C.c_int32),
                ("n_sig_layers", C.c_int32), ("n_desc_layers", C.c_int32), ("enc_channels", C.c_int32 * 4),
                ("norm_height", C.c_int32), ("norm_width", C.c_int32), ("bn_batch_stats", C.c_int32)]


class LineRec(C.Structure):
    _fields_ = [("sp", C.c_double * 2), ("ep", C.c_double * 2), ("length", C.c_double),
                ("angle", C.c_double * 2), ("first_sub", C.c_int32), ("n_tok", C.c_int32),
                ("n_sub", C.c_int32), ("image", C.c_int32), ("line_local", C.c_int32), ("first_tok", C.c_int32)]


REC_DTYPE = np.dtype([("sp", "<f8", 2), ("ep", "<f8", 2), ("length", "<f8"), ("angle", "<f8", 2),
                      ("first_sub", "<i4"), ("n_tok", "<i4"), ("n_sub", "<i4"), ("image", "<i4"),
                      ("line_local", "<i4"), ("first_tok", "<i4")])
assert REC_DTYPE.itemsize == C.sizeof(LineRec) == 80


class Tokens(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("klines", "length", "angles", "sublines", "pnt", "mask", "resp",
                                          "angle_sub", "desc", "score", "mat", "h_cu_klines")]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("calls", C.c_int32), ("ms", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


_libs = {}


def lib(path=None):
    """Load the in-tree shared library (built by ``python -m linetr_amd.build`` / __graft_entry__.build()).
    `path`: another build of the same sources (EXPERIMENTS_LIB_PATH, a tools/ variant); default = the product library,
    or $LINETR_LIB when set."""
    # torch ships its own libamdhip64; importing it first makes liblinetr_hip.so bind to the SAME HIP runtime
    # (one context, shared streams and device pointers) instead of a second copy from /opt/rocm.
    import torch  # noqa: F401
    path = os.path.abspath(path or os.environ.get("LINETR_LIB", LIB_PATH))
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built (run `python -m linetr_amd.build`). "
            "linetr_amd has no CPU fallback.")
    L = C.CDLL(path)
    vp, i32, i64, f64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_float
    L.linetr_abi_version.restype = i32
    L.linetr_last_error.restype = C.c_char_p
    L.linetr_create.argtypes = [C.POINTER(ModelConfig), i32, C.POINTER(C.c_char_p), C.POINTER(vp), C.POINTER(i64), i32,
                                C.POINTER(vp)]
    L.linetr_destroy.argtypes = [vp]
    L.linetr_destroy.restype = None
    L.linetr_prefilter.argtypes = [vp, i32, i32, i32, i32, f64, i32, vp, f64, i32, i32, i32, i32, vp, i32, C.POINTER(i32),
                                   C.POINTER(i32)]
    L.linetr_prefilter_batch.argtypes = [vp, vp, i32, i32, i32, i32, f64, i32, vp, f64, i32, i32, vp, i32, vp, vp]
    L.linetr_prefilter_tied_images.argtypes = [vp, i32]
    L.linetr_prefilter_tied_images.restype = i32
    L.linetr_pack_lines.argtypes = [vp, vp, vp, i32, f64, i32, i32, i32, i32, vp, C.POINTER(i32)]
    L.linetr_describe_workspace_bytes.argtypes = [vp, i32, i32, i32, i32, i64]
    L.linetr_describe_workspace_bytes.restype = i64
    L.linetr_describe.argtypes = [vp, vp, i32, i32, i64, vp, vp, i32, f64, i32, vp, vp, i32, i32, i32, i32, Tokens, vp, vp,
                                  vp, i64, vp]
    L.linetr_describe_submit.argtypes = [vp, vp, i32, i32, i64, vp, vp, i32, f64, i32, vp, vp, i32, i32, i32, i32, Tokens, vp, vp,
                                         vp, i64, i32, i32, vp]
    L.linetr_describe_join.argtypes = [vp, i32, vp]
    L.linetr_pipeline_max_slots.argtypes = []
    L.linetr_tokenize_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.linetr_tokenize_workspace_bytes.restype = i64
    L.linetr_tokenize.argtypes = [vp, vp, i32, i32, f64, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, Tokens, vp, vp, i64, vp]
    L.linetr_forward_workspace_bytes.argtypes = [vp, i32, i32]
    L.linetr_forward_workspace_bytes.restype = i64
    L.linetr_forward.argtypes = [vp, C.POINTER(Tokens), vp, vp, i32, i32, vp, vp, i64, vp]
    L.linetr_bn_stats_floats.argtypes = [vp]
    L.linetr_bn_stats_floats.restype = i64
    L.linetr_forward_train_workspace_bytes.argtypes = [vp, i32, i32]
    L.linetr_forward_train_workspace_bytes.restype = i64
    L.linetr_forward_train.argtypes = [vp, C.POINTER(Tokens), vp, vp, i32, i32, f32, vp, vp, vp, vp, i64, vp]
    L.linetr_match_workspace_bytes.argtypes = [i32, i64, i64, i64]
    L.linetr_match_workspace_bytes.restype = i64
    L.linetr_match.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, i64, vp]
    L.linetr_match_gathered.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, i64, vp]
    L.linetr_match_points.argtypes = [vp, vp, i32, vp, i32, f32, i32, vp, vp, vp, i64, vp]
    L.linetr_match_distmat.argtypes = [vp, vp, i32, i32, f32, i32, vp, vp, i64, vp]
    L.linetr_match_distmat_workspace_bytes.argtypes = [i32, i32]
    L.linetr_match_distmat_workspace_bytes.restype = i64
    L.linetr_match_distmat_f64.argtypes = [vp, vp, i32, i32, f64, i32, vp, vp, i64, vp]
    L.linetr_match_distmat_f64_workspace_bytes.argtypes = [i32, i32]
    L.linetr_match_distmat_f64_workspace_bytes.restype = i64
    L.linetr_pair_tail_workspace_bytes.argtypes = [i32, i32, i32, i32, i32, i32]
    L.linetr_pair_tail_workspace_bytes.restype = i64
    L.linetr_pair_tail_output_bytes.argtypes = [i32, i32, i32, i32, vp]
    L.linetr_pair_tail_output_bytes.restype = i64
    L.linetr_pair_tail.argtypes = [vp, vp, i32, vp, i32, f32, vp, i32, vp, i32, vp, i32, vp, i32, f32, i32, vp, i64, vp, i64, vp]
    L.linetr_superpoint_heads.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.linetr_set_precision.argtypes = [vp, i32]
    L.linetr_get_precision.argtypes = [vp]
    L.linetr_debug_posenc.argtypes = [vp, i32, vp, vp, vp, i64, vp, vp]
    L.linetr_debug_gemm.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    if hasattr(L, "linetr_debug_gemm_st"):      # experiments build only (include/linetr_hip.h, LINETR_EXPERIMENTS)
        L.linetr_st_bytes.argtypes = [i64, i32]
        L.linetr_st_bytes.restype = i64
        L.linetr_debug_to_st.argtypes = [vp, vp, i32, i32, i32, vp, vp]
        L.linetr_debug_from_st.argtypes = [vp, vp, i32, i32, vp, i32, vp]
        L.linetr_debug_gemm_st.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        L.linetr_debug_pairnet_stamps.argtypes = [vp, vp]
    L.linetr_allgather_desc.argtypes = [vp, vp, vp, i64, vp]
    L.linetr_set_allgather_fn.argtypes = [vp]
    L.linetr_pack_slab.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, i32, vp, vp]
    L.linetr_sample_descriptors_workspace_bytes.argtypes = [i32, i32, i32]
    L.linetr_sample_descriptors_workspace_bytes.restype = i64
    L.linetr_sample_descriptors.argtypes = [vp, vp, i64, vp, i32, i32, i32, i32, vp, vp, i64, vp]
    L.linetr_pool_distmat_workspace_bytes.argtypes = [i32, i32]
    L.linetr_pool_distmat_workspace_bytes.restype = i64
    L.linetr_pool_distmat.argtypes = [vp, vp, i32, i32, vp, i32, vp, i32, vp, vp, i64, vp]
    L.linetr_pool_distmat_dense_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.linetr_pool_distmat_dense_workspace_bytes.restype = i64
    L.linetr_pool_distmat_dense.argtypes = [vp, vp, i32, i32, vp, i32, vp, i32, vp, vp, i64, vp]
    L.linetr_set_profiling.argtypes = [vp, i32]
    L.linetr_get_profile.argtypes = [vp, C.POINTER(ProfileEntry), i32, C.POINTER(i32)]
    if L.linetr_abi_version() != 5:
        raise RuntimeError("liblinetr_hip.so ABI version mismatch")
    _libs[path] = L
    return L


EXPORTS = ["linetr_abi_version", "linetr_last_error", "linetr_create", "linetr_destroy", "linetr_prefilter",
           "linetr_prefilter_batch", "linetr_prefilter_tied_images", "linetr_pack_lines", "linetr_tokenize_workspace_bytes", "linetr_tokenize", "linetr_forward_workspace_bytes",
           "linetr_forward", "linetr_bn_stats_floats", "linetr_forward_train_workspace_bytes", "linetr_forward_train", "linetr_describe_workspace_bytes", "linetr_describe", "linetr_describe_submit", "linetr_describe_join", "linetr_pipeline_max_slots", "linetr_match_workspace_bytes", "linetr_match", "linetr_match_gathered", "linetr_match_points",
           "linetr_match_distmat", "linetr_match_distmat_workspace_bytes", "linetr_match_distmat_f64", "linetr_match_distmat_f64_workspace_bytes", "linetr_pair_tail_workspace_bytes", "linetr_pair_tail_output_bytes", "linetr_pair_tail", "linetr_superpoint_heads", "linetr_set_precision", "linetr_get_precision", "linetr_debug_posenc", "linetr_debug_gemm", "linetr_allgather_desc", "linetr_set_allgather_fn", "linetr_pack_slab", "linetr_sample_descriptors_workspace_bytes",
           "linetr_sample_descriptors", "linetr_pool_distmat_workspace_bytes", "linetr_pool_distmat", "linetr_pool_distmat_dense_workspace_bytes", "linetr_pool_distmat_dense", "linetr_set_profiling", "linetr_get_profile"]


EXPERIMENT_EXPORTS = ["linetr_st_bytes", "linetr_debug_to_st", "linetr_debug_from_st", "linetr_debug_gemm_st", "linetr_debug_pairnet_stamps"]


class NativeError(RuntimeError):
    pass


def check(code: int, L=None):
    """`L`: the library the failing call went through (every .so keeps its own thread-local error text); default = the
    product library."""
    if code == 0:
        return
    msg = (L if L is not None else lib()).linetr_last_error().decode("utf-8", "replace")
    if code == E_ASSERT:
        raise AssertionError(msg)     # same exception type the reference raises (line_process.py:44-45)
    raise NativeError(f"liblinetr_hip error {code}: {msg}")


def np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)
