"""Drop-in for the reference's models/nn_matcher.py: same function names, arguments and return values
(NumPy in, NumPy out), computed by the HIP matcher kernels (linetr_match_distmat / linetr_match_points).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _native as nat

_ws = {}


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("linetr_amd.nn_matcher needs a HIP device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _workspace(dev, nbytes):
    ws = _ws.get(dev)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.5) + 4096, dtype=torch.uint8, device=dev)
        _ws[dev] = ws
    return ws


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def match01_to_matrix(m01: np.ndarray, n1: int) -> np.ndarray:
    """index form -> the reference's [1,n0,n1] float64 0/1 matrix (nn_matcher.py:8,:29)."""
    n0 = len(m01)
    mat = np.zeros((1, n0, n1))
    rows = np.nonzero(m01 >= 0)[0]
    mat[0, rows, m01[rows]] = 1
    return mat


def nn_matcher_distmat(dist_mat, nn_thresh, is_mutual_NN=True):
    """Nearest-neighbour matching on a [1,n0,n1] distance matrix (reference: nn_matcher.py:3-31).

    The comparison runs in the matrix's own precision, as NumPy's does: a float32 matrix (what `Matching` passes) is compared in
    float32 against the float32 of `nn_thresh`, a float64 matrix in float64 against the Python float (linetr_match_distmat /
    linetr_match_distmat_f64); other dtypes are taken to float64.  The matrix must be NaN-free: np.argmin returns the first NaN, the
    kernels would skip it, so a NaN raises ValueError here instead of silently diverging."""
    dist_mat = np.asarray(dist_mat)
    n0, n1 = dist_mat.shape[1], dist_mat.shape[2]
    if n0 == 0 or n1 == 0:
        return np.zeros((1, n0, n1))
    if np.isnan(dist_mat).any():
        raise ValueError("nn_matcher_distmat: the distance matrix contains NaN")
    dev = _device()
    m01 = torch.empty((n0,), dtype=torch.int32, device=dev)
    L = nat.lib()
    if dist_mat.dtype == np.float32:
        d = torch.from_numpy(np.ascontiguousarray(dist_mat[0])).to(dev)
        ws = _workspace(dev, L.linetr_match_distmat_workspace_bytes(n0, n1))
        nat.check(L.linetr_match_distmat(None, d.data_ptr(), n0, n1, float(np.float32(nn_thresh)), int(bool(is_mutual_NN)),
                                         m01.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)))
    else:
        d = torch.from_numpy(np.ascontiguousarray(dist_mat[0], dtype=np.float64)).to(dev)
        ws = _workspace(dev, L.linetr_match_distmat_f64_workspace_bytes(n0, n1))
        nat.check(L.linetr_match_distmat_f64(None, d.data_ptr(), n0, n1, float(nn_thresh), int(bool(is_mutual_NN)), m01.data_ptr(),
                                             ws.data_ptr(), ws.numel(), _stream(dev)))
    return match01_to_matrix(m01.cpu().numpy(), n1)


def nn_matcher(desc0, desc1, nn_thresh=0.8, is_mutual_NN=True):
    """Nearest-neighbour matching of two [256,n] descriptor sets (reference: nn_matcher.py:33-42).
    Returns (mat_nn [1,n0,n1] float64, dist_mat [1,n0,n1] float32)."""
    desc0, desc1 = np.asarray(desc0), np.asarray(desc1)
    n0, n1 = desc0.shape[1], desc1.shape[1]
    if n0 == 0 or n1 == 0:
        return np.zeros((1, n0, n1)), np.zeros((1, n0, n1), dtype=np.float32)
    if desc0.shape[0] != 256:
        raise ValueError("linetr_amd.nn_matcher supports 256-d descriptors")
    dev = _device()
    d0 = torch.from_numpy(np.ascontiguousarray(desc0, dtype=np.float32)).to(dev)
    d1 = torch.from_numpy(np.ascontiguousarray(desc1, dtype=np.float32)).to(dev)
    dist = torch.empty((n0, n1), dtype=torch.float32, device=dev)
    m01 = torch.empty((n0,), dtype=torch.int32, device=dev)
    L = nat.lib()
    need = 4 * (n0 + n1) * 257 + 4096 + L.linetr_match_workspace_bytes(1, n0 * n1, 0, n0 + n1)
    ws = _workspace(dev, need)
    nat.check(L.linetr_match_points(None, d0.data_ptr(), n0, d1.data_ptr(), n1, float(np.float32(nn_thresh)),
                                    int(bool(is_mutual_NN)), dist.data_ptr(), m01.data_ptr(), ws.data_ptr(), ws.numel(),
                                    _stream(dev)))
    return match01_to_matrix(m01.cpu().numpy(), n1), dist.cpu().numpy()[None]
