"""Drop-in for the reference's ``models.line_process`` module: the same module-level names, arguments and return
values (``dataloaders/build_homography_dataset.py:19`` imports ``preprocess`` and ``line_tokenizer`` from it,
``models/line_transformer.py:6`` star-imports it).

O(K) glue that must reproduce NumPy's own ordering (``np.argsort`` tie order) stays NumPy on purpose; everything per token
-- ``line_tokenizer``, ``sample_descriptors``, ``get_dist_matrix`` -- runs in liblinetr_hip.so (no CPU path: without a HIP
device these raise).  Reference: models/line_process.py (line numbers cited per function).
"""
from __future__ import annotations

import itertools
import operator

import numpy as np
import torch

__all__ = ["filter_by_length", "get_line_dist", "get_angles", "point_on_line", "remove_borders", "sample_descriptors",
           "line_tokenizer", "get_dist_matrix", "change_cv2_T_np", "preprocess"]


# ------------------------------------------------------------------------------------------------ host glue (NumPy)

_KEYLINE_FIELDS = operator.attrgetter("startPointX", "startPointY", "endPointX", "endPointY", "lineLength", "octave")


def get_line_dist(line):
    """Euclidean length of one [2,2] line (line_process.py:23-26)."""
    d = np.asarray(line[1]) - np.asarray(line[0])
    return np.sqrt(np.sum(d ** 2))


def get_angles(lines):
    """(cos 2theta, sin 2theta) with theta = arctan2(dx, dy) folded into [0, pi) (line_process.py:28-41)."""
    if len(lines) == 0:
        return []
    theta = np.arctan2(lines[:, 1, 0] - lines[:, 0, 0], lines[:, 1, 1] - lines[:, 0, 1])
    theta = np.where(theta < 0, theta + np.pi, theta)
    return np.stack([np.cos(2 * theta), np.sin(2 * theta)], axis=1)


def point_on_line(line, dist_px):
    """The point at arclength `dist_px` from the start point, in the reference's slope form (line_process.py:43-57):
    float64, same operation order, same AssertionErrors (the device tokeniser walks lines with exactly this arithmetic,
    csrc/lt_token.h walk_along)."""
    assert dist_px >= 0, "distance should be positive!"
    assert get_line_dist(line) >= dist_px, "distance should be smaller than line length!"
    sp, ep = line
    v = ep - sp
    if v[0] != 0:
        slope = v[1] / v[0]
        dx = np.sqrt(dist_px ** 2 / (1 + slope ** 2))
        step = (dx, slope * dx)
    else:
        step = (0, dist_px if ep[1] - sp[1] > 0 else -dist_px)
    return step + sp


def change_cv2_T_np(klines_cv):
    """KeyLine objects -> {'klines' [K,2,2], 'length_klines' [K], 'angles' [K,2]} (float64; line_process.py:203-231)."""
    if len(klines_cv) == 0:
        return {"klines": np.zeros((0, 2, 2)), "length_klines": np.zeros((0,)), "angles": []}
    return lines_from_rows(np.fromiter(itertools.chain.from_iterable(map(_KEYLINE_FIELDS, klines_cv)), dtype=np.float64,
                                       count=6 * len(klines_cv)).reshape(-1, 6))


def lines_from_rows(raw, with_angles=True):
    """change_cv2_T_np for detector lines that are already [K,6] rows (startX, startY, endX, endY, lineLength, octave).
    with_angles=False leaves 'angles' empty ([K,0]) for callers that go on to filter_by_length, which recomputes them from the
    clipped, filtered, ordered lines anyway (line_process.py:20)."""
    if len(raw) == 0:
        return {"klines": np.zeros((0, 2, 2)), "length_klines": np.zeros((0,)), "angles": []}
    keep_order = raw[:, 0] < raw[:, 2]
    sp = np.where(keep_order[:, None], raw[:, 0:2], raw[:, 2:4])
    ep = np.where(keep_order[:, None], raw[:, 2:4], raw[:, 0:2])
    klines = np.stack([sp, ep], axis=1)
    return {"klines": klines, "length_klines": raw[:, 4] * np.exp2(raw[:, 5]),
            "angles": get_angles(klines) if with_angles else np.empty((len(klines), 0))}


def keylines_to_array(klines_cv) -> np.ndarray:
    """[K,6] float64 rows (startX, startY, endX, endY, lineLength, octave) -- the input format of the batched native
    pre-filter (linetr_prefilter_batch) -- from KeyLine-like objects."""
    if len(klines_cv) == 0:
        return np.zeros((0, 6), dtype=np.float64)
    return np.fromiter(itertools.chain.from_iterable(map(_KEYLINE_FIELDS, klines_cv)), dtype=np.float64,
                       count=6 * len(klines_cv)).reshape(-1, 6)


def remove_borders(lines, border, height, width, valid_mask_given=None):
    """line_process.py:59-84: strict-upper border test on both end points, in-place clip, ndarray masks honoured."""
    kl = lines["klines"]
    if len(kl) == 0:
        return lines
    xs, ys = kl[:, :, 0], kl[:, :, 1]
    ok = ((xs >= border) & (xs < width - border) & (ys >= border) & (ys < height - border)).all(axis=1)
    np.minimum(xs, width - 0.001 - border, out=xs)     # in place, like the reference
    np.minimum(ys, height - 0.001 - border, out=ys)
    if isinstance(valid_mask_given, np.ndarray):
        idx = np.floor(kl).astype(int)
        either = valid_mask_given[idx[:, 0, 1], idx[:, 0, 0]] + valid_mask_given[idx[:, 1, 1], idx[:, 1, 0]]
        ok &= either.astype(bool)
    return {k: v[ok] for k, v in lines.items()}


def filter_by_length(lines, min_length, max_sublines):
    """line_process.py:6-21: strict `>`, descending by reversed np.argsort, python slice [:max_sublines]."""
    sel = lines["length_klines"] > min_length
    kl, ln = lines["klines"][sel], lines["length_klines"][sel]
    order = np.argsort(ln)[::-1][:max_sublines]
    kl = kl[order]
    return {"klines": kl, "length_klines": ln[order], "angles": get_angles(kl)}


# ------------------------------------------------------------------------------------------------ device stages

_engines = {}


def _token_engine(device):
    """A weight-less engine (tokeniser / sampler / matcher entry points need no model) per device."""
    from .engine import Engine
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("linetr_amd.line_process runs on a HIP device only: pred_superpoint's dense maps must live on "
                           "'cuda' (there is no CPU fallback)")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    eng = _engines.get(dev)
    if eng is None:
        eng = _engines[dev] = Engine.heads_only(dev)
    return eng


def mask_fits_image(mask, height, width) -> bool:
    """True when an ndarray valid mask has the [height, width] layout the native pre-filter indexes (Engine.prefilter)."""
    return mask.ndim == 2 and mask.shape[0] == int(height) and mask.shape[1] == int(width)


def attach_sub2line(mat: torch.Tensor, sub2line: torch.Tensor) -> torch.Tensor:
    """The sub-line -> key-line map rides along with the mat_klines2sublines it describes, so that the matcher needs no pass over
    the matrix (Matching.match_lines).  It is stamped with the matrix's version counter: a matrix edited in place afterwards --
    through any view -- is read by its CONTENTS again (sub2line_of), as the reference multiplies whatever the matrix holds."""
    try:
        stamp = mat._version
    except RuntimeError:          # an inference tensor keeps no version counter: its map is never trusted
        stamp = None
    mat._linetr_sub2line = sub2line
    mat._linetr_sub2line_stamp = stamp
    return mat


def sub2line_of(mat):
    """The map attach_sub2line put on `mat`, or None when there is none or the matrix has been written to since."""
    s = getattr(mat, "_linetr_sub2line", None)
    if s is None:
        return None
    stamp = getattr(mat, "_linetr_sub2line_stamp", None)
    try:
        now = mat._version
    except RuntimeError:
        return None
    return s if stamp is not None and stamp == now else None


def prefilter_tokenize(klines_cv, height, width, border, min_length, max_sublines, token_distance, max_tokens, pred_superpoint,
                       mask=None):
    """change_cv2_T_np -> remove_borders -> filter_by_length -> line_tokenizer of ONE image (line_process.py:203-231, :59-84, :6-21,
    :100-196) with a1-a3 on the native pre-filter: bit-identical to the NumPy functions above on klines / lengths / order (equal
    lengths: NumPy's own argsort decides, Engine.prefilter tie_order), ~20 us instead of ~75.  The angles are NumPy's, computed on
    the filtered lines as filter_by_length does (:20): libm's cos / sin may differ from NumPy's in the last ulp.  An ndarray mask
    is honoured, anything else ignored (:76).  Returns the reference's dict (the NumPy functions' own, empty one when no line
    survives)."""
    if len(klines_cv) == 0:
        return filter_by_length(remove_borders(change_cv2_T_np(klines_cv), border, height, width, mask), min_length, max_sublines)
    dd = pred_superpoint.get("dense_descriptor_nhwc")
    if dd is None:
        dd = pred_superpoint["dense_descriptor"]
    eng = _token_engine(dd.device)
    if isinstance(mask, np.ndarray) and not mask_fits_image(mask, height, width):
        # the native pre-filter addresses the mask as [height, width]; a mask of any other shape goes through the NumPy indexing of
        # remove_borders, which does with it exactly what the reference does (IndexError, or whatever rows it selects)
        lines = filter_by_length(remove_borders(change_cv2_T_np(klines_cv), border, height, width, mask), min_length, max_sublines)
        return lines if len(lines["klines"]) == 0 else tokenize_into(lines, eng, token_distance, max_tokens, pred_superpoint)
    vm = [mask] if isinstance(mask, np.ndarray) else None
    recs, cu_k, cu_n = eng.prefilter([keylines_to_array(klines_cv)], height, width, remove_borders=border, min_length=min_length,
                                     max_keylines=max_sublines, token_distance=token_distance, max_tokens=max_tokens, valid_masks=vm)
    if cu_k[1] == 0:
        return filter_by_length(remove_borders(change_cv2_T_np(klines_cv), border, height, width, mask), min_length, max_sublines)
    recs["angle"] = get_angles(np.stack([recs["sp"], recs["ep"]], axis=1))
    return tokenize_into({}, eng, token_distance, max_tokens, pred_superpoint, packed=(recs, int(cu_n[1])))


def tokenize_into(klines, eng, token_distance, max_tokens, pred_superpoint, packed=None, clip_shape=None):
    """line_tokenizer body shared by the module-level function and LineTransformer.preprocess: fills the reference's dict
    (line_process.py:182-196; 11 tensor entries with a leading batch axis of 1) from ONE native tokeniser call.
    `packed`: (records, number of sub-lines) when the lines are already packed (prefilter_tokenize); `klines` is then the empty
    dict to fill."""
    ds = pred_superpoint["dense_score"]
    # a producer that also hands out the channel-last map (linetr_amd.superpoint.FusedHeadSuperPoint) saves the
    # NCHW -> NHWC pass; the reference's key is used otherwise
    layout = "nhwc" if pred_superpoint.get("dense_descriptor_nhwc") is not None else "nchw"
    dd = pred_superpoint["dense_descriptor_nhwc" if layout == "nhwc" else "dense_descriptor"]
    td, T = token_distance, max_tokens
    recs, N = packed if packed is not None else eng.pack(klines["klines"], klines["length_klines"], klines["angles"], td, T)
    K = len(recs)
    align = int(torch.__version__[2]) > 2   # the reference's own version switch (line_process.py:93)
    tb = eng.tokenize(recs, np.array([0, K], np.int32), np.array([0, N], np.int32), dd, ds, token_distance=td,
                      max_tokens=T, align_corners=align, dense_layout=layout, want_mat=True, clip_shape=clip_shape)
    # the sub-line -> key-line map rides along with the matrix it describes (Matching.match_lines); a matrix from anywhere else -- or
    # one indexed / edited since -- simply lacks a valid one and is reduced to its map on the device (linetr_pool_distmat_dense)
    mat = attach_sub2line(tb.mat[None], tb.sub2line)
    # the reference clips the end points through a view, so the exported key-lines carry the clip
    klines["klines"] = tb.klines[None]
    klines["length_klines"] = tb.length[None]
    klines["angles"] = tb.angles[None]
    klines["sublines"] = tb.sublines[None]
    klines["pnt_sublines"] = tb.pnt[None]
    klines["mask_sublines"] = tb.mask[None, :, :, None]
    klines["resp_sublines"] = tb.resp[None, :, None]
    klines["angle_sublines"] = tb.angle_sub[None]
    klines["desc_sublines"] = tb.desc[None]
    klines["score_sublines"] = tb.score[None, :, :, None]
    klines["mat_klines2sublines"] = mat
    return klines


def line_tokenizer(klines, token_distance, max_tokens, pred_superpoint, image_shape):
    """line_process.py:100-196.  `klines`: {'klines' [K,2,2], 'length_klines' [K], 'angles' [K,2]} float64 NumPy;
    `image_shape` = (height, width): as in the reference it only sets the end-point clip (width - 0.6, height - 0.6, :115-116) and
    need not be the maps' shape -- conv_fixed_size hands over conf['data']['resize'] = (640, 480) for 480 x 640 images
    (dataloaders/utils/util_lines.py:682,703).  Scores and descriptors are gathered within the maps' own bounds (:174-179)."""
    height, width = image_shape
    dd = pred_superpoint.get("dense_descriptor_nhwc")       # either descriptor key may be the one supplied (tokenize_into)
    if dd is None:
        dd = pred_superpoint["dense_descriptor"]
    eng = _token_engine(dd.device)
    return tokenize_into(klines, eng, token_distance, max_tokens, pred_superpoint, clip_shape=(int(height), int(width)))


def preprocess(klines_cv, image_shape, pred_superpoint, mask=None, conf={}):
    """line_process.py:233-260 (the dataset builder's entry point; the model's own is LineTransformer.preprocess)."""
    conf = {"min_length": 16, "max_sublines": 256, "token_distance": 8, "max_tokens": 21, "remove_borders": 0, **conf}
    height, width = image_shape
    ds = pred_superpoint["dense_score"]
    if (int(ds.shape[-2]), int(ds.shape[-1])) != (int(height), int(width)):
        raise ValueError(f"preprocess: image_shape {tuple(image_shape)} does not match dense_score {tuple(ds.shape)}")
    return prefilter_tokenize(klines_cv, height, width, conf["remove_borders"], conf["min_length"], conf["max_sublines"],
                              conf["token_distance"], conf["max_tokens"], pred_superpoint, mask)


def sample_descriptors(keypoints, descriptors, s: int = 8):
    """Bilinear sampling + L2 normalisation of dense descriptors at key-point locations (line_process.py:86-98).
    keypoints [b, ..., 2] pixel coordinates (x, y), descriptors [b,256,h,w] -> [b,256,n].  Native for the cell size the
    path uses (s = 8) and 256 channels."""
    b, c, h, w = descriptors.shape
    if s != 8 or c != 256:
        raise ValueError("linetr_amd.sample_descriptors supports s=8 and 256-channel maps (the Line-Transformer path)")
    eng = _token_engine(descriptors.device)
    align = int(torch.__version__[2]) > 2
    pts = keypoints.reshape(b, -1, 2)
    return torch.stack([eng.sample_descriptors(pts[i], descriptors[i], align_corners=align).t() for i in range(b)])


def get_dist_matrix(desc0, desc1):
    """[b,256,N0],[b,256,N1] NumPy -> clip(2 - 2 d0^T d1, 0) [b,N0,N1] float32, on the HIP matcher kernel
    (line_process.py:198-201)."""
    from .nn_matcher import nn_matcher
    desc0, desc1 = np.asarray(desc0), np.asarray(desc1)
    return np.concatenate([nn_matcher(desc0[b], desc1[b], np.inf, False)[1] for b in range(desc0.shape[0])], 0)
