"""Deterministic synthetic inputs for the LineTR hot path.

The reference's pretrained blobs (models/weights/LineTR_weight.pth,
superpoint_v1.pth) are absent from the checkout and cv2/LSD is not installed,
so every test, golden fixture and benchmark here runs on *seeded* weights,
KeyLine stand-ins and dense maps.  Everything is generated from
``numpy.random.RandomState`` (bit-stable across numpy versions) except the
cfg2/cfg3 dense maps, which follow BASELINE.md §3's recipe
(``torch.Generator().manual_seed(seed)`` on CPU).

Nothing in this module touches the GPU or the oracle.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

# ----------------------------------------------------------------------------
# KeyLine stand-in (the attributes change_cv2_T_np reads:
# /root/reference/models/line_process.py:206-220)
# ----------------------------------------------------------------------------


class KeyLine:
    """Minimal stand-in for cv2.line_descriptor.KeyLine (attrs are Python floats of f32 values)."""

    __slots__ = ("startPointX", "startPointY", "endPointX", "endPointY", "lineLength", "octave")

    def __init__(self, sx, sy, ex, ey, length=None, octave=0):
        self.startPointX = float(np.float32(sx))
        self.startPointY = float(np.float32(sy))
        self.endPointX = float(np.float32(ex))
        self.endPointY = float(np.float32(ey))
        if length is None:
            length = np.float32(math.hypot(self.endPointX - self.startPointX,
                                           self.endPointY - self.startPointY))
        self.lineLength = float(np.float32(length))
        self.octave = int(octave)

    def as_row(self):
        return [self.startPointX, self.startPointY, self.endPointX, self.endPointY,
                self.lineLength, float(self.octave)]


def keylines_to_array(klines_cv) -> np.ndarray:
    """[K,6] float64 rows (spx, spy, epx, epy, lineLength, octave) from KeyLine-like objects."""
    if len(klines_cv) == 0:
        return np.zeros((0, 6), dtype=np.float64)
    return np.asarray([[l.startPointX, l.startPointY, l.endPointX, l.endPointY,
                        l.lineLength, float(l.octave)] for l in klines_cv], dtype=np.float64)


def array_to_keylines(arr) -> list:
    return [KeyLine(r[0], r[1], r[2], r[3], r[4], int(r[5])) for r in np.asarray(arr)]


def synth_lines(seed: int, n_lines: int = 200, height: int = 480, width: int = 640,
                len_lo: float = 17.0, len_hi: float = 167.0, margin: float = 10.0) -> np.ndarray:
    """BASELINE.md §3 cfg2 recipe: start ~U(margin box), length ~U(len_lo,len_hi), angle ~U(0,2pi),
    rejected unless the end point also lies inside the margin; coords rounded to f32, octave 0,
    lineLength = f32(hypot).  Returns [n_lines,6] float64 (values exactly representable in f32)."""
    rs = np.random.RandomState(seed)
    rows = []
    while len(rows) < n_lines:
        sx = rs.uniform(margin, width - margin)
        sy = rs.uniform(margin, height - margin)
        ln = rs.uniform(len_lo, len_hi)
        th = rs.uniform(0.0, 2.0 * math.pi)
        ex = sx + ln * math.cos(th)
        ey = sy + ln * math.sin(th)
        if not (margin <= ex <= width - margin and margin <= ey <= height - margin):
            continue
        sx, sy, ex, ey = (float(np.float32(v)) for v in (sx, sy, ex, ey))
        length = float(np.float32(math.hypot(ex - sx, ey - sy)))
        rows.append([sx, sy, ex, ey, length, 0.0])
    return np.asarray(rows, dtype=np.float64)


def synth_dense_maps(seed: int, height: int = 480, width: int = 640, dim: int = 256):
    """cfg2 recipe: dense_descriptor = normalize(randn(1,256,H/8,W/8)), dense_score = rand(1,H,W)
    from torch.Generator().manual_seed(seed) (CPU tensors, f32)."""
    import torch
    g = torch.Generator().manual_seed(int(seed))
    dd = torch.randn(1, dim, height // 8, width // 8, generator=g)
    dd = torch.nn.functional.normalize(dd, p=2, dim=1)
    ds = torch.rand(1, height, width, generator=g)
    return dd, ds


def synth_dense_maps_np(seed: int, height: int, width: int, dim: int = 256):
    """numpy-only variant (used by the committed golden fixtures so they never depend on torch's RNG)."""
    rs = np.random.RandomState(seed)
    dd = rs.standard_normal((1, dim, height // 8, width // 8)).astype(np.float32)
    dd /= np.maximum(np.sqrt((dd.astype(np.float64) ** 2).sum(1, keepdims=True)), 1e-12).astype(np.float32)
    ds = rs.uniform(0.0, 1.0, (1, height, width)).astype(np.float32)
    return dd, ds


def jitter_pair(lines: np.ndarray, seed: int, sigma: float = 0.3):
    """A 'matching-meaningful' second view: a random permutation of the same lines with sub-pixel
    endpoint jitter.  Returns (lines1[K,6], perm) with lines1[i] ~ lines[perm[i]]."""
    rs = np.random.RandomState(seed)
    perm = rs.permutation(len(lines))
    out = lines[perm].copy()
    out[:, :4] += rs.normal(0.0, sigma, (len(lines), 4))
    out[:, :4] = out[:, :4].astype(np.float32).astype(np.float64)
    out[:, 4] = np.hypot(out[:, 2] - out[:, 0], out[:, 3] - out[:, 1]).astype(np.float32).astype(np.float64)
    return out, perm


# ----------------------------------------------------------------------------
# Seeded state_dict with the layout of SURVEY.md Appendix B
# (/root/reference/models/line_transformer.py:207-218 builds these modules)
# ----------------------------------------------------------------------------


def state_dict_spec(n_desc_layers: int = 1, d: int = 256, enc=(32, 64, 128, 256), d_inner: int = 1024,
                    n_sig_layers: int = 7):
    """Ordered (name, shape, kind) list in torch's state_dict order."""
    spec = [("klenc.cls_token", (1, 1, 1, d), "randn")]

    def mlp(prefix, chans):
        n = len(chans)
        idx = 0
        for i in range(1, n):
            spec.append((f"{prefix}.{idx}.weight", (chans[i], chans[i - 1], 1), "w"))
            spec.append((f"{prefix}.{idx}.bias", (chans[i],), "b_last" if i == n - 1 else "b"))
            idx += 1
            if i < n - 1:
                spec.append((f"{prefix}.{idx}.weight", (chans[i],), "gamma"))
                spec.append((f"{prefix}.{idx}.bias", (chans[i],), "beta"))
                spec.append((f"{prefix}.{idx}.running_mean", (chans[i],), "rmean"))
                spec.append((f"{prefix}.{idx}.running_var", (chans[i],), "rvar"))
                spec.append((f"{prefix}.{idx}.num_batches_tracked", (), "nbt"))
                idx += 2  # BN + ReLU

    mlp("klenc.line_position_enc.encoder", [5, *enc, d])
    mlp("klenc.word_position_enc.encoder", [3, *enc, d])
    for i in range(n_desc_layers):
        p = f"klenc.desc_layers.{i}"
        for nm in ("w_qs", "w_ks", "w_vs", "fc"):
            spec.append((f"{p}.slf_attn.{nm}.weight", (d, d), "w"))
            spec.append((f"{p}.slf_attn.{nm}.bias", (d,), "b"))
        spec.append((f"{p}.slf_attn.layer_norm.weight", (d,), "gamma"))
        spec.append((f"{p}.slf_attn.layer_norm.bias", (d,), "beta"))
        spec.append((f"{p}.pos_ffn.w_1.weight", (d_inner, d), "w"))
        spec.append((f"{p}.pos_ffn.w_1.bias", (d_inner,), "b"))
        spec.append((f"{p}.pos_ffn.w_2.weight", (d, d_inner), "w"))
        spec.append((f"{p}.pos_ffn.w_2.bias", (d,), "b"))
        spec.append((f"{p}.pos_ffn.layer_norm.weight", (d,), "gamma"))
        spec.append((f"{p}.pos_ffn.layer_norm.bias", (d,), "beta"))
    for l in range(n_sig_layers):
        p = f"selfattn.layers.{l}"
        spec.append((f"{p}.attn.merge.weight", (d, d, 1), "w"))
        spec.append((f"{p}.attn.merge.bias", (d,), "b"))
        for j in range(3):
            spec.append((f"{p}.attn.proj.{j}.weight", (d, d, 1), "w"))
            spec.append((f"{p}.attn.proj.{j}.bias", (d,), "b"))
        mlp(f"{p}.mlp", [2 * d, 2 * d, d])
    spec.append(("final_proj.weight", (d, d, 1), "w"))
    spec.append(("final_proj.bias", (d,), "b"))
    return spec


def make_state_dict(seed: int = 0, n_desc_layers: int = 1, gain: float = 1.0, **kw) -> "OrderedDict[str, np.ndarray]":
    """Seeded weights as numpy arrays keyed like LineTransformer.state_dict().

    Conv/Linear weights & biases ~ U(-g/sqrt(fan_in), g/sqrt(fan_in)) (torch's default init scale),
    BN running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5), affine gamma ~ U(0.5,1.5), beta ~ N(0,0.1)
    so BN folding and LayerNorm affine are genuinely exercised; last-MLP biases are NOT zeroed
    (a trained checkpoint would not have them at zero either)."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for name, shape, kind in state_dict_spec(n_desc_layers, **kw):
        if kind == "randn":
            v = rs.standard_normal(shape)
        elif kind == "w":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            b = gain / math.sqrt(fan_in)
            v = rs.uniform(-b, b, shape)
        elif kind in ("b", "b_last"):
            v = rs.uniform(-0.1, 0.1, shape)
        elif kind == "gamma":
            v = rs.uniform(0.5, 1.5, shape)
        elif kind == "beta":
            v = rs.normal(0.0, 0.1, shape)
        elif kind == "rmean":
            v = rs.normal(0.0, 0.1, shape)
        elif kind == "rvar":
            v = rs.uniform(0.5, 1.5, shape)
        elif kind == "nbt":
            sd[name] = np.asarray(100, dtype=np.int64)
            continue
        else:
            raise ValueError(kind)
        sd[name] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def to_torch_state_dict(sd):
    import torch
    return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in sd.items())


def calibrated_state_dict(n_desc_layers: int = 1) -> "OrderedDict[str, np.ndarray]":
    """seed-0 weights with BatchNorm running stats overlaid from data/bn_calib_seed0.npz
    (produced by tests/golden/make_calib.py).  With matched BN statistics the descriptors of
    different lines are well separated (pairwise distances 0.06..1.5, argmin margins >> 1e-4), as a
    trained checkpoint's would be, which makes index-exact match parity a meaningful test."""
    import os
    sd = make_state_dict(0, n_desc_layers)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bn_calib_seed0.npz")
    with np.load(path) as cal:
        for k in cal.files:
            assert k in sd and sd[k].shape == cal[k].shape, k
            sd[k] = cal[k].astype(np.float32)
    return sd


# ---- SuperPoint (only for the fused head producer) -------------------------

SUPERPOINT_LAYERS = (  # name, out channels, in channels, kernel   (models/superpoint.py:117-134)
    ("conv1a", 64, 1, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 128, 64, 3), ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 256, 128, 3), ("convPb", 65, 256, 1), ("convDa", 256, 128, 3), ("convDb", 256, 256, 1),
)


def superpoint_state_dict(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Seeded SuperPoint weights in the reference's state_dict layout (the authors' blob is not redistributable
    here).  He-scaled normals keep activations O(1); the score head gets a larger gain so the softmax is peaked
    enough for some cells to pass the 0.005 keypoint threshold."""
    rs = np.random.RandomState(1234 + seed)
    sd = OrderedDict()
    for name, co, ci, k in SUPERPOINT_LAYERS:
        gain = 3.0 if name == "convPb" else 1.0
        std = gain * math.sqrt(2.0 / (ci * k * k))
        sd[name + ".weight"] = (rs.standard_normal((co, ci, k, k)) * std).astype(np.float32)
        sd[name + ".bias"] = (rs.standard_normal(co) * 0.05).astype(np.float32)
    return sd


# ----------------------------------------------------------------------------
# NOTE ON PROVENANCE: `sample_homography` below follows the reference's sampler STEP FOR STEP -- same argument list, same order of
# perspective -> scaling -> translation -> rotation, same validity rules -- because SURVEY.md (section 2 row 13, section 8(d) cfg4) prescribes
# that recipe as the definition of the cfg4 workload.  It is WORKLOAD GENERATION for bench.py and the tests, runs outside every timed
# region, and is not part of the product (`linetr_amd/`).  The RNG (a seeded RandomState instead of the global one), the truncated normal
# (rejection instead of scipy.stats) and the DLT solve (instead of cv2.getPerspectiveTransform) are this repo's own.
# cfg4: homography-augmented pairs.  Restates the recipe of the reference's dataset builder
# (/root/reference/dataloaders/utils/homographies.py:12-141, called with the parameters of
# /root/reference/dataloaders/confs/homography.yaml:31-46 on the normalised [-1,1]^2 square and rescaled to pixels as
# /root/reference/dataloaders/build_homography_dataset.py:122-145 does) on a seeded RandomState.  Image 1 is image 0
# seen through the homography: line end points are warped, lines leaving the image are dropped.
# ----------------------------------------------------------------------------

HOMOGRAPHY_PARAMS = dict(perspective=True, scaling=True, translation=True, rotation=True, patch_ratio=0.85,
                         perspective_amplitude_x=0.2, perspective_amplitude_y=0.2, scaling_amplitude=0.2,
                         max_angle=1.0472, allow_artifacts=True)


def _truncnorm(rs, scale, size, loc=0.0, std_trunc=2.0):
    """N(loc, scale) truncated to +-std_trunc sigma (scipy.stats.truncnorm(-2, 2, loc, scale)) by rejection."""
    out = np.empty(size, dtype=np.float64)
    i = 0
    while i < size:
        v = rs.standard_normal()
        if abs(v) <= std_trunc:
            out[i] = loc + scale * v
            i += 1
    return out


def _perspective_transform(src, dst):
    """3x3 H with dst ~ H src for four point pairs (cv2.getPerspectiveTransform on float32-rounded points)."""
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    a, b = [], []
    for (x, y), (u, v) in zip(src, dst):
        a.append([x, y, 1, 0, 0, 0, -x * u, -y * u]); b.append(u)
        a.append([0, 0, 0, x, y, 1, -x * v, -y * v]); b.append(v)
    h = np.linalg.solve(np.asarray(a), np.asarray(b))
    return np.append(h, 1.0).reshape(3, 3)


def sample_homography(rs, shape=(2.0, 2.0), shift=-1.0, perspective=True, scaling=True, rotation=True,
                      translation=True, n_scales=5, n_angles=25, scaling_amplitude=0.1, perspective_amplitude_x=0.1,
                      perspective_amplitude_y=0.1, patch_ratio=0.5, max_angle=math.pi / 2, allow_artifacts=False,
                      translation_overflow=0.0):
    """A random homography between the unit square's corners and a perturbed centred patch
    (homographies.py:48-141: perspective -> scaling -> translation -> rotation, in that order, with the same
    validity rules; `rs` replaces the global numpy RNG)."""
    pts1 = np.array([[0., 0.], [0., 1.], [1., 1.], [1., 0.]])
    margin = (1 - patch_ratio) / 2
    pts2 = margin + np.array([[0, 0], [0, patch_ratio], [patch_ratio, patch_ratio], [patch_ratio, 0]], dtype=np.float64)
    if perspective:
        if not allow_artifacts:
            perspective_amplitude_x = min(perspective_amplitude_x, margin)
            perspective_amplitude_y = min(perspective_amplitude_y, margin)
        pd = _truncnorm(rs, perspective_amplitude_y / 2, 1)[0]
        hl = _truncnorm(rs, perspective_amplitude_x / 2, 1)[0]
        hr = _truncnorm(rs, perspective_amplitude_x / 2, 1)[0]
        pts2 = pts2 + np.array([[hl, pd], [hl, -pd], [hr, pd], [hr, -pd]])
    if scaling:
        scales = np.concatenate(([1.0], _truncnorm(rs, scaling_amplitude / 2, n_scales, loc=1.0)))
        center = pts2.mean(axis=0, keepdims=True)
        scaled = (pts2 - center)[None] * scales[:, None, None] + center
        if allow_artifacts:
            valid = np.arange(n_scales)
        else:
            valid = np.where(((scaled >= 0.) & (scaled < 1.)).all(axis=(1, 2)))[0]
        pts2 = scaled[valid[rs.randint(valid.shape[0])]]
    if translation:
        t_min, t_max = pts2.min(axis=0), (1 - pts2).min(axis=0)
        if allow_artifacts:
            t_min = t_min + translation_overflow
            t_max = t_max + translation_overflow
        pts2 = pts2 + np.array([rs.uniform(-t_min[0], t_max[0]), rs.uniform(-t_min[1], t_max[1])])[None]
    if rotation:
        angles = np.concatenate((np.linspace(-max_angle, max_angle, num=n_angles), [0.0]))
        center = pts2.mean(axis=0, keepdims=True)
        rot = np.stack([np.cos(angles), -np.sin(angles), np.sin(angles), np.cos(angles)], axis=1).reshape(-1, 2, 2)
        rotated = np.matmul((pts2 - center)[None], rot) + center
        if allow_artifacts:
            valid = np.arange(n_angles)
        else:
            valid = np.where(((rotated >= 0.) & (rotated < 1.)).all(axis=(1, 2)))[0]
        pts2 = rotated[valid[rs.randint(valid.shape[0])]]
    size = np.asarray(shape, dtype=np.float64)[::-1][None]
    return _perspective_transform(pts1 * size + shift, pts2 * size + shift)


def pixel_homography(rs, height, width, strength: float = 1.0, **params):
    """Homography x1 ~ M x0 taking image-0 pixels to image-1 pixels: the builder samples on [-1,1]^2, rescales to
    pixels and warps the image with the INVERSE (build_homography_dataset.py:141-164).
    strength < 1 shrinks the sampled homography towards the identity (H -> I + strength (H - I) on the normalised
    square): the seeded, untrained weights of this repo are not viewpoint-invariant, so recall against the known
    homography is only meaningful for mild views; strength = 1 is the yaml's recipe unchanged."""
    hn = sample_homography(rs, **{**HOMOGRAPHY_PARAMS, **params})
    if strength != 1.0:
        hn = hn / hn[2, 2]
        hn = np.eye(3) + float(strength) * (hn - np.eye(3))
    trans = np.array([[2. / width, 0., -1.], [0., 2. / height, -1.], [0., 0., 1.]])
    h_pix = np.linalg.inv(trans) @ hn @ trans
    m = np.linalg.inv(h_pix)
    return m / m[2, 2]


def warp_points(m, xy):
    """[...,2] points through the 3x3 homography m."""
    xy = np.asarray(xy, dtype=np.float64)
    q = xy @ m[:2, :2].T + m[:2, 2]
    w = xy @ m[2, :2] + m[2, 2]
    return q / w[..., None]


def homography_pair(seed: int, n_lines: int = 200, height: int = 480, width: int = 640, len_lo: float = 17.0,
                    len_hi: float = 167.0, margin: float = 10.0, common_frac: float = 0.85, strength: float = 1.0):
    """One homography-augmented pair of detector outputs.

    Returns (lines0 [n,6], lines1 [n,6], M, gt) where M takes image-0 pixels to image-1 pixels and gt [n] gives, for
    every row of lines0, the row of lines1 that is its warp (or -1).  About `common_frac` of the lines are seen in
    both images (their warped end points stay inside the margin box); the rest are lines of image 0 whose warp leaves
    image 1, respectively fresh lines that only image 1 has.  Image-1 rows are shuffled.  All coordinates are f32
    values, lineLength = f32(hypot) of the (warped) end points, octave 0."""
    rs = np.random.RandomState(seed)
    m = pixel_homography(rs, height, width, strength)
    n_common = int(round(n_lines * common_frac))

    def inside(x, y):
        return margin <= x <= width - margin and margin <= y <= height - margin

    def row(sx, sy, ex, ey):
        sx, sy, ex, ey = (float(np.float32(v)) for v in (sx, sy, ex, ey))
        return [sx, sy, ex, ey, float(np.float32(math.hypot(ex - sx, ey - sy))), 0.0]

    def draw():
        while True:
            sx = rs.uniform(margin, width - margin)
            sy = rs.uniform(margin, height - margin)
            ln = rs.uniform(len_lo, len_hi)
            th = rs.uniform(0.0, 2.0 * math.pi)
            ex, ey = sx + ln * math.cos(th), sy + ln * math.sin(th)
            if inside(ex, ey):
                return sx, sy, ex, ey

    common0, common1, only0 = [], [], []
    tries = 0
    while len(common0) < n_common or len(only0) < n_lines - n_common:
        tries += 1
        if tries > 200 * n_lines:     # a degenerate view (hardly any overlap): fill up with whatever fits
            break
        l0 = draw()
        w = warp_points(m, np.array([[l0[0], l0[1]], [l0[2], l0[3]]]))
        ok = inside(*w[0]) and inside(*w[1]) and math.hypot(*(w[1] - w[0])) >= len_lo
        if ok and len(common0) < n_common:
            common0.append(row(*l0)); common1.append(row(w[0, 0], w[0, 1], w[1, 0], w[1, 1]))
        elif not ok and len(only0) < n_lines - n_common:
            only0.append(row(*l0))
    while len(common0) + len(only0) < n_lines:
        only0.append(row(*draw()))
    only1 = [row(*draw()) for _ in range(n_lines - len(common1))]
    lines0 = np.asarray(common0 + only0, dtype=np.float64).reshape(-1, 6)
    l1 = np.asarray(common1 + only1, dtype=np.float64).reshape(-1, 6)
    perm0 = rs.permutation(n_lines)
    perm1 = rs.permutation(n_lines)
    lines0, lines1 = lines0[perm0], l1[perm1]
    inv1 = np.empty(n_lines, dtype=np.int64)
    inv1[perm1] = np.arange(n_lines)
    src = perm0                                            # lines0[i] was row perm0[i] of the unshuffled list
    gt = np.where(src < len(common0), inv1[np.minimum(src, n_lines - 1)], -1)
    return lines0, lines1, m, gt


def warp_dense_maps(dense_desc, dense_score, m, noise: float = 0.05, seed: int = 0):
    """Dense maps of image 1 = the maps of image 0 seen through m (x1 ~ m x0): bilinear resampling at the maps' own
    resolution (descriptor grid 1/8), a little seeded noise, descriptors re-normalised.  torch tensors in, same
    device/dtype out ([1,256,Hc,Wc], [1,H,W])."""
    import torch
    import torch.nn.functional as F
    dev = dense_desc.device
    H, W = int(dense_score.shape[-2]), int(dense_score.shape[-1])
    minv = torch.from_numpy(np.linalg.inv(m)).to(device=dev, dtype=torch.float32)

    def grid(h, w, step):
        ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                                torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
        x1 = xs * step + (step - 1) / 2.0                 # pixel centre of the cell in image 1
        y1 = ys * step + (step - 1) / 2.0
        den = minv[2, 0] * x1 + minv[2, 1] * y1 + minv[2, 2]
        x0 = (minv[0, 0] * x1 + minv[0, 1] * y1 + minv[0, 2]) / den
        y0 = (minv[1, 0] * x1 + minv[1, 1] * y1 + minv[1, 2]) / den
        gx = (x0 - (step - 1) / 2.0) / step               # cell coordinates in image 0's map
        gy = (y0 - (step - 1) / 2.0) / step
        return torch.stack([2 * (gx + 0.5) / w - 1, 2 * (gy + 0.5) / h - 1], dim=-1)[None]

    g = torch.Generator(device=dev).manual_seed(int(seed))
    dd = F.grid_sample(dense_desc, grid(H // 8, W // 8, 8), mode="bilinear", padding_mode="border", align_corners=False)
    dd = dd + noise * torch.randn(dd.shape, generator=g, device=dev) / math.sqrt(dd.shape[1])
    dd = F.normalize(dd, p=2, dim=1)
    ds = F.grid_sample(dense_score[None], grid(H, W, 1), mode="bilinear", padding_mode="border", align_corners=False)[0]
    return dd.contiguous(), ds.clamp(0, 1).contiguous()
