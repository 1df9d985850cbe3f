#!/usr/bin/env python3
"""Golden fixture with REAL-IMAGE statistics: the first pair of the reference's own demo list (assets/input_pairs.txt:1,
scannet_0a.png / scannet_0b.png) through the reference's SuperPoint and the line branch of its Matching.forward, the way
match_line_pairs.py:80-104 drives them (config of :56-72; --resize 640 480 is the images' own size).

Run in the build container only (needs /root/reference and PIL):

    python tests/golden/make_golden_asset_pair.py

Every other fixture feeds i.i.d. dense maps (normalize(randn) / rand).  Here the maps come from the reference's SuperPoint
(models/superpoint.py:146-205) run on real photographs: neighbouring descriptor cells are nearly identical, scores are a softmax
output with long runs of near-equal values -- the regime in which bilinear taps, score gathers and argmin margins behave differently.

What is NOT the reference's and is stated as such:
  * SuperPoint's weights (superpoint_v1.pth is not in the checkout): seeded ones (workloads.synth.superpoint_state_dict(0)) handed to the
    reference's constructor through torch.load, as make_golden_superpoint.py does.  Random filters on a real image still give spatially
    correlated maps -- the statistics this fixture is about -- but not SuperPoint's trained features.
  * the detector: cv2's LSD (models/line_detector.py:19-27) is not installed.  The line list is a deterministic stand-in made HERE from
    the image itself (edge_segments below: seeded candidate segments ranked by how well they sit on image edges) and is committed with
    the fixture; the product consumes KeyLine-like objects and never re-implements LSD.
  * the image decoder: PIL's "L" conversion instead of cv2.imread(..., IMREAD_GRAYSCALE) (models/utils.py:267); same ITU-R 601 weights,
    possibly other rounding.  The grey images are only an input to SuperPoint here; what is frozen are SuperPoint's OUTPUT maps.
  * LineTR's weights: the seeded, BatchNorm-calibrated set every other fixture uses (workloads.synth.calibrated_state_dict()).
`models.matching` cannot be imported (it constructs the cv2 detector and loads the missing blobs: SURVEY.md 8(c)); its statements
(models/matching.py:18-86) are executed one by one on the reference's own SuperPoint / LineTransformer / nn_matcher functions.
Only data is written: inputs (line lists, dense maps, key-point descriptors) and expected outputs.  No reference source, no image file.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")   # FIRST on the path: `models` must be the reference's package, not this repo's shim

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

torch.set_grad_enabled(False)

import models as _ref_models  # noqa: E402
assert _ref_models.__file__.startswith("/root/reference/"), _ref_models.__file__
import models.superpoint as ref_sp  # noqa: E402  (reference)
from models.line_transformer import LineTransformer  # noqa: E402  (reference)
from models.nn_matcher import nn_matcher, nn_matcher_distmat  # noqa: E402  (reference)
from models.line_process import get_dist_matrix  # noqa: E402  (reference)

from workloads import synth  # noqa: E402

ASSETS = "/root/reference/assets"
PAIR = ("scannet_0a.png", "scannet_0b.png")          # assets/input_pairs.txt:1
TOK_KEYS = ["klines", "length_klines", "angles", "sublines", "pnt_sublines", "mask_sublines", "resp_sublines",
            "angle_sublines", "score_sublines", "mat_klines2sublines", "line_desc"]


def superpoint(seed):
    sd = {k: torch.from_numpy(v) for k, v in synth.superpoint_state_dict(seed).items()}
    real_load = torch.load
    torch.load = lambda *a, **k: sd
    try:   # match_line_pairs.py:58-63
        return ref_sp.SuperPoint({"nms_radius": 4, "keypoint_threshold": 0.005, "max_keypoints": 1024, "nn_threshold": 0.7}).eval()
    finally:
        torch.load = real_load


def edge_segments(gray, n_lines, seed, n_cand=60000, samples=48):
    """Deterministic stand-in for the line detector: `n_cand` seeded candidate segments (random midpoint, length 18 .. 260 px, direction),
    scored by the mean image-gradient component ACROSS the segment sampled along it (a segment lying on an intensity edge scores high),
    then the best `n_lines` that do not duplicate an already chosen one (midpoints > 6 px apart or directions > 10 degrees apart).
    Rows as the detector's KeyLines would give them: (sx, sy, ex, ey, lineLength, octave), coordinates rounded to float32."""
    h, w = gray.shape
    gy, gx = np.gradient(gray.astype(np.float64))
    rs = np.random.RandomState(seed)
    mx, my = rs.uniform(12, w - 12, n_cand), rs.uniform(12, h - 12, n_cand)
    ln = rs.uniform(18.0, 260.0, n_cand)
    th = rs.uniform(0.0, np.pi, n_cand)
    dx, dy = np.cos(th), np.sin(th)
    sx, sy, ex, ey = mx - 0.5 * ln * dx, my - 0.5 * ln * dy, mx + 0.5 * ln * dx, my + 0.5 * ln * dy
    ok = (np.minimum(sx, ex) >= 2) & (np.maximum(sx, ex) <= w - 3) & (np.minimum(sy, ey) >= 2) & (np.maximum(sy, ey) <= h - 3)
    t = np.linspace(0.0, 1.0, samples)[None, :]
    px = np.clip(np.rint(sx[:, None] + (ex - sx)[:, None] * t).astype(int), 0, w - 1)
    py = np.clip(np.rint(sy[:, None] + (ey - sy)[:, None] * t).astype(int), 0, h - 1)
    across = gx[py, px] * (-dy)[:, None] + gy[py, px] * dx[:, None]           # gradient component normal to the segment
    score = np.abs(across.mean(1)) - 0.5 * across.std(1)                       # consistently signed edge response along the whole segment
    score[~ok] = -np.inf
    rows = []
    for i in np.argsort(-score, kind="stable"):
        if not np.isfinite(score[i]) or len(rows) >= n_lines:
            break
        if any(np.hypot(mx[i] - r[6], my[i] - r[7]) < 6.0 and abs(((th[i] - r[8] + np.pi / 2) % np.pi) - np.pi / 2) < np.deg2rad(10) for r in rows):
            continue
        c = [float(np.float32(v)) for v in (sx[i], sy[i], ex[i], ey[i])]
        rows.append([c[0], c[1], c[2], c[3], float(np.float32(np.hypot(c[2] - c[0], c[3] - c[1]))), 0.0, mx[i], my[i], th[i]])
    return np.asarray(rows)[:, :6]


def main():
    sp = superpoint(0)
    lt = LineTransformer({"max_keylines": -1, "min_length": 16, "token_distance": 8, "nn_threshold": 0.8, "mode": "train"}).eval()   # match_line_pairs.py:67-72
    lt.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
    arrs, outs, preds = {}, {}, {}
    for s, name in zip("01", PAIR):
        img = Image.open(os.path.join(ASSETS, name)).convert("L")
        assert img.size == (640, 480)
        gray = np.asarray(img, dtype=np.float32)
        image = torch.from_numpy(gray / 255.0)[None, None]                      # frame2tensor (models/utils.py)
        pred = sp({"image": image})                                              # models/matching.py:20-26
        lines = edge_segments(gray, 230, seed=100 + int(s))
        image_shape = image.shape
        lt.config["min_length"] = max(16, max(image_shape) / 40)                 # :31-32 (auto_min_length is on by default, match_line_pairs.py:40)
        lt.config["token_distance"] = max(8, max(image_shape) / 80)
        valid_mask = torch.ones_like(image)                                      # :37-38
        out = lt(lt.preprocess(synth.array_to_keylines(lines), image_shape, pred, valid_mask))
        outs[s], preds[s] = out, pred
        arrs["lines" + s] = lines
        arrs["dense_descriptor" + s] = pred["dense_descriptor"].numpy()
        arrs["dense_score" + s] = pred["dense_score"].numpy()
        arrs["descriptors" + s] = pred["descriptors"][0].numpy()                # [256, n_kp]
        arrs["keypoints" + s] = pred["keypoints"][0].numpy()
        for k in TOK_KEYS:
            arrs[k + s] = out[k].numpy().copy()
        desc = out["desc_sublines"].numpy()[0]
        arrs["desc_checksum" + s] = desc.astype(np.float64).sum(-1)
        arrs["desc_abs_checksum" + s] = np.abs(desc.astype(np.float64)).sum(-1)
        rs = np.random.RandomState(5 + int(s))
        idx = np.stack([rs.randint(0, desc.shape[0], 96), rs.randint(0, desc.shape[1], 96)], 1)
        arrs["desc_sample_idx" + s], arrs["desc_sample" + s] = idx, desc[idx[:, 0], idx[:, 1]]
    # ---- models/matching.py:66-84
    mp, dp = nn_matcher(arrs["descriptors0"], arrs["descriptors1"], sp.config["nn_threshold"], is_mutual_NN=True)
    D = get_dist_matrix(outs["0"]["line_desc"].cpu().numpy(), outs["1"]["line_desc"].cpu().numpy())[0]
    Dk = lt.subline2keyline(D, outs["0"]["mat_klines2sublines"][0], outs["1"]["mat_klines2sublines"][0])
    M = nn_matcher_distmat(Dk, lt.config["nn_threshold"], is_mutual_NN=True)
    # the point matcher's [n0, n1] float64 / float32 matrices are 12 MB: kept as the matched index per row + the row minima of the distances
    mp0 = mp[0]
    arrs.update(matches_l=M, matching_scores_l=Dk, dist_sublines=D,
                matches_p_index=np.where(mp0.sum(1) > 0, mp0.argmax(1), -1).astype(np.int32), matches_p_count=np.asarray(int(mp0.sum())),
                matching_scores_p_rowmin=dp[0].min(1), matching_scores_p_colmin=dp[0].min(0), hw=np.asarray((480, 640)))
    two = np.sort(Dk[0], axis=1)[:, :2]
    twoc = np.sort(Dk[0], axis=0)[:2, :]
    print(f"K = {[outs[s]['klines'].shape[1] for s in '01']}, N = {[outs[s]['sublines'].shape[1] for s in '01']}, line matches = {int(M.sum())}, "
          f"point matches = {int(mp0.sum())} of {mp0.shape}")
    print(f"Dk range {Dk.min():.4g} .. {Dk.max():.4g}; smallest argmin margin rows {np.min(two[:, 1] - two[:, 0]):.3g}, cols {np.min(twoc[1] - twoc[0]):.3g}; "
          f"margins < 1e-5: {int((two[:, 1] - two[:, 0] < 1e-5).sum() + (twoc[1] - twoc[0] < 1e-5).sum())}")
    ds = arrs["dense_score0"]
    dd = arrs["dense_descriptor0"][0]
    cos_nb = (dd[:, :, 1:] * dd[:, :, :-1]).sum(0).mean()
    print(f"dense_score0: min {ds.min():.3g} max {ds.max():.3g} exact zeros {int((ds == 0).sum())}; mean cosine of horizontally neighbouring descriptor cells {cos_nb:.4f}")
    path = os.path.join(HERE, "asset_pair.npz")
    np.savez_compressed(path, **arrs)
    print(f"asset_pair.npz  {os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
