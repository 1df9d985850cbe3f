#!/usr/bin/env python3
"""Golden fixture for a pair of DIFFERENT image sizes through the line branch of the reference's Matching.forward
(models/matching.py:28-60, :77-84) with auto_min_length on: min_length / token_distance are recomputed PER IMAGE from that image's
own shape (:29-32 for image0, :45-48 for image1), while normalize_keylines keeps the constructor's image_shape
(models/line_transformer.py:206, :238).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_mixed.py

`models.matching` itself cannot be imported here (it constructs the cv2 LSD detector and loads the SuperPoint blob: SURVEY.md
section 8(c)); the statements of its line branch are executed one by one on the reference's own LineTransformer / get_dist_matrix /
nn_matcher_distmat, as make_golden.py does for the matching tail.  Only data is written.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_grad_enabled(False)

import models as _ref_models  # noqa: E402
assert _ref_models.__file__.startswith("/root/reference/"), _ref_models.__file__
from models.line_transformer import LineTransformer  # noqa: E402  (reference)
from models.nn_matcher import nn_matcher_distmat  # noqa: E402  (reference)
from models.line_process import get_dist_matrix  # noqa: E402  (reference)

from workloads import synth  # noqa: E402

KEYS = ["klines", "length_klines", "angles", "sublines", "pnt_sublines", "mask_sublines", "resp_sublines",
        "angle_sublines", "score_sublines", "mat_klines2sublines", "line_desc"]


def main():
    m = LineTransformer({"mode": "train", "nn_threshold": 0.8, "max_keylines": -1, "min_length": 16, "token_distance": 8}).eval()
    m.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
    sizes = {"0": (480, 640), "1": (960, 1280)}
    lines = {"0": synth.synth_lines(71, 70, 480, 640, 12.0, 300.0),       # some below min_length 16, some two sub-lines (> 168 px)
             "1": synth.synth_lines(72, 110, 960, 1280, 20.0, 700.0)}     # some below min_length 32, some three sub-lines (> 672 px)
    arrs, outs = {}, {}
    for s in ("0", "1"):
        h, w = sizes[s]
        dd, ds = synth.synth_dense_maps(70 + int(s), h, w)
        image_shape = (1, 1, h, w)
        # models/matching.py:29-32 / :45-48
        m.config["min_length"] = max(16, max(image_shape) / 40)
        m.config["token_distance"] = max(8, max(image_shape) / 80)
        valid_mask = torch.ones(image_shape)                               # :37-38: a tensor, ignored by remove_borders
        out = m.preprocess(synth.array_to_keylines(lines[s]), image_shape, {"dense_descriptor": dd, "dense_score": ds}, valid_mask)
        out = m(out)
        outs[s] = out
        arrs["lines" + s] = lines[s]
        arrs["map_seed" + s] = np.asarray(70 + int(s))
        arrs["hw" + s] = np.asarray(sizes[s])
        arrs["min_length" + s] = np.asarray(m.config["min_length"], dtype=np.float64)
        arrs["token_distance" + s] = np.asarray(m.config["token_distance"], dtype=np.float64)
        for k in KEYS:
            arrs[k + s] = out[k].numpy().copy()
    # :77-84
    D = get_dist_matrix(outs["0"]["line_desc"].cpu().numpy(), outs["1"]["line_desc"].cpu().numpy())[0]
    Dk = m.subline2keyline(D, outs["0"]["mat_klines2sublines"][0], outs["1"]["mat_klines2sublines"][0])
    M = nn_matcher_distmat(Dk, m.config["nn_threshold"], is_mutual_NN=True)
    arrs.update(matches_l=M, matching_scores_l=Dk, final_min_length=np.asarray(m.config["min_length"], dtype=np.float64),
                final_token_distance=np.asarray(m.config["token_distance"], dtype=np.float64))
    path = os.path.join(HERE, "mixed_size_pair.npz")
    np.savez_compressed(path, **arrs)
    print(f"mixed_size_pair.npz  {os.path.getsize(path) / 1024:.0f} KB; K = {[outs[s]['klines'].shape[1] for s in '01']}, "
          f"N = {[outs[s]['sublines'].shape[1] for s in '01']}, matches = {int(M.sum())}")


if __name__ == "__main__":
    main()
