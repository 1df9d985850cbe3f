"""Shared helpers for the test-suite (oracle side).  Test infrastructure only."""
import os

import numpy as np
import torch

from workloads import synth
from oracle import linetr_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BASE_CFG = dict(min_length=16, token_distance=8, max_tokens=21, remove_borders=8, max_keylines=-1,
                nn_threshold=0.8)
TOK_KEYS = ["klines", "length_klines", "angles", "sublines", "pnt_sublines", "mask_sublines",
            "resp_sublines", "angle_sublines", "score_sublines", "mat_klines2sublines"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_cfg(g, **extra):
    cfg = dict(BASE_CFG)
    for k in g.files:
        if k.startswith("cfg_"):
            v = g[k]
            cfg[k[4:]] = v.item()
    cfg.update(extra)
    return cfg


def weights_for(g):
    seed = int(g["weight_seed"]) if "weight_seed" in g.files else 0
    nl = int(g["n_desc_layers"]) if "n_desc_layers" in g.files else 1
    return synth.calibrated_state_dict(nl) if seed == 0 else synth.make_state_dict(seed, nl)


def tiny_maps(g):
    hw = tuple(int(v) for v in g["hw"])
    dd, ds = synth.synth_dense_maps_np(int(g["map_seed"]), *hw)
    return torch.from_numpy(dd), torch.from_numpy(ds), hw


def oracle_image(sd_t, rows, dd, ds, hw, cfg, valid_mask=None, align_corners=None, image_shape=None):
    out = O.preprocess(synth.array_to_keylines(rows), (1, 1, hw[0], hw[1]), dd, ds, cfg, valid_mask, align_corners)
    if len(out["klines"]) == 0:
        return out
    return O.forward(sd_t, out, image_shape or hw)


TRAIN_KEYS = ("sublines", "pnt_sublines", "mask_sublines", "resp_sublines", "angle_sublines", "desc_sublines", "score_sublines")


def train_mode_batches(g, preprocess_fn, tokenizer_fn, to_dev=lambda t: t):
    """The two [3, 250, ...] batches of tests/golden/train_mode.npz rebuilt from the fixture's inputs: every image is tokenised by
    `preprocess_fn(rows, pred)`, padded with the fixture's pseudo lines -- tokenised by `tokenizer_fn(lines_dict, pred)` with the
    (640, 480) image_shape conv_fixed_size passes -- or truncated to 250 sub-lines (dataloaders/utils/util_lines.py:670-766: a
    concatenation / a slice of the seven per-sub-line tensors; no arithmetic).  `pred` = {'dense_descriptor', 'dense_score'}."""
    hw = tuple(int(v) for v in g["hw"])
    n_max = int(g["max_sublines"])
    batches = []
    for c in range(2):
        outs = []
        for b in range(3):
            dd, ds = synth.synth_dense_maps_np(int(g[f"map_seed_{c}_{b}"]), *hw)
            pred = {"dense_descriptor": to_dev(torch.from_numpy(dd)), "dense_score": to_dev(torch.from_numpy(ds))}
            real = preprocess_fn(g[f"lines_{c}_{b}"], pred)
            one = {k: real[k][:, :n_max] for k in TRAIN_KEYS}
            if int(g[f"n_pseudo_{c}_{b}"]):
                lines = {k: g[f"pseudo_{k}_{c}_{b}"].copy() for k in ("klines", "length_klines", "angles")}
                pseudo = tokenizer_fn(lines, pred)
                one = {k: torch.cat([one[k], pseudo[k]], dim=1) for k in TRAIN_KEYS}
            assert one["sublines"].shape[1] == n_max
            outs.append(one)
        batch = {k: torch.cat([o[k] for o in outs], dim=0) for k in TRAIN_KEYS}
        batch["klines"] = batch["sublines"]           # only its length is read (models/line_transformer.py:226)
        batches.append(batch)
    return batches
