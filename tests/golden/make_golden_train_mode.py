#!/usr/bin/env python3
"""Golden fixture of the TRAIN-MODE forward: the reference model in .train() (train.py:127), called twice
in a row the way train.py:163-164 does (pred0 = model(data0); pred1 = model(data1)) on batches of B = 3 fixed-size samples of 250
sub-lines, padded with tokenised pseudo lines / truncated by the reference's own conv_fixed_size
(dataloaders/utils/util_lines.py:670-766, with the dataset builder's conf: resize (640, 480), max_sublines 250 --
dataloaders/confs/homography.yaml:7,57), with the training configuration's 12 line-descriptive layers (train_manager.yaml:40).

    python tests/golden/make_golden_train_mode.py      ->  tests/golden/train_mode.npz

BatchNorm1d runs on batch statistics and updates its running statistics; the probability of every nn.Dropout INSTANCE is set to 0
(an attribute of the module objects; no reference source is touched) because dropout masks are RNG- and device-specific and
cannot be part of a fixture.  Runs the REAL reference (same harness as make_golden.py).  Stored: detector rows + map seeds, the
pseudo-line dicts exactly as conv_fixed_size hands them to line_tokenizer (float64, before its in-place clip), the reference's
line_desc of both calls and every BatchNorm's running statistics after each call.  Forward only (no loss / backward)."""
import copy
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (sets up the cv2 stub and puts /root/reference first on sys.path)
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, "/root/reference/dataloaders")
from utils import util_lines  # noqa: E402  (reference; numpy / torch / math only)
assert util_lines.__file__.startswith("/root/reference/"), util_lines.__file__
from models.line_process import line_tokenizer  # noqa: E402  (reference)

from workloads import synth  # noqa: E402

KEYS = ["sublines", "pnt_sublines", "mask_sublines", "resp_sublines", "angle_sublines", "desc_sublines", "score_sublines"]
CONF = {"data": {"resize": (640, 480)},
        "feature": {"linetr": {"token_distance": 8, "max_tokens": 21, "max_sublines": 250, "min_length": 16}}}


def bn_state(m):
    out = {}
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            out[name + ".running_mean"] = mod.running_mean.numpy().copy()
            out[name + ".running_var"] = mod.running_var.numpy().copy()
            out[name + ".num_batches_tracked"] = mod.num_batches_tracked.numpy().copy()
    return out


def main():
    hw, n_layers = (480, 640), 12
    m = G.ref_model(0, n_desc_layers=n_layers)
    m.train()                                                    # train.py:127
    n_drop = 0
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
            n_drop += 1
    assert n_drop == 3 * n_layers, n_drop                        # ScaledDotProduct + MultiHeadAttention + FeedForward per layer
    arrs = {"n_desc_layers": n_layers, "hw": np.array(hw), "max_sublines": 250}
    batches = []
    # (seed, detector lines, longest line): 139 / 197 sub-lines -> padded with pseudo lines; 240 lines of up to 330 px -> more than
    # 250 sub-lines from fewer than 250 key-lines (lines above 168 px split in two) -> the truncation branch (util_lines.py:736-757;
    # that branch needs num_klns <= max_sublines, which the dataset builder's own preprocess guarantees)
    plan = [[(401, 140, 167.0), (402, 198, 167.0), (403, 240, 330.0)], [(404, 120, 167.0), (405, 230, 330.0), (406, 175, 167.0)]]
    for c, images in enumerate(plan):
        outs = []
        for b, (seed, n_lines, len_hi) in enumerate(images):
            rows = synth.synth_lines(seed, n_lines, *hw, 17.0, len_hi)
            dd, ds = synth.synth_dense_maps_np(seed, *hw)
            pred = {"dense_descriptor": torch.from_numpy(dd), "dense_score": torch.from_numpy(ds)}
            kl = m.preprocess(synth.array_to_keylines(rows), (1, 1, *hw), pred, None)
            captured, tokenised = [], []

            def tokenizer(lines, td, T, pred_sp, image_shape):
                captured.append({k: np.array(v, dtype=np.float64, copy=True) for k, v in lines.items()})
                assert tuple(image_shape) == (640, 480)          # conv_fixed_size passes conf['data']['resize'] (width, height)
                out = line_tokenizer(lines, td, T, pred_sp, image_shape)
                tokenised.append({k: out[k].numpy().copy() for k in G.TENSOR_KEYS})
                return out
            np.random.seed(1000 + seed)                          # make_pseudo_lines draws from the global NumPy generator
            fixed = util_lines.conv_fixed_size(kl, copy.deepcopy(CONF), func_token=tokenizer, pred_sp=pred)
            assert fixed["sublines"].shape[1] == 250 and len(captured) <= 1
            print(f"image {c}.{b}: {int(fixed['num_klns'])} key-lines, {int(fixed['num_slns'])} sub-lines, {len(captured[0]['klines']) if captured else 0} pseudo lines")
            outs.append(fixed)
            arrs[f"lines_{c}_{b}"] = rows
            arrs[f"map_seed_{c}_{b}"] = seed
            arrs[f"n_pseudo_{c}_{b}"] = len(captured[0]["klines"]) if captured else 0
            if captured:
                for k, v in captured[0].items():
                    arrs[f"pseudo_{k}_{c}_{b}"] = v
                if c == 0 and b == 0:      # the tokeniser's own outputs for one pseudo set: pins the (640, 480) end-point clip
                    for k, v in tokenised[0].items():
                        arrs[f"pseudo_tok_{k}"] = v
        batch = {k: torch.cat([o[k] for o in outs], dim=0) for k in KEYS}
        batch["klines"] = torch.cat([o["klines"] for o in outs], dim=0)
        batches.append(batch)
    for c, batch in enumerate(batches):                          # train.py:163-164: two calls, statistics move twice
        res = m(batch)
        ld = res["line_desc"].detach().numpy().copy()
        assert ld.shape == (3, 256, 250)
        arrs[f"line_desc_{c}"] = ld if c == 0 else ld[:, :, ::5].copy()       # second call: every fifth sub-line (fixture size)
        for k, v in bn_state(m).items():
            arrs[f"bn{c}.{k}"] = v
    G.save("train_mode", **arrs)


if __name__ == "__main__":
    main()
