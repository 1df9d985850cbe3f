#!/usr/bin/env python3
"""Golden fixture of the TRAINING-TIME forward: the reference model called the way train.py:163-164
calls it -- one dict whose tensors carry a batch axis B > 1 and a FIXED number of sub-lines per image (the dataset
builder pads / truncates every image to max_sublines, dataloaders/utils/util_lines.py:670-766), with the training
configuration's 12 line-descriptive layers (train_manager.yaml:40; only the last one reaches the output).

    python tests/golden/make_golden_train.py      ->  tests/golden/train_batch.npz

Runs the REAL reference (same harness as make_golden.py).  Stored: the detector rows + map seeds of 3 images, the fixed
sub-line count, and the reference's line_desc [3,256,N].  Forward only: backward / the loss are out of scope."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (sets up the cv2 stub and puts /root/reference first on sys.path)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from workloads import synth  # noqa: E402

BATCH_KEYS = ["sublines", "pnt_sublines", "mask_sublines", "resp_sublines", "angle_sublines", "desc_sublines", "score_sublines"]


def main():
    hw, n_fix, n_layers = (480, 640), 40, 12
    m = G.ref_model(0, n_desc_layers=n_layers)
    outs, arrs = [], {"n_fix": n_fix, "n_desc_layers": n_layers, "hw": np.array(hw)}
    for b, (seed, n_lines) in enumerate(((301, 55), (302, 48), (303, 61))):
        rows = synth.synth_lines(seed, n_lines, *hw)
        dd, ds = synth.synth_dense_maps_np(seed, *hw)
        out = m.preprocess(synth.array_to_keylines(rows), (1, 1, *hw), {"dense_descriptor": torch.from_numpy(dd),
                                                                         "dense_score": torch.from_numpy(ds)}, None)
        assert out["sublines"].shape[1] >= n_fix
        outs.append(out)
        arrs[f"lines_{b}"] = rows
        arrs[f"map_seed_{b}"] = seed
    # fixed size like conv_fixed_size's truncation branch (util_lines.py:748-749), then the batch axis
    batch = {k: torch.cat([o[k][:, :n_fix] for o in outs], dim=0) for k in BATCH_KEYS}
    batch["klines"] = torch.cat([o["klines"][:, :n_fix] for o in outs], dim=0)
    res = m(batch)
    arrs["line_desc"] = res["line_desc"].numpy().copy()            # [3,256,n_fix]
    assert arrs["line_desc"].shape == (3, 256, n_fix)
    G.save("train_batch", **arrs)


if __name__ == "__main__":
    main()
