#!/usr/bin/env python3
"""Calibrate the BatchNorm running statistics of the seeded synthetic weights.

Random Conv/BN weights make every line descriptor collapse onto one common vector (pairwise
distances ~1e-4, argmin margins ~1e-6), which would make "matches bit-exact by index" a coin flip.
A trained checkpoint does not behave like that because its BN running stats match its activations.
This script reproduces that property: it runs the CPU oracle once on a calibration image
(synthetic lines/maps, seed 13) and records, for every BatchNorm, the per-channel mean/variance of
its input.  The result (36 KB of data) is committed as workloads/data/bn_calib_seed0.npz and
overlaid on ``synth.make_state_dict(0)`` by ``synth.calibrated_state_dict()``.

    python tests/golden/make_calib.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.set_grad_enabled(False)
from workloads import synth  # noqa: E402
from oracle import linetr_oracle as O  # noqa: E402


def main(seed=0, image_seed=13):
    sd = synth.to_torch_state_dict(synth.make_state_dict(seed))
    rows = synth.synth_lines(image_seed)
    dd, ds = synth.synth_dense_maps(image_seed, 480, 640)
    cfg = dict(min_length=16, token_distance=8, max_tokens=21, remove_borders=8, max_keylines=-1)
    data = O.preprocess(synth.array_to_keylines(rows), (1, 1, 480, 640), dd, ds, cfg)
    stats = {}

    def mlp_recording(sd_, prefix, x):
        idx = 0
        while f"{prefix}.{idx}.weight" in sd_:
            x = F.linear(x, sd_[f"{prefix}.{idx}.weight"][:, :, 0], sd_[f"{prefix}.{idx}.bias"])
            idx += 1
            if f"{prefix}.{idx}.running_mean" in sd_:
                mu, var = x.mean(0), x.var(0, unbiased=False) + 1e-3
                sd_[f"{prefix}.{idx}.running_mean"], sd_[f"{prefix}.{idx}.running_var"] = mu, var
                stats[f"{prefix}.{idx}.running_mean"] = mu.numpy().copy()
                stats[f"{prefix}.{idx}.running_var"] = var.numpy().copy()
                x = F.relu(F.batch_norm(x, mu, var, sd_[f"{prefix}.{idx}.weight"], sd_[f"{prefix}.{idx}.bias"],
                                        False, 0.0, 1e-5))
                idx += 2
        return x

    keep = O._mlp
    O._mlp = mlp_recording
    try:
        O.forward(sd, data, (480, 640))
    finally:
        O._mlp = keep
    path = os.path.join(ROOT, "workloads", "data", f"bn_calib_seed{seed}.npz")
    np.savez_compressed(path, **stats)
    print(path, len(stats), "tensors", os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
