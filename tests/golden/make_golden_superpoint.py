#!/usr/bin/env python3
"""Golden fixture for the fused SuperPoint-head producer, from the REAL reference.

    python tests/golden/make_golden_superpoint.py        (build container only: needs /root/reference)

The reference's SuperPoint constructor loads `weights/superpoint_v1.pth`, which is not in the checkout; weights are
data, so `torch.load` is pointed at the seeded state_dict of `workloads.synth.superpoint_state_dict` for the
duration of the constructor.  Forward hooks capture the raw head outputs (convPb / convDb); the fixture holds those
inputs plus everything `SuperPoint.forward` returns.  Only data is written.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))   # linetr_amd (seeded inputs / weights)
sys.path.insert(0, "/root/reference")   # FIRST on the path: `models` must be the reference's package, not this repo's shim

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_grad_enabled(False)
import models.superpoint as ref_sp  # noqa: E402  (reference)
assert ref_sp.__file__.startswith("/root/reference/"), ref_sp.__file__

from workloads import synth  # noqa: E402


def build(seed, **cfg):
    sd = {k: torch.from_numpy(v) for k, v in synth.superpoint_state_dict(seed).items()}
    real_load = torch.load
    torch.load = lambda *a, **k: sd
    try:
        m = ref_sp.SuperPoint(cfg).eval()
    finally:
        torch.load = real_load
    return m


def main():
    m = build(0, max_keypoints=-1, keypoint_threshold=0.005, nms_radius=4, remove_borders=4)
    cap = {}
    m.convPb.register_forward_hook(lambda mod, i, o: cap.__setitem__("score_logits", o.detach().clone()))
    m.convDb.register_forward_hook(lambda mod, i, o: cap.__setitem__("desc_raw", o.detach().clone()))
    g = torch.Generator().manual_seed(7)
    img = torch.rand(2, 1, 64, 96, generator=g)
    out = m({"image": img})
    fx = {"image": img.numpy(), "score_logits": cap["score_logits"].numpy(), "desc_raw": cap["desc_raw"].numpy(),
          "dense_score": out["dense_score"].numpy(), "dense_descriptor": out["dense_descriptor"].numpy(),
          "weights_seed": np.int64(0)}
    for b in range(2):
        fx[f"keypoints{b}"] = out["keypoints"][b].numpy()
        fx[f"scores{b}"] = out["scores"][b].numpy()
        fx[f"descriptors{b}"] = out["descriptors"][b].numpy()
    path = os.path.join(HERE, "superpoint_heads.npz")
    np.savez_compressed(path, **fx)
    print(path, {k: getattr(v, "shape", None) for k, v in fx.items()})


if __name__ == "__main__":
    main()
