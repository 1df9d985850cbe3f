"""Freezes a small cfg4 job for the multi-rank tests: 8 homography pairs (mild views, 40 lines per image), the CPU
oracle's line descriptors / key-line maps of all 16 images, and the oracle's global-matching answers.

    python tests/golden/make_cfg4_fixture.py        ->  tests/golden/cfg4_job.npz

Uses this repo's own synth + oracle only (the reference is not needed: the oracle is pinned to it by the other
fixtures).  The multi-process gloo test replays these arrays as the "engine output" of 4 ranks; the GPU test compares
the HIP path with them."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from workloads import synth  # noqa: E402
from oracle import linetr_oracle as O  # noqa: E402

P, S, N_LINES, H, W, STRENGTH = 8, 3, 40, 480, 640, 0.05
CFG = dict(min_length=16, token_distance=8, max_tokens=21, remove_borders=8, max_keylines=-1)


def main():
    torch.set_grad_enabled(False)
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    out = {"P": P, "S": S, "n_lines": N_LINES, "hw": np.array([H, W]), "strength": STRENGTH, "seed_base": 40000}
    imgs = []
    for p in range(P):
        l0, l1, m, gt = synth.homography_pair(40000 + p, N_LINES, H, W, strength=STRENGTH)
        dd0, ds0 = synth.synth_dense_maps_np(40000 + p, H, W)
        dd0, ds0 = torch.from_numpy(dd0), torch.from_numpy(ds0)
        dd1, ds1 = synth.warp_dense_maps(dd0, ds0, m, seed=p)
        for side, (rows, dd, ds) in enumerate(((l0, dd0, ds0), (l1, dd1, ds1))):
            o = O.forward(sd, O.preprocess(synth.array_to_keylines(rows), (1, 1, H, W), dd, ds, CFG), (H, W))
            imgs.append(o)
            i = 2 * p + side
            out[f"lines_{i}"] = rows
            out[f"desc_{i}"] = o["line_desc"][0].numpy().T.copy()                  # [n,256]
            out[f"s2l_{i}"] = o["mat_klines2sublines"][0].argmax(0).numpy().astype(np.int32)
            out[f"k_{i}"] = np.int32(o["klines"].shape[1])
            out[f"klines_{i}"] = o["klines"][0].numpy()
        out[f"homography_{p}"] = m
    for p in range(P):
        for s in range(S):
            c = (p + s) % P
            a, b = imgs[2 * p], imgs[2 * c + 1]
            M, Dk = O.match_lines(a["line_desc"], b["line_desc"], a["mat_klines2sublines"][0], b["mat_klines2sublines"][0], 0.8)
            m01 = np.where(M[0].sum(1) > 0, M[0].argmax(1), -1).astype(np.int32)
            out[f"match_{p}_{s}"] = m01
            out[f"dk_{p}_{s}"] = Dk[0].astype(np.float32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg4_job.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
