#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REAL reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports the reference's hot-path modules exactly as SURVEY.md Appendix D prescribes (cv2 stub,
mode='train' so no weight file is needed), loads the seeded weights of ``workloads.synth`` into
the reference model, pushes synthetic KeyLines + dense maps through ``preprocess`` -> ``forward`` ->
the line-matching tail of ``Matching.forward`` and freezes inputs + outputs as small .npz files.
Only data is written; no reference source or bytecode is copied.  The fixtures are what pins
``oracle/linetr_oracle.py`` (tests/test_oracle_golden.py) and, on the GPU box where the reference
does not exist, the HIP path (tests/test_gpu_parity.py).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # imported at line_process.py:2, never called
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))   # linetr_amd (seeded inputs / weights)
sys.path.insert(0, "/root/reference")   # FIRST on the path: `models` must be the reference's package, not this repo's shim

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_grad_enabled(False)

import models as _ref_models  # noqa: E402
assert _ref_models.__file__.startswith("/root/reference/"), _ref_models.__file__
from models.line_transformer import LineTransformer  # noqa: E402  (reference)
from models.nn_matcher import nn_matcher_distmat, nn_matcher  # noqa: E402  (reference)
from models.line_process import get_dist_matrix  # noqa: E402  (reference)

from workloads import synth  # noqa: E402

TENSOR_KEYS = ["klines", "length_klines", "angles", "sublines", "pnt_sublines", "mask_sublines",
               "resp_sublines", "angle_sublines", "score_sublines", "mat_klines2sublines"]


def ref_model(seed, n_desc_layers=1, image_shape=(480, 640), **cfg):
    conf = {"mode": "train", "nn_threshold": 0.8, "image_shape": list(image_shape),
            "n_line_descriptive_layers": n_desc_layers, **cfg}
    m = LineTransformer(conf).eval()
    # seed 0 = the BN-calibrated weights (well-separated descriptors); other seeds are raw
    sdn = synth.calibrated_state_dict(n_desc_layers) if seed == 0 else synth.make_state_dict(seed, n_desc_layers)
    m.load_state_dict(synth.to_torch_state_dict(sdn), strict=True)
    return m


def run_image(m, rows, dd, ds, hw, valid_mask=None, torch_version=None):
    """reference preprocess + forward on one image; returns dict of numpy outputs."""
    kl = synth.array_to_keylines(rows)
    shape4 = (1, 1, hw[0], hw[1])
    saved = torch.__version__
    if torch_version is not None:      # exercises the align_corners switch at line_process.py:93
        torch.__version__ = torch_version
    try:
        out = m.preprocess(kl, shape4, {"dense_descriptor": dd, "dense_score": ds}, valid_mask)
    finally:
        torch.__version__ = saved
    if len(out["klines"]) == 0:
        return {"empty": np.asarray(1)}, out
    out = m(out)
    res = {k: out[k].numpy().copy() for k in TENSOR_KEYS}
    res["line_desc"] = out["line_desc"].numpy().copy()
    desc = out["desc_sublines"].numpy()[0]                      # [N,T,256]
    res["desc_checksum"] = desc.astype(np.float64).sum(-1)       # [N,T]
    res["desc_abs_checksum"] = np.abs(desc.astype(np.float64)).sum(-1)
    return res, out


def match_tail(m, out0, out1, thr):
    """models/matching.py:77-84 reproduced with the reference's own functions."""
    d0 = out0["line_desc"].cpu().numpy()
    d1 = out1["line_desc"].cpu().numpy()
    D = get_dist_matrix(d0, d1)[0]
    Dk = m.subline2keyline(D, out0["mat_klines2sublines"][0], out1["mat_klines2sublines"][0])
    M = nn_matcher_distmat(Dk, thr, True)
    return D, Dk, M


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KB")


def pack(prefix, res):
    return {f"{prefix}_{k}": v for k, v in res.items()}


def special_lines(hw):
    """Hand-made lines exercising Appendix A quirks: multi-subline, vertical, octave 1,
    reversed endpoints, border rejects, short rejects, equal-x swap, near-border clip."""
    H, W = hw
    L = synth.KeyLine
    lines = [
        L(20.5, 30.25, 20.5 + 400.0, 30.25 + 120.0),                 # 418 px -> 53 tokens -> 3 sublines
        L(300.0, 400.0, 300.0, 100.0),                               # vertical, going up (equal x -> swapped)
        L(310.0, 100.0, 310.0, 380.0),                               # vertical, going down
        L(500.0, 50.0, 100.0, 300.0),                                # reversed endpoints (sp.x > ep.x)
        L(50.0, 200.0, 250.0, 210.0, length=np.float32(100.12492), octave=1),  # octave 1: length*2
        L(3.0, 100.0, 80.0, 120.0),                                  # start inside border -> removed
        L(100.0, 100.0, 108.0, 106.0),                               # 10 px -> too short
        L(W - 8.0005, 20.0, W - 60.0, 90.0),                         # inside [b, W-b) but clipped by eps
        L(120.0, H - 8.0004, 200.0, H - 50.0),
        L(150.0, 150.0, 150.0 + 168.0, 150.0),                       # exactly 21 tokens, horizontal
        L(150.0, 170.0, 150.0 + 168.5, 170.0),                       # 22 tokens -> 2 sublines (1 token in 2nd)
        L(60.0, 60.0, 60.0 + 17.0, 60.0 + 1.0),                      # just above min_length
    ]
    return np.asarray([l.as_row() for l in lines])


def main():
    # ---------------------------------------------------------------- cfg2: one 640x480 pair
    m = ref_model(0)
    hw = (480, 640)
    outs = []
    arrs = {}
    for tag, seed in (("a", 11), ("b", 12)):
        rows = synth.synth_lines(seed, 200, *hw)
        dd, ds = synth.synth_dense_maps(seed, *hw)
        res, out = run_image(m, rows, dd, ds, hw)
        outs.append(out)
        # keep a 64-row sample of the 4.3 MB desc_sublines tensor + full-tensor checksums
        rs = np.random.RandomState(seed)
        desc = out["desc_sublines"].numpy()[0]
        ii = rs.randint(0, desc.shape[0], 64)
        jj = rs.randint(0, desc.shape[1], 64)
        res["desc_sample_idx"] = np.stack([ii, jj], 1)
        res["desc_sample"] = desc[ii, jj]
        arrs.update(pack(tag, res))
        arrs[f"{tag}_lines"] = rows
        arrs[f"{tag}_seed"] = np.asarray(seed)
    D, Dk, M = match_tail(m, outs[0], outs[1], 0.8)
    arrs.update(pair_D=D, pair_Dk=Dk, pair_M=M, weight_seed=np.asarray(0), hw=np.asarray(hw))
    save("cfg2_pair", **arrs)

    # ---------------------------------------------------------------- cfg2 matching-meaningful pair
    rows0 = synth.synth_lines(21, 200, *hw)
    rows1, perm = synth.jitter_pair(rows0, 22, 0.3)
    dd, ds = synth.synth_dense_maps(21, *hw)
    r0, o0 = run_image(m, rows0, dd, ds, hw)
    r1, o1 = run_image(m, rows1, dd, ds, hw)
    D, Dk, M = match_tail(m, o0, o1, 0.8)
    save("cfg2_jitter_pair", a_lines=rows0, b_lines=rows1, perm=perm, a_line_desc=r0["line_desc"],
         b_line_desc=r1["line_desc"], a_klines=r0["klines"], b_klines=r1["klines"],
         pair_Dk=Dk, pair_M=M, weight_seed=np.asarray(0), map_seed=np.asarray(21), hw=np.asarray(hw))

    # ---------------------------------------------------------------- tiny quirk cases (full tensors)
    rows = special_lines(hw)
    dd_np, ds_np = synth.synth_dense_maps_np(5, *hw)
    dd, ds = torch.from_numpy(dd_np), torch.from_numpy(ds_np)
    for name, kw in (
        ("tiny_default", dict()),
        ("tiny_float_td", dict(cfg=dict(token_distance=12.8, min_length=25.6))),
        ("tiny_align_true", dict(torch_version="1.8.0")),
        ("tiny_two_layers", dict(n_desc_layers=2, wseed=3)),
        ("tiny_max3", dict(cfg=dict(max_keylines=3))),
        ("tiny_noborder", dict(cfg=dict(remove_borders=0, min_length=8))),
    ):
        mm = ref_model(kw.get("wseed", 0), kw.get("n_desc_layers", 1), hw, **kw.get("cfg", {}))
        res, out = run_image(mm, rows.copy(), dd, ds, hw, torch_version=kw.get("torch_version"))
        res["desc_sublines"] = out["desc_sublines"].numpy().copy()
        cfg_items = {f"cfg_{k}": np.asarray(v) for k, v in kw.get("cfg", {}).items()}
        save(name, lines=rows, map_seed=np.asarray(5), hw=np.asarray(hw),
             weight_seed=np.asarray(kw.get("wseed", 0)), n_desc_layers=np.asarray(kw.get("n_desc_layers", 1)),
             align_corners=np.asarray(kw.get("torch_version") is not None), **cfg_items, **res)

    # valid-mask (ndarray) path used by demo_LineTR.py / the dataset builder (quirk 4)
    vm = np.ones(hw)
    vm[:, :330] = 0           # start points of many lines invalid, some end points valid
    mm = ref_model(0, 1, hw)
    res, out = run_image(mm, rows.copy(), dd, ds, hw, valid_mask=vm)
    save("tiny_validmask", lines=rows, map_seed=np.asarray(5), hw=np.asarray(hw), weight_seed=np.asarray(0),
         valid_mask_cols=np.asarray(330), **{k: res[k] for k in ("klines", "sublines", "mat_klines2sublines", "line_desc")})

    # one survivor + max_keylines=-1 -> empty -> default_ret (quirk 23)
    one = rows[[11]].copy()
    out = mm.preprocess(synth.array_to_keylines(one), (1, 1, *hw), {"dense_descriptor": dd, "dense_score": ds})
    ret = mm(out)
    save("tiny_single_line", lines=one, n_klines=np.asarray(len(out["klines"])),
         **{f"ret_{k}_shape": np.asarray(v.shape) for k, v in ret.items()})

    # ---------------------------------------------------------------- cfg5-like: 1280x960, T=41
    hw5 = (960, 1280)
    m5 = ref_model(0, 1, hw5, max_tokens=41, token_distance=8, min_length=16)
    rows = synth.synth_lines(51, 160, hw5[0], hw5[1], 40.0, 327.0)
    dd_np, ds_np = synth.synth_dense_maps_np(51, *hw5)
    res, out = run_image(m5, rows, torch.from_numpy(dd_np), torch.from_numpy(ds_np), hw5)
    keep = {k: res[k] for k in ("klines", "sublines", "pnt_sublines", "mask_sublines", "resp_sublines",
                                "angle_sublines", "score_sublines", "line_desc", "desc_checksum",
                                "desc_abs_checksum")}
    save("cfg5_small", lines=rows, map_seed=np.asarray(51), hw=np.asarray(hw5), weight_seed=np.asarray(0),
         max_tokens=np.asarray(41), **keep)

    # ---------------------------------------------------------------- matcher known-answer cases
    rs = np.random.RandomState(7)
    cases = {}
    d = rs.uniform(0.1, 1.9, (1, 6, 5)).astype(np.float32)
    d[0, 1] = d[0, 0]                  # duplicate rows -> column argmin tie -> first index wins
    d[0, 2, 3] = d[0, 2, 1] = 0.05     # row tie -> first index
    d[0, 4, 4] = np.float32(0.8)       # threshold-equal (strict <) on an otherwise-mutual pair
    d[0, :, 4] = np.maximum(d[0, :, 4], 0.9); d[0, 4, :] = np.maximum(d[0, 4, :], 0.9); d[0, 4, 4] = np.float32(0.8)
    d[0, 5, 0] = -0.3                  # negative -> clipped to 0
    cases["ties"] = d
    cases["big"] = rs.uniform(0, 2, (1, 37, 53)).astype(np.float32)
    cases["empty0"] = np.zeros((1, 0, 4), np.float32)
    cases["empty1"] = np.zeros((1, 3, 0), np.float32)
    arrs = {}
    for k, v in cases.items():
        arrs[f"{k}_dist"] = v
        arrs[f"{k}_mutual"] = nn_matcher_distmat(v, 0.8, True)
        arrs[f"{k}_oneway"] = nn_matcher_distmat(v, 0.8, False)
    p0 = rs.standard_normal((256, 40)).astype(np.float32); p0 /= np.linalg.norm(p0, axis=0, keepdims=True)
    p1 = np.concatenate([p0[:, rs.permutation(40)[:30]] + 0.05 * rs.standard_normal((256, 30)).astype(np.float32),
                         rs.standard_normal((256, 9)).astype(np.float32)], 1)
    p1 /= np.linalg.norm(p1, axis=0, keepdims=True)
    pm, pd = nn_matcher(p0, p1.astype(np.float32), 0.7, True)
    arrs.update(point_desc0=p0, point_desc1=p1.astype(np.float32), point_M=pm, point_D=pd)
    save("matcher_cases", **arrs)


if __name__ == "__main__":
    main()
