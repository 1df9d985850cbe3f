"""Drop-in for the reference's ``models.matching.Matching`` (models/matching.py:8-86).

Same constructor config, same ``.superpoint`` / ``.lsd`` / ``.linetransformer`` attributes, same
``forward(data)`` contract and output keys.  The line branch (tokenise -> describe -> distance matrix ->
key-line pooling -> mutual NN) and the point NN matcher run on the HIP kernels.  SuperPoint and the
OpenCV LSD detector are upstream producers outside this package's scope:
they are taken from the host project (``models.superpoint`` / ``models.line_detector`` next to this shim)
or injected through the ``superpoint=`` / ``lsd=`` arguments.
"""
from __future__ import annotations

import importlib

import numpy as np
import torch

from .line_transformer import LineTransformer
from .nn_matcher import match01_to_matrix, nn_matcher


def _frontend(module: str, cls: str, cfg: dict):
    for pkg in ("models", "linetr_amd.frontends"):
        try:
            return getattr(importlib.import_module(f"{pkg}.{module}"), cls)(cfg)
        except ModuleNotFoundError:
            continue
    raise ImportError(
        f"{cls} is not part of linetr_amd (it is an upstream producer of the LineTR hot path). Keep the host "
        f"project's models/{module}.py next to the models/ shim, or pass an instance to Matching(..., "
        f"{'superpoint' if cls == 'SuperPoint' else 'lsd'}=...).")


class Matching(torch.nn.Module):
    """Image matching front-end: SuperPoint + LSD + Line-Transformer + NN matchers."""

    def __init__(self, config={}, superpoint=None, lsd=None):
        super().__init__()
        self.auto_min_length = config["auto_min_length"]
        self.superpoint = superpoint if superpoint is not None else _frontend("superpoint", "SuperPoint",
                                                                              config.get("superpoint", {}))
        self.lsd = lsd if lsd is not None else _frontend("line_detector", "LSD", config.get("lsd", {}))
        self.linetransformer = LineTransformer(config.get("linetransformer", {}))

    def _describe_lines(self, image, pred_sp, valid_mask):
        lt = self.linetransformer
        shape = image.shape
        if self.auto_min_length:                                  # matching.py:30-32
            lt.config["min_length"] = max(16, max(shape) / 40)
            lt.config["token_distance"] = max(8, max(shape) / 80)
        klines_cv = self.lsd.detect_torch(image)
        return lt(lt.preprocess(klines_cv, shape, pred_sp, valid_mask))

    def forward(self, data):
        pred = {}
        sp = {}
        for s in ("0", "1"):
            if "keypoints" + s not in data:
                sp[s] = self.superpoint({"image": data["image" + s]})
                pred.update({k + s: v for k, v in sp[s].items()})
        for s in ("0", "1"):
            if "klines" + s not in data:
                img = data["image" + s]
                if "valid_mask" + s not in data:
                    data["valid_mask" + s] = torch.ones_like(img)   # a tensor: ignored downstream (matching.py:37-40)
                out = self._describe_lines(img, sp[s], data["valid_mask" + s])
                pred.update({k + s: v for k, v in out.items()})
        data = {**data, **pred}
        for k in data:
            if isinstance(data[k], (list, tuple)):
                data[k] = torch.stack(data[k])

        # point matches (nn_matcher on the device)
        m_p, d_p = nn_matcher(data["descriptors0"][0].detach().cpu().numpy(), data["descriptors1"][0].detach().cpu().numpy(),
                              self.superpoint.config["nn_threshold"], is_mutual_NN=True)
        pred["matches_p"] = torch.from_numpy(m_p)
        pred["matching_scores_p"] = torch.from_numpy(d_p)

        # line matches: D -> key-line pooling -> mutual NN in one native call
        m_l, d_l = self.match_lines(data["line_desc0"], data["mat_klines2sublines0"], data["line_desc1"],
                                    data["mat_klines2sublines1"], self.linetransformer.config["nn_threshold"])
        pred["matches_l"] = torch.from_numpy(m_l)
        pred["matching_scores_l"] = torch.from_numpy(d_l)
        return pred

    def match_lines(self, line_desc0, mat0, line_desc1, mat1, thr):
        """(matches [1,K0,K1] float64, Dk [1,K0,K1] float32) as NumPy, like matching.py:77-84."""
        K0, N0 = int(mat0.shape[1]), int(mat0.shape[2])
        K1, N1 = int(mat1.shape[1]), int(mat1.shape[2])
        if K0 == 0 or K1 == 0:
            return np.zeros((1, K0, K1)), np.zeros((1, K0, K1), dtype=np.float32)
        eng = self.linetransformer.engine(line_desc0.device if line_desc0.is_cuda else None)
        dev = eng.device
        d0 = line_desc0[0].to(dev).t().contiguous()
        d1 = line_desc1[0].to(dev).t().contiguous()
        s0 = mat0[0].to(dev).argmax(dim=0).to(torch.int32)
        s1 = mat1[0].to(dev).argmax(dim=0).to(torch.int32)
        dk, _, m01 = eng.match(d0, np.array([0, N0]), s0, np.array([0, K0]), d1, np.array([0, N1]), s1,
                               np.array([0, K1]), float(np.float32(thr)), True)
        return match01_to_matrix(m01.cpu().numpy(), K1), dk.view(1, K0, K1).cpu().numpy()
