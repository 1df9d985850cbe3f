"""SuperPoint with the fused head producer.

`FusedHeadSuperPoint(sp)` wraps an existing SuperPoint module -- the reference's `models.superpoint.SuperPoint`,
unchanged, or anything with the same conv attributes -- and returns the same dict from `forward`
(models/superpoint.py:146-201), but computes the two dense maps with `linetr_superpoint_heads`:

    softmax / drop dustbin / depth-to-space  (superpoint.py:161-167)   -> 'dense_score'
    F.normalize over channels                (superpoint.py:190-193)   -> 'dense_descriptor' (NCHW, as the reference)
                                                                        + 'dense_descriptor_nhwc' [B,Hc,Wc,256]

The extra NHWC key is the layout the line tokeniser samples from, so `Engine.describe_lines(...,
dense_layout='nhwc')` skips its NCHW->NHWC pass.  The convolutions stay in PyTorch (MIOpen); key-point extraction
(`simple_nms`, `remove_borders`, `top_k_keypoints`, `sample_descriptors`) is delegated to the wrapped module's own
module-level functions -- those are the reference's and out of this path's scope.
"""
from __future__ import annotations

import sys

import torch
from torch import nn


class FusedHeadSuperPoint(nn.Module):
    def __init__(self, superpoint: nn.Module, engine=None, helpers=None, want_nchw: bool = True):
        super().__init__()
        self.sp = superpoint
        self.config = superpoint.config
        self.want_nchw = want_nchw
        self._engine = engine
        # simple_nms / remove_borders / top_k_keypoints / sample_descriptors of the wrapped implementation
        self._fn = helpers if helpers is not None else sys.modules[type(superpoint).__module__]

    def engine(self):
        if self._engine is None:
            from .engine import Engine
            dev = next(self.sp.parameters()).device
            self._engine = Engine.heads_only(dev)
        return self._engine

    def encode(self, image):
        """shared encoder + the two head convolutions (superpoint.py:148-160, 188-189): raw head outputs."""
        sp, relu = self.sp, self.sp.relu
        x = relu(sp.conv1a(image)); x = relu(sp.conv1b(x)); x = sp.pool(x)
        x = relu(sp.conv2a(x)); x = relu(sp.conv2b(x)); x = sp.pool(x)
        x = relu(sp.conv3a(x)); x = relu(sp.conv3b(x)); x = sp.pool(x)
        x = relu(sp.conv4a(x)); x = relu(sp.conv4b(x))
        return sp.convPb(relu(sp.convPa(x))), sp.convDb(relu(sp.convDa(x)))

    def forward(self, data):
        fn, cfg = self._fn, self.config
        score_logits, desc_raw = self.encode(data["image"])
        dense_score, d_nhwc, d_nchw = self.engine().superpoint_heads(score_logits, desc_raw, nhwc=True,
                                                                     nchw=self.want_nchw)
        if d_nchw is None:                       # a view with the reference's shape (not contiguous)
            d_nchw = d_nhwc.permute(0, 3, 1, 2)
        b, hc, wc = score_logits.shape[0], score_logits.shape[2], score_logits.shape[3]
        scores = fn.simple_nms(dense_score, cfg["nms_radius"])
        keypoints = [torch.nonzero(s > cfg["keypoint_threshold"]) for s in scores]
        scores = [s[tuple(k.t())] for s, k in zip(scores, keypoints)]
        keypoints, scores = list(zip(*[fn.remove_borders(k, s, cfg["remove_borders"], hc * 8, wc * 8)
                                       for k, s in zip(keypoints, scores)]))
        if cfg["max_keypoints"] >= 0:
            keypoints, scores = list(zip(*[fn.top_k_keypoints(k, s, cfg["max_keypoints"])
                                           for k, s in zip(keypoints, scores)]))
        keypoints = [torch.flip(k, [1]).float() for k in keypoints]
        descriptors = [fn.sample_descriptors(k[None], d[None], 8)[0] for k, d in zip(keypoints, d_nchw)]
        return {"keypoints": keypoints, "scores": scores, "descriptors": descriptors, "dense_descriptor": d_nchw,
                "dense_score": dense_score, "dense_descriptor_nhwc": d_nhwc}
