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
        score_logits, desc_raw = self.encode(data["image"])
        dense_score, d_nhwc, d_nchw = self.engine().superpoint_heads(score_logits, desc_raw, nhwc=True,
                                                                     nchw=self.want_nchw)
        if d_nchw is None:                       # a view with the reference's shape (not contiguous)
            d_nchw = d_nhwc.permute(0, 3, 1, 2)
        hc, wc = int(score_logits.shape[2]), int(score_logits.shape[3])
        keypoints, scores, descriptors = self._keypoints(dense_score, d_nchw, hc * 8, wc * 8)
        return {"keypoints": keypoints, "scores": scores, "descriptors": descriptors, "dense_descriptor": d_nchw,
                "dense_score": dense_score, "dense_descriptor_nhwc": d_nhwc}

    def _keypoints(self, dense_score, dense_desc, height, width):
        """Key-point branch of the wrapped implementation (superpoint.py:168-187, 195-197), image by image, through its
        own helpers: NMS, threshold, border removal, optional top-k, (row, col) -> (x, y), descriptor sampling."""
        fn, cfg = self._fn, self.config
        nms = fn.simple_nms(dense_score, cfg["nms_radius"])
        kps, scs, descs = [], [], []
        for b in range(nms.shape[0]):
            rc = torch.nonzero(nms[b] > cfg["keypoint_threshold"])
            val = nms[b][rc[:, 0], rc[:, 1]]
            rc, val = fn.remove_borders(rc, val, cfg["remove_borders"], height, width)
            if cfg["max_keypoints"] >= 0:
                rc, val = fn.top_k_keypoints(rc, val, cfg["max_keypoints"])
            xy = rc.flip(1).float()
            kps.append(xy)
            scs.append(val)
            descs.append(fn.sample_descriptors(xy[None], dense_desc[b][None], 8)[0])
        return kps, tuple(scs), descs
