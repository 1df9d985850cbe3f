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
        if superpoint is None:
            superpoint = _frontend("superpoint", "SuperPoint", config.get("superpoint", {}))
            # The host project's SuperPoint (models/superpoint.py:100-205) is wrapped so that its two head post-processing
            # steps run on linetr_superpoint_heads and the descriptor map ALSO comes out channel-last: the tokeniser then
            # needs no NCHW -> NHWC pass.  Same dict as the reference's forward plus 'dense_descriptor_nhwc'.
            # An injected instance is used as given; config['fuse_superpoint_heads'] = False keeps the plain module.
            if config.get("fuse_superpoint_heads", True) and all(hasattr(superpoint, a) for a in
                                                                  ("convPa", "convPb", "convDa", "convDb", "relu", "pool")):
                from .superpoint import FusedHeadSuperPoint
                superpoint = FusedHeadSuperPoint(superpoint)
        self.superpoint = superpoint
        self.lsd = lsd if lsd is not None else _frontend("line_detector", "LSD", config.get("lsd", {}))
        self.linetransformer = LineTransformer(config.get("linetransformer", {}))

    def _describe_fused(self, data, sp, sides, detected):
        """Both images of a pair through ONE fused native call (linetr_describe: tokeniser + descriptor network on the real tokens,
        every tensor of the reference's dict materialised, mat_klines2sublines from the same launch): 45 launches instead of 2 x 44,
        and the pair's GEMMs see 398 rows instead of twice 199.  Every tensor of the two dicts equals the per-image path's bit for bit
        except the descriptors (fp32 round-off: other GEMM tiles for 398 rows; tests/test_gpu_dropin.py).
        Returns the two dicts (preprocess + forward of models/line_transformer.py:225-275), or None when the pair does not
        qualify (an image without lines, maps of different shapes or not on the device)."""
        from .line_process import attach_sub2line, get_angles, keylines_to_array, mask_fits_image
        lt = self.linetransformer
        imgs = [data["image" + s] for s in sides]
        shape = tuple(imgs[0].shape)
        nhwc = all(sp[s].get("dense_descriptor_nhwc") is not None for s in sides)
        key = "dense_descriptor_nhwc" if nhwc else "dense_descriptor"
        dds, dss = [sp[s][key] for s in sides], [sp[s]["dense_score"] for s in sides]
        if tuple(imgs[1].shape) != shape or not all(t.is_cuda for t in dds + dss) or dds[0].shape != dds[1].shape:
            return None
        _, _, height, width = lt.config["image_shape"] = shape   # (LineTransformer.preprocess writes it too: line_transformer.py:258)
        c = lt.config
        td, T = c["token_distance"], c["max_tokens"]
        # a1-a3 (change_cv2_T_np, remove_borders, filter_by_length: line_process.py:203-231, :59-84, :6-21) by the native pre-filter, which
        # is bit-identical to the NumPy glue on klines / lengths / order (images that hold EQUAL lengths are re-ordered by NumPy's own
        # argsort: Engine.prefilter, tie_order); the angles are NumPy's, computed per image as filter_by_length does (:20), because
        # libm's cos / sin may differ from NumPy's in the last ulp
        masks = [data["valid_mask" + s] if isinstance(data["valid_mask" + s], np.ndarray) else None for s in sides]
        if any(m is not None and not mask_fits_image(m, height, width) for m in masks):
            return None                                           # oddly shaped masks: NumPy's own indexing, image by image
        eng = lt.engine(dds[0].device)
        recs, cu_k, cu_n = eng.prefilter([keylines_to_array(detected[s]) for s in sides], height, width,
                                         remove_borders=c["remove_borders"], min_length=c["min_length"], max_keylines=c["max_keylines"],
                                         token_distance=td, max_tokens=T, valid_masks=masks if any(m is not None for m in masks) else None)
        if cu_k[1] == 0 or cu_k[2] == cu_k[1]:
            return None
        for i in range(2):
            r = recs[cu_k[i]:cu_k[i + 1]]
            r["angle"] = get_angles(np.stack([r["sp"], r["ep"]], axis=1))
        align = int(torch.__version__[2]) > 2                     # the reference's own version switch (line_process.py:93)
        tb, ld = eng.describe(recs, cu_k, cu_n, torch.cat(dds), torch.cat(dss), token_distance=td, max_tokens=T,
                              align_corners=align, want_tokens=True, dense_layout="nhwc" if nhwc else "nchw", want_mat=True)
        outs = []
        for i in range(2):      # one indexing call per entry (a torch view costs ~2 us of host time; there are 24 of them)
            k = slice(int(cu_k[i]), int(cu_k[i + 1]))
            n = slice(int(cu_n[i]), int(cu_n[i + 1]))
            mat = attach_sub2line(tb.mat_of(i)[None], tb.sub2line[n])
            outs.append({"klines": tb.klines[None, k], "length_klines": tb.length[None, k], "angles": tb.angles[None, k],
                         "sublines": tb.sublines[None, n], "pnt_sublines": tb.pnt[None, n],
                         "mask_sublines": tb.mask[None, n, :, None], "resp_sublines": tb.resp[None, n, None],
                         "angle_sublines": tb.angle_sub[None, n], "desc_sublines": tb.desc[None, n],
                         "score_sublines": tb.score[None, n, :, None], "mat_klines2sublines": mat,
                         "line_desc": ld[None, n].transpose(1, 2)})
        return outs

    def forward(self, data):
        pred = {}
        sp = {}
        for s in ("0", "1"):
            if "keypoints" + s not in data:
                sp[s] = self.superpoint({"image": data["image" + s]})
                pred.update({k + s: v for k, v in sp[s].items()})
        # Both matchers run BEHIND the descriptors, in one native call that also requests the four device -> host copies of their
        # results (linetr_pair_tail): nothing is queued in front of the line branch's host glue, which is what the pair waits for.
        from .line_process import _token_engine, sub2line_of
        d0, d1 = ((sp[s]["descriptors"] if s in sp else data["descriptors" + s])[0].detach() for s in ("0", "1"))
        on_dev = d0.is_cuda and d0.dim() == 2 and d0.shape[0] == 256 and d0.shape[1] > 0 and d1.shape[1] > 0
        thr_p = self.superpoint.config["nn_threshold"]
        eng_m = _token_engine(d0.device) if on_dev else None     # the matchers need no weights (and no weight-version check)
        # detect + tokenise + describe the images that need it (matching.py:34-41, :52-59)
        sides = [s for s in ("0", "1") if "klines" + s not in data]
        for s in sides:
            if "valid_mask" + s not in data:
                data["valid_mask" + s] = torch.ones_like(data["image" + s])   # a tensor: ignored downstream (matching.py:37-40)
        lt = self.linetransformer

        def auto_lengths(shape):
            """min_length / token_distance follow the image about to be tokenised -- per image, matching.py:29-32 and :45-48."""
            if self.auto_min_length:
                lt.config["min_length"] = max(16, max(shape) / 40)
                lt.config["token_distance"] = max(8, max(shape) / 80)

        detected = {s: self.lsd.detect_torch(data["image" + s]) for s in sides}
        outs = None
        if len(sides) == 2 and tuple(data["image0"].shape) == tuple(data["image1"].shape):
            auto_lengths(data["image0"].shape)                    # one shape: both images get the same two values
            outs = self._describe_fused(data, sp, sides, detected)
        if outs is None:     # one image (anchor cached), an image without lines, host tensors, two image sizes: tokenised image by
            pres = []        # image, each with the thresholds of ITS shape (lt.config ends on the last image's, as in the reference)
            for s in sides:
                auto_lengths(data["image" + s].shape)
                pres.append(lt.preprocess(detected[s], data["image" + s].shape, sp[s], data["valid_mask" + s]))
            outs = lt.forward_many(pres)
        for s, out in zip(sides, outs):
            pred.update({k + s: v for k, v in out.items()})
        data = {**data, **pred}    # (the reference also stacks list entries of this local dict, matching.py:62-64: nothing below reads one)

        # point matches (matching.py:67-74) and line matches (D -> key-line pooling -> mutual NN, matching.py:77-84): ONE native call
        # queued behind the descriptors, ONE synchronisation for the four result arrays
        thr_l = self.linetransformer.config["nn_threshold"]
        ld0, ld1, mat0, mat1 = data["line_desc0"], data["line_desc1"], data["mat_klines2sublines0"], data["mat_klines2sublines1"]
        line_args = (ld0, mat0, ld1, mat1, thr_l)
        K0, K1 = int(mat0.shape[1]), int(mat1.shape[1])
        tail = have_lines = None
        if on_dev:
            # maps made by this package's tokeniser ride along with their (unmodified) matrices; an edited or foreign matrix is read by
            # its contents on the slower path below
            # (a matrix THIS call has just made cannot have been written to by anyone: its map is taken as it is -- also under
            # torch.inference_mode(), where tensors keep no version counter to stamp; a matrix handed in by the caller goes by its stamp)
            s0, s1 = (getattr(m, "_linetr_sub2line", None) if s in sides else sub2line_of(m) for s, m in (("0", mat0), ("1", mat1)))
            have_lines = K0 > 0 and K1 > 0 and s0 is not None and s1 is not None and ld0.is_cuda and ld1.is_cuda
            if have_lines or K0 == 0 or K1 == 0:
                tail = eng_m.pair_tail(d0, d1, float(np.float32(thr_p)), ld0[0].t() if have_lines else None, s0, K0,
                                       ld1[0].t() if have_lines else None, s1, K1, float(np.float32(thr_l)), True)
        if tail is not None:
            dist_h, m01_h, dk_h, m01_lh = eng_m.collect_tail(tail)
            m_p, d_p = match01_to_matrix(m01_h, int(d1.shape[1])), dist_h[None]
            if have_lines:
                m_l, d_l = match01_to_matrix(m01_lh, K1), dk_h[None]
            else:           # a side without key-lines: the reference's matcher returns zeros (nn_matcher.py:9-10)
                m_l, d_l = np.zeros((1, K0, K1)), np.zeros((1, K0, K1), dtype=np.float32)
        else:   # edited / foreign matrices (read by their contents), matrices made under inference_mode, host tensors
            ticket_p = None
            if on_dev:     # the point matcher and its device -> host copies are queued FIRST and travel under the line branch
                dist, m01 = eng_m.match_points(d0, d1, float(np.float32(thr_p)), True)
                ticket_p = eng_m.to_host_async(m01, dist)
            else:
                m_p, d_p = nn_matcher(d0.cpu().numpy(), d1.cpu().numpy(), thr_p, is_mutual_NN=True)
            m_l, d_l = self.match_lines(*line_args)        # its own wait covers everything queued before it on the stream
            if ticket_p is not None:
                m01_h, dist_h = eng_m.collect(ticket_p)
                m_p, d_p = match01_to_matrix(m01_h, int(d1.shape[1])), dist_h[None]
        pred["matches_p"] = torch.from_numpy(m_p)
        pred["matching_scores_p"] = torch.from_numpy(d_p)
        pred["matches_l"] = torch.from_numpy(m_l)
        pred["matching_scores_l"] = torch.from_numpy(d_l)
        return pred

    def forward_batch(self, pairs):
        """Batched counterpart of forward() (section 8(f)-3: the reference's surface is one pair per call).

        `pairs` is a list of dicts with 'image0' / 'image1' ([1,1,H,W], all pairs the same size).  SuperPoint and
        LSD still run per image (out-of-scope front-ends); line tokenisation + description of ALL 2P images is ONE
        fused native call (linetr_prefilter_batch + linetr_describe) and the line matching of all P pairs is ONE
        linetr_match call.  Returns a list of P dicts with the keys forward() produces, except that the dense
        per-token tensors (pnt/mask/desc/score_sublines) are not materialised.  Key-line order is forward()'s: the native
        pre-filter sorts, and images that hold equal lengths are ordered by NumPy's own argsort (Engine.prefilter, tie_order); the
        angles are NumPy's, as in forward() (describe_lines(angles="numpy")): every tensor both return is the same bit for bit,
        up to the descriptors' fp32 round-off (other GEMM tiles for other row counts)."""
        from .line_process import attach_sub2line, keylines_to_array
        lt = self.linetransformer
        P = len(pairs)
        if P == 0:
            return []
        shape = tuple(pairs[0]["image0"].shape)
        sp_out, lines, valid = [], [], []
        for d in pairs:
            for s in ("0", "1"):
                img = d["image" + s]
                if tuple(img.shape) != shape:
                    raise ValueError("forward_batch needs equally sized images")
                sp = self.superpoint({"image": img})
                sp_out.append(sp)
                lines.append(keylines_to_array(self.lsd.detect_torch(img)))
        if self.auto_min_length:
            lt.config["min_length"] = max(16, max(shape) / 40)
            lt.config["token_distance"] = max(8, max(shape) / 80)
        lt.config["image_shape"] = shape
        c = lt.config
        # a producer that already emits the tokeniser's layout (linetr_amd.superpoint.FusedHeadSuperPoint) saves the
        # NCHW -> NHWC pass inside linetr_describe
        layout = "nhwc" if all("dense_descriptor_nhwc" in sp for sp in sp_out) else "nchw"
        dd = torch.cat([sp["dense_descriptor_nhwc" if layout == "nhwc" else "dense_descriptor"] for sp in sp_out])
        ds = torch.cat([sp["dense_score"] for sp in sp_out])
        eng = lt.engine(dd.device)
        off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)
        cat = np.concatenate(lines) if off[-1] else np.zeros((0, 6))
        align = int(torch.__version__[2]) > 2
        tb, ld = eng.describe_lines(cat, off, dd, ds, remove_borders=c["remove_borders"], min_length=c["min_length"],
                                    max_keylines=c["max_keylines"], token_distance=c["token_distance"],
                                    max_tokens=c["max_tokens"], align_corners=align, dense_layout=layout, angles="numpy")
        cu_n, cu_k = tb.cu_n, tb.cu_k
        n, k = np.diff(cu_n), np.diff(cu_k)
        dev = ld.device
        idx0 = torch.cat([torch.arange(cu_n[i], cu_n[i + 1]) for i in range(0, 2 * P, 2)]).to(dev)
        idx1 = torch.cat([torch.arange(cu_n[i], cu_n[i + 1]) for i in range(1, 2 * P, 2)]).to(dev)
        cs = lambda v: np.concatenate([[0], np.cumsum(v)]).astype(np.int32)
        dk, off_dk, m01 = eng.match(ld[idx0], cs(n[0::2]), tb.sub2line[idx0], cs(k[0::2]), ld[idx1], cs(n[1::2]),
                                    tb.sub2line[idx1], cs(k[1::2]), float(np.float32(c["nn_threshold"])), True)
        # point matcher of every pair queued behind the line matcher; ONE synchronisation brings everything to the host
        descs = []
        for sp in sp_out:
            v = sp["descriptors"]
            descs.append((torch.stack(list(v)) if isinstance(v, (list, tuple)) else v)[0].detach())
        thr_p = self.superpoint.config["nn_threshold"]
        pts_on_dev = all(d.is_cuda and d.dim() == 2 and d.shape[0] == 256 and d.shape[1] > 0 for d in descs)
        pts = [eng.match_points(descs[2 * p], descs[2 * p + 1], float(np.float32(thr_p)), True) for p in range(P)] if pts_on_dev else []
        host = eng.to_host(dk, m01, *[t for dist, mp in pts for t in (mp, dist)])
        dk_h, m01_h = host[0], host[1]
        ck0 = cs(k[0::2])
        # mat_klines2sublines of all 2P images in one buffer (line_process.py:168-180: row = key-line, 1 / num_sublines at its
        # sub-lines; the float64 quotient rounded to float32): indices from the host records, one scatter on the device
        recs = tb.recs
        K_all, N_all = int(cu_k[-1]), int(cu_n[-1])
        blk = np.concatenate([[0], np.cumsum(k.astype(np.int64) * n.astype(np.int64))])
        A_flat = torch.zeros((int(blk[-1]),), device=dev)
        if N_all:
            rec_of_sub = np.repeat(np.arange(K_all), recs["n_sub"][:K_all])
            img_s = recs["image"][:K_all][rec_of_sub].astype(np.int64)
            flat = blk[img_s] + recs["line_local"][:K_all][rec_of_sub].astype(np.int64) * n[img_s] + (np.arange(N_all) - cu_n[img_s])
            vals = (1.0 / recs["n_sub"][:K_all][rec_of_sub].astype(np.float64)).astype(np.float32)
            A_flat[torch.from_numpy(flat).to(dev)] = torch.from_numpy(vals).to(dev)
        preds = []
        for p in range(P):
            pred = {}
            for s, img_i in (("0", 2 * p), ("1", 2 * p + 1)):
                pred.update({key + s: v for key, v in sp_out[img_i].items()})
                kk, nn = slice(int(cu_k[img_i]), int(cu_k[img_i + 1])), slice(int(cu_n[img_i]), int(cu_n[img_i + 1]))
                A = attach_sub2line(A_flat[int(blk[img_i]):int(blk[img_i + 1])].view(int(k[img_i]), int(n[img_i]))[None], tb.sub2line[nn])
                pred.update({"klines" + s: tb.klines[None, kk], "length_klines" + s: tb.length[None, kk],
                             "angles" + s: tb.angles[None, kk], "sublines" + s: tb.sublines[None, nn],
                             "resp_sublines" + s: tb.resp[None, nn, None], "angle_sublines" + s: tb.angle_sub[None, nn],
                             "line_desc" + s: ld[None, nn].transpose(1, 2), "mat_klines2sublines" + s: A})
            if pts_on_dev:
                m_p, d_p = match01_to_matrix(host[2 + 2 * p], int(descs[2 * p + 1].shape[1])), host[3 + 2 * p][None]
            else:
                m_p, d_p = nn_matcher(descs[2 * p].cpu().numpy(), descs[2 * p + 1].cpu().numpy(), thr_p, is_mutual_NN=True)
            pred["matches_p"], pred["matching_scores_p"] = torch.from_numpy(m_p), torch.from_numpy(d_p)
            K0, K1 = int(k[2 * p]), int(k[2 * p + 1])
            pred["matches_l"] = torch.from_numpy(match01_to_matrix(m01_h[ck0[p]:ck0[p] + K0], K1))
            pred["matching_scores_l"] = torch.from_numpy(dk_h[off_dk[p]:off_dk[p + 1]].reshape(1, K0, K1).copy())
            for key in list(pred):
                if isinstance(pred[key], (list, tuple)):
                    pred[key] = torch.stack(list(pred[key]))
            preds.append(pred)
        return preds

    def _queue_line_match(self, line_desc0, mat0, line_desc1, mat1, thr):
        """Queues get_dist_matrix + subline2keyline + nn_matcher_distmat of one pair on the device (asynchronous).
        Returns (Dk flat, match01, K0, K1) as device tensors, or None when a side has no key-line."""
        K0, N0 = int(mat0.shape[1]), int(mat0.shape[2])
        K1, N1 = int(mat1.shape[1]), int(mat1.shape[2])
        if K0 == 0 or K1 == 0:
            return None
        # the matcher needs no weights: the weight-less engine serves it (no weight-version check on this call)
        from .line_process import _token_engine, sub2line_of
        eng = _token_engine(line_desc0.device if line_desc0.is_cuda else self.linetransformer._device())
        dev = eng.device
        # matrices made by this package's tokeniser carry their sub-line -> key-line map (valid while the matrix is unmodified)
        s0, s1 = sub2line_of(mat0), sub2line_of(mat1)
        if s0 is not None and s1 is not None:
            d0 = line_desc0[0].to(dev).t()      # [N,256] rows: line_desc is a transposed view of exactly that, so no copy
            d1 = line_desc1[0].to(dev).t()
            dk, _, m01 = eng.match(d0, np.array([0, N0]), s0, np.array([0, K0]), d1, np.array([0, N1]), s1,
                                   np.array([0, K1]), float(np.float32(thr)), True)
            return dk, m01, K0, K1
        # any other matrix (an anchor reloaded from an .npz, an edited one): the reference's three steps one by one, each native --
        # D from the [256,N] descriptors as they are, the pooling from the matrices' CONTENTS (a tokeniser's matrix is recognised
        # on the device, anything else is multiplied out as given), then the mutual-NN kernels
        D, _ = eng.match_points(line_desc0[0].to(dev), line_desc1[0].to(dev), float("inf"), False)
        dk = eng.pool_distmat_dense(D, mat0[0].to(dev), mat1[0].to(dev))
        m01 = eng.match_distmat(dk, float(np.float32(thr)), True)
        return dk.reshape(-1), m01, K0, K1

    def match_lines(self, line_desc0, mat0, line_desc1, mat1, thr):
        """(matches [1,K0,K1] float64, Dk [1,K0,K1] float32) as NumPy, like matching.py:77-84."""
        q = self._queue_line_match(line_desc0, mat0, line_desc1, mat1, thr)
        if q is None:
            K0, K1 = int(mat0.shape[1]), int(mat1.shape[1])
            return np.zeros((1, K0, K1)), np.zeros((1, K0, K1), dtype=np.float32)
        dk, m01, K0, K1 = q
        from .line_process import _token_engine
        m01_h, dk_h = _token_engine(dk.device).to_host(m01, dk)   # one synchronisation for both results
        return match01_to_matrix(m01_h, K1), dk_h.reshape(1, K0, K1)
