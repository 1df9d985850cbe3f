"""Host-side driver of the native library: owns the handle, device workspaces and the var-len batch
plumbing.  PyTorch is used for device memory, streams and H2D/D2H copies only; every arithmetic step
of the hot path is a HIP kernel behind the C ABI (include/linetr_hip.h).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _native as nat

D = 256

DEFAULT_MODEL = dict(descriptor_dim=256, keyline_encoder=[32, 64, 128, 256], n_heads=4,
                     n_line_descriptive_layers=1, d_inner=1024, n_sig_layers=7, image_shape=[480, 640], bn_batch_stats=False)


def _as_numpy_f32(v):
    if torch.is_tensor(v):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


@dataclass
class TokenBatch:
    """Device tensors produced by the tokeniser for a batch of images (images concatenated)."""
    n_images: int
    max_tokens: int
    cu_k: np.ndarray          # [B+1] host prefix sums of key-lines
    cu_n: np.ndarray          # [B+1] host prefix sums of sub-lines
    recs: np.ndarray          # host LineRec array [K]
    klines: torch.Tensor      # [K,2,2]
    length: torch.Tensor      # [K]
    angles: torch.Tensor      # [K,2]
    sublines: torch.Tensor    # [N,2,2]
    pnt: torch.Tensor         # [N,T,2]
    mask: torch.Tensor        # [N,T+1]
    resp: torch.Tensor        # [N]
    angle_sub: torch.Tensor   # [N,2]
    desc: torch.Tensor        # [N,T,256]
    score: torch.Tensor       # [N,T]
    sub2line: torch.Tensor    # [N] int32, key-line index inside the image
    mat: torch.Tensor = None  # mat_klines2sublines with want_mat=True: [K,N] for one image, the flat per-image blocks for a batch (mat_of)
    extra: dict = field(default_factory=dict)

    @property
    def K(self):
        return int(self.cu_k[-1])

    @property
    def N(self):
        return int(self.cu_n[-1])

    def mat_of(self, i: int) -> torch.Tensor:
        """mat_klines2sublines [K_i,N_i] of image i (describe(..., want_mat=True))."""
        k = np.diff(self.cu_k).astype(np.int64)
        n = np.diff(self.cu_n).astype(np.int64)
        off = int((k[:i] * n[:i]).sum())
        return self.mat.reshape(-1)[off:off + int(k[i] * n[i])].view(int(k[i]), int(n[i]))

    def c_tokens(self) -> nat.Tokens:
        t = nat.Tokens()
        for k in ("klines", "length", "angles", "sublines", "pnt", "mask", "resp", "angle_sub", "desc", "score"):
            setattr(t, k, getattr(self, k).data_ptr())
        t.mat = self.mat.data_ptr() if self.mat is not None else None
        return t


def repack_like_numpy(L, recs, cu_k, cu_n, i, rows, height, width, border, min_length, max_keylines, token_distance,
                      max_tokens, valid_mask=None):
    """Image i of a pre-filtered batch holds equal lengths: redo a1-a3 the way the per-image path does (NumPy's own argsort,
    models/line_process.py:6-21) and overwrite the image's records in place.  Equal lengths mean equal token counts, so the
    number of key-lines / sub-lines / tokens of the image -- and every later image's offsets -- stay as they are."""
    from . import line_process as lp
    vm = np.asarray(valid_mask, dtype=np.float64) if valid_mask is not None else None
    kl = lp.filter_by_length(lp.remove_borders(lp.lines_from_rows(rows.copy()), border, height, width, vm), min_length,
                             max_keylines)
    k0, k1 = int(cu_k[i]), int(cu_k[i + 1])
    if len(kl["klines"]) != k1 - k0:
        raise RuntimeError(f"pre-filter: NumPy keeps {len(kl['klines'])} lines of image {i}, the native pass {k1 - k0}")
    if k1 == k0:
        return
    n_out = C.c_int32()
    tok = int(recs["first_tok"][k0])
    n_tok = int(recs["n_tok"][k0:k1].sum())
    a, b, c = (np.ascontiguousarray(kl[k], dtype=np.float64) for k in ("klines", "length_klines", "angles"))
    nat.check(L.linetr_pack_lines(nat.np_ptr(a), nat.np_ptr(b), nat.np_ptr(c), k1 - k0, float(token_distance), int(max_tokens), i,
                                  int(cu_n[i]), tok, nat.np_ptr(recs[k0:k1]), C.byref(n_out)), L)
    if n_out.value != int(cu_n[i + 1] - cu_n[i]) or int(recs["n_tok"][k0:k1].sum()) != n_tok:
        raise RuntimeError(f"pre-filter: re-ordering image {i} changed its sub-line / token count")


class Engine:
    """One native model instance on one GPU."""

    def __init__(self, state_dict, device="cuda:0", lib_path=None, **model_cfg):
        """lib_path: another build of the library (nat.EXPERIMENTS_LIB_PATH for tools / experiment tests)."""
        cfg = {**DEFAULT_MODEL, **model_cfg}
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("linetr_amd.Engine needs a HIP device (torch device 'cuda:N'); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.cfg = cfg
        L = nat.lib(lib_path)
        mc = nat.ModelConfig()
        mc.d_model, mc.n_heads, mc.d_inner = cfg["descriptor_dim"], cfg["n_heads"], cfg["d_inner"]
        mc.n_sig_layers, mc.n_desc_layers = cfg["n_sig_layers"], cfg["n_line_descriptive_layers"]
        for i, c in enumerate(cfg["keyline_encoder"]):
            mc.enc_channels[i] = c
        shape = cfg["image_shape"]
        mc.norm_height, mc.norm_width = int(shape[-2]), int(shape[-1])
        mc.bn_batch_stats = int(bool(cfg["bn_batch_stats"]))     # a training-mode handle: forward_train_tensors only
        names, arrs = [], []
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                continue
            names.append(k.encode())
            arrs.append(_as_numpy_f32(v))
        n = len(names)
        c_names = (C.c_char_p * n)(*names)
        c_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        c_numel = (C.c_int64 * n)(*[a.size for a in arrs])
        h = C.c_void_p()
        nat.check(L.linetr_create(C.byref(mc), n, c_names, c_ptrs, c_numel, self.device.index, C.byref(h)))
        self._h = h
        self._L = L
        self._ws = {}

    @classmethod
    def heads_only(cls, device="cuda:0"):
        """An Engine without a LineTR model: only the weight-free entry points work (superpoint_heads, match_points,
        match_distmat).  Used by FusedHeadSuperPoint when no LineTransformer engine is at hand."""
        self = cls.__new__(cls)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("linetr_amd.Engine needs a HIP device (torch device 'cuda:N'); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.cfg = dict(DEFAULT_MODEL)
        self._L = nat.lib()
        self._h = None
        self._ws = {}
        return self

    def close(self):
        if getattr(self, "_h", None):
            self._L.linetr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, tag: str, nbytes: int) -> torch.Tensor:
        ws = self._ws.get(tag)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=self.device)
            self._ws[tag] = ws
        return ws

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
        return t

    # ------------------------------------------------------------------ host pre-filter
    def _pinned_slot(self, nbytes):
        """ring of pinned staging buffers; a slot is reused only after its last H2D copy has completed."""
        ring = self.__dict__.setdefault("_ring", {"i": 0, "slots": [None] * 12})
        ring["i"] = (ring["i"] + 1) % len(ring["slots"])
        slot = ring["slots"][ring["i"]]
        if slot is None or slot["buf"].numel() < nbytes:
            slot = {"buf": torch.empty(int(nbytes * 1.5) + 4096, dtype=torch.uint8, pin_memory=True), "event": None}
            ring["slots"][ring["i"]] = slot
        if slot["event"] is not None:
            slot["event"].synchronize()
            slot["event"] = None
        return slot

    def prefilter(self, lines6, height, width, *, remove_borders, min_length, max_keylines, token_distance,
                  max_tokens, valid_masks=None, offsets=None, n_threads=0, tie_order="numpy"):
        """a1-a3 for a batch.  `lines6` is a list of [K_i,6] float64 arrays, or one concatenated [sum K,6] array
        with `offsets` [B+1].  Returns (recs, cu_k, cu_n); recs lives in pinned memory ready for an async H2D.

        tie_order: the length sort of models/line_process.py:15-16 is np.argsort, whose order among EQUAL lengths is NumPy's
        business (unstable, CPU-dispatched).  "numpy" (default): the images the native pre-filter reports as holding equal
        lengths (linetr_prefilter_tied_images; rare with a real detector) are re-ordered with NumPy itself and re-packed, so the
        batched path gives the reference's rows on this machine, like the per-image path does.  "stable": keep the native order
        (stable ascending argsort, reversed) -- machine-independent."""
        if tie_order not in ("numpy", "stable"):
            raise ValueError("tie_order must be 'numpy' or 'stable'")
        if offsets is None:
            lens = [len(l) for l in lines6]
            offsets = np.zeros(len(lens) + 1, dtype=np.int32)
            np.cumsum(lens, out=offsets[1:])
            cat = (np.concatenate([np.asarray(l, dtype=np.float64).reshape(-1, 6) for l in lines6])
                   if len(lens) else np.zeros((0, 6)))
        else:
            cat = lines6
            offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        cat = np.ascontiguousarray(cat, dtype=np.float64)
        B = len(offsets) - 1
        cap = int(offsets[-1])
        rec_bytes = max(cap, 1) * nat.REC_DTYPE.itemsize
        slot = self._pinned_slot(rec_bytes + 8 * (B + 1) + 64)
        host = slot["buf"].numpy()
        recs = host[:rec_bytes].view(nat.REC_DTYPE)
        cu_k = host[rec_bytes:rec_bytes + 4 * (B + 1)].view(np.int32)
        cu_n = host[rec_bytes + 4 * (B + 1):rec_bytes + 8 * (B + 1)].view(np.int32)
        vm_ptrs = None
        keep = []
        if valid_masks is not None:
            arr = (C.c_void_p * B)()
            for i, vm in enumerate(valid_masks):
                if vm is not None:
                    if np.ndim(vm) != 2 or np.shape(vm) != (int(height), int(width)):
                        # prefilter_core reads vm[y * width + x]: any other shape would be read out of bounds or mis-addressed
                        raise ValueError(f"valid mask {i} has shape {np.shape(vm)}, the native pre-filter needs ({int(height)}, "
                                         f"{int(width)}) (other shapes: the NumPy remove_borders of linetr_amd.line_process)")
                    vm = np.ascontiguousarray(vm, dtype=np.float64)
                    keep.append(vm)
                    arr[i] = vm.ctypes.data
            vm_ptrs = arr
        nat.check(self._L.linetr_prefilter_batch(nat.np_ptr(cat), nat.np_ptr(offsets), B, int(height), int(width),
                                                 int(remove_borders), float(min_length), int(max_keylines), vm_ptrs,
                                                 float(token_distance), int(max_tokens), int(n_threads),
                                                 nat.np_ptr(recs), cap, nat.np_ptr(cu_k), nat.np_ptr(cu_n)), self._L)
        if tie_order == "numpy":
            tied = np.empty(max(B, 1), dtype=np.int32)
            n_tied = self._L.linetr_prefilter_tied_images(nat.np_ptr(tied), B)
            for i in tied[:n_tied].tolist():
                repack_like_numpy(self._L, recs, cu_k, cu_n, i, cat[offsets[i]:offsets[i + 1]], int(height), int(width),
                                  remove_borders, min_length, max_keylines, token_distance, max_tokens,
                                  valid_masks[i] if valid_masks is not None else None)
        K = int(cu_k[-1])
        out = recs[:K]
        self._last_host = {"slot": slot, "recs_ptr": recs.ctypes.data, "rec_bytes": rec_bytes, "B": B}
        return out, cu_k.copy(), cu_n.copy()

    def pack(self, klines, length, angles, token_distance, max_tokens, image=0, sub_base=0):
        """records for already-filtered lines (float64 arrays, reference layout).  They are written into a pinned staging
        slot together with the two prefix sums, so that tokenize() uploads them with ONE asynchronous copy."""
        K = len(klines)
        rec_bytes = max(K, 1) * nat.REC_DTYPE.itemsize
        slot = self._pinned_slot(rec_bytes + 16 + 64)
        host = slot["buf"].numpy()
        recs = host[:rec_bytes].view(nat.REC_DTYPE)
        n_out = C.c_int32()
        kl = np.ascontiguousarray(klines, dtype=np.float64)
        ln = np.ascontiguousarray(length, dtype=np.float64)
        an = np.ascontiguousarray(angles, dtype=np.float64)
        nat.check(self._L.linetr_pack_lines(nat.np_ptr(kl), nat.np_ptr(ln), nat.np_ptr(an), K, float(token_distance),
                                            int(max_tokens), int(image), int(sub_base), 0, nat.np_ptr(recs), C.byref(n_out)), self._L)
        cu = host[rec_bytes:rec_bytes + 16].view(np.int32)
        cu[:] = (0, K, 0, n_out.value)                     # cu_k | cu_n of the one image
        self._last_host = {"slot": slot, "recs_ptr": recs.ctypes.data, "rec_bytes": rec_bytes, "B": 1}
        return recs[:K], n_out.value

    def pack_many(self, lines, token_distance, max_tokens):
        """pack() for several images of one batch (each entry: {'klines', 'length_klines', 'angles'} float64, already filtered and
        ordered): one pinned blob  records | cu_k | cu_n, laid out like prefilter()'s, so describe() uploads it with one async copy."""
        B = len(lines)
        ks = [len(l["klines"]) for l in lines]
        K = int(sum(ks))
        rec_bytes = max(K, 1) * nat.REC_DTYPE.itemsize
        slot = self._pinned_slot(rec_bytes + 8 * (B + 1) + 64)
        host = slot["buf"].numpy()
        recs = host[:rec_bytes].view(nat.REC_DTYPE)
        cu_k = host[rec_bytes:rec_bytes + 4 * (B + 1)].view(np.int32)
        cu_n = host[rec_bytes + 4 * (B + 1):rec_bytes + 8 * (B + 1)].view(np.int32)
        cu_k[0] = cu_n[0] = 0
        tok = 0
        n_out = C.c_int32()
        for i, l in enumerate(lines):
            k0 = int(cu_k[i])
            kl = np.ascontiguousarray(l["klines"], dtype=np.float64)
            ln = np.ascontiguousarray(l["length_klines"], dtype=np.float64)
            an = np.ascontiguousarray(l["angles"], dtype=np.float64)
            nat.check(self._L.linetr_pack_lines(nat.np_ptr(kl), nat.np_ptr(ln), nat.np_ptr(an), ks[i], float(token_distance),
                                                int(max_tokens), i, int(cu_n[i]), tok, nat.np_ptr(recs[k0:k0 + max(ks[i], 1)]),
                                                C.byref(n_out)), self._L)
            cu_k[i + 1] = k0 + ks[i]
            cu_n[i + 1] = cu_n[i] + n_out.value
            tok += int(recs["n_tok"][k0:k0 + ks[i]].sum())
        self._last_host = {"slot": slot, "recs_ptr": recs.ctypes.data, "rec_bytes": rec_bytes, "B": B}
        return recs[:K], cu_k.copy(), cu_n.copy()

    # ------------------------------------------------------------------ device stages
    def tokenize(self, recs, cu_k, cu_n, dense_desc, dense_score, *, token_distance, max_tokens, align_corners=False,
                 sample_desc=True, dense_layout="nchw", want_mat=False, clip_shape=None) -> TokenBatch:
        """line_tokenizer on the device.  dense_desc [B,256,H/8,W/8] (dense_layout='nchw') or [B,H/8,W/8,256]
        ('nhwc', the producer's layout: no transposition pass), dense_score [B,H,W].  clip_shape: the (height, width) the
        reference's `image_shape` argument carries when it is not the maps' shape (it only sets the end-point clip)."""
        B = len(cu_k) - 1
        K, N, T = int(cu_k[-1]), int(cu_n[-1]), int(max_tokens)
        dense_desc = self._f32(dense_desc)
        dense_score = self._f32(dense_score)
        if dense_score.dim() == 2:
            dense_score = dense_score[None]
        if dense_desc.dim() == 3:
            dense_desc = dense_desc[None]
        if dense_desc.shape[0] != B or dense_score.shape[0] != B:
            raise ValueError("dense maps must have one entry per image")
        H, W = int(dense_score.shape[-2]), int(dense_score.shape[-1])
        nhwc = dense_layout == "nhwc"
        want = (B, H // 8, W // 8, D) if nhwc else (B, D, H // 8, W // 8)
        if tuple(dense_desc.shape) != want:
            raise ValueError(f"dense_descriptor shape {tuple(dense_desc.shape)} does not match {want} ({dense_layout})")
        dev = self.device
        f = dict(dtype=torch.float32, device=dev)
        if want_mat and B != 1:
            raise ValueError("want_mat needs a single-image call (the reference's matrix is per image)")
        # the small token tensors are views of ONE allocation (a dozen torch.empty calls cost ~40 us of host time per image on
        # the drop-in path); every view starts on a 16-byte boundary
        shapes = [("klines", (K, 2, 2)), ("length", (K,)), ("angles", (K, 2)), ("sublines", (N, 2, 2)), ("pnt", (N, T, 2)),
                  ("mask", (N, T + 1)), ("resp", (N,)), ("angle_sub", (N, 2)), ("score", (N, T)), ("sub2line", (N,))]
        if want_mat:
            shapes.append(("mat", (K, N)))                  # written by extra blocks of the tokeniser's own launch
        sizes = [(math.prod(sh) + 3) // 4 * 4 for _, sh in shapes]
        pool = torch.empty((sum(sizes),), **f)
        views, o = {}, 0
        for (name, sh), sz in zip(shapes, sizes):
            views[name] = pool[o:o + math.prod(sh)].view(sh)
            o += sz
        views["sub2line"] = views["sub2line"].view(torch.int32)
        tb = TokenBatch(n_images=B, max_tokens=T, cu_k=np.asarray(cu_k, np.int32), cu_n=np.asarray(cu_n, np.int32), recs=recs,
                        desc=torch.empty((N, T, D), **f) if sample_desc else torch.empty((0,), **f), **views)
        if K == 0 or N == 0:
            return tb
        last = getattr(self, "_last_host", None)
        if last is not None and K > 0 and recs.ctypes.data == last["recs_ptr"] and last["B"] == B:
            # records + prefix sums sit in one pinned blob: ONE async H2D, no host/device synchronisation
            nb = last["rec_bytes"] + 8 * (B + 1)
            d_blob = torch.empty(nb, dtype=torch.uint8, device=dev)
            d_blob.copy_(last["slot"]["buf"][:nb], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            last["slot"]["event"] = ev
            d_recs = d_blob
            tb.extra["d_cu_n"] = d_blob[last["rec_bytes"] + 4 * (B + 1):].view(torch.int32)
        else:
            d_recs = torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(dev)
        nbytes = self._L.linetr_tokenize_workspace_bytes(B, H, W, N)
        ws = self._workspace("tok", nbytes)
        ct = tb.c_tokens()
        if not sample_desc:
            ct.desc = None
        with torch.cuda.device(self.device):     # a weight-less engine has no handle: the current device is used
            nat.check(self._L.linetr_tokenize(self._h, d_recs.data_ptr(), K, N, float(token_distance), T,
                                              dense_desc.data_ptr(), dense_score.data_ptr(), B, H, W,
                                              int(clip_shape[0]) if clip_shape is not None else 0,
                                              int(clip_shape[1]) if clip_shape is not None else 0, int(bool(align_corners)),
                                              int(nhwc), ct, tb.sub2line.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), self._L)
        tb.extra["d_recs"] = d_recs  # keep alive until the stream has consumed it
        return tb

    def describe(self, recs, cu_k, cu_n, dense_desc, dense_score, *, token_distance, max_tokens, align_corners=False,
                 want_tokens=False, dense_layout="nchw", want_mat=False, pipeline_slot=None):
        """Fused tokenise + descriptor network for a batch (linetr_describe): real tokens only, descriptors sampled
        on the fly.  Returns (TokenBatch, line_desc [N,256]).  With want_tokens=False the dense [N,T,...] token
        tensors (pnt / mask / score / desc) are not materialised (zero-sized in the TokenBatch).

        pipeline_slot = (slot, n_slots): linetr_describe_submit -- the batch runs on the library's stage streams, overlapped with the
        batches submitted to the other slots, and is NOT joined: the returned tensors may be read only after describe_join(slot)
        (class DescribePipeline does the bookkeeping)."""
        B = len(cu_k) - 1
        K, N, T = int(cu_k[-1]), int(cu_n[-1]), int(max_tokens)
        dense_desc = self._f32(dense_desc)
        dense_score = self._f32(dense_score)
        if dense_score.dim() == 2:
            dense_score = dense_score[None]
        if dense_desc.dim() == 3:
            dense_desc = dense_desc[None]
        if dense_desc.shape[0] != B or dense_score.shape[0] != B:
            raise ValueError("dense maps must have one entry per image")
        H, W = int(dense_score.shape[-2]), int(dense_score.shape[-1])
        nhwc = dense_layout == "nhwc"
        want = (B, H // 8, W // 8, D) if nhwc else (B, D, H // 8, W // 8)
        if tuple(dense_desc.shape) != want:
            raise ValueError(f"dense_descriptor shape {tuple(dense_desc.shape)} does not match {want} ({dense_layout})")
        dev = self.device
        f = dict(dtype=torch.float32, device=dev)
        # the small outputs and line_desc are views of ONE allocation (every view on a 16-byte boundary): a dozen torch.empty
        # calls are ~40 us of host time on the latency path of a single pair
        shapes = [("ld", (N, D)), ("klines", (K, 2, 2)), ("length", (K,)), ("angles", (K, 2)), ("sublines", (N, 2, 2)),
                  ("resp", (N,)), ("angle_sub", (N, 2)), ("sub2line", (N,))]
        if want_tokens:
            shapes += [("pnt", (N, T, 2)), ("mask", (N, T + 1)), ("score", (N, T))]
        if want_mat:     # the per-image [K_i,N_i] blocks back to back, written by extra blocks of the tokeniser's launch (<= 8 images)
            shapes.append(("mat", (int((np.diff(np.asarray(cu_k, np.int64)) * np.diff(np.asarray(cu_n, np.int64))).sum()),)))
        sizes = [(math.prod(sh) + 3) // 4 * 4 for _, sh in shapes]
        pool = torch.empty((sum(sizes),), **f)
        views, o = {}, 0
        for (name, sh), sz in zip(shapes, sizes):
            views[name] = pool[o:o + math.prod(sh)].view(sh)
            o += sz
        views["sub2line"] = views["sub2line"].view(torch.int32)
        ld = views.pop("ld")
        z = pool[:0]
        for name in ("pnt", "mask", "score"):
            views.setdefault(name, z)
        tb = TokenBatch(n_images=B, max_tokens=T, cu_k=np.asarray(cu_k, np.int32), cu_n=np.asarray(cu_n, np.int32), recs=recs,
                        desc=torch.empty((N, T, D), **f) if want_tokens else z, **views)
        if K == 0 or N == 0:
            return tb, ld
        n_real = int(recs["n_tok"][:K].sum())
        d_recs, d_cu = self._upload_recs(recs, K, B, tb)
        ct = tb.c_tokens()
        if not want_tokens:
            ct.pnt = ct.mask = ct.desc = ct.score = None
        cu_k32 = np.ascontiguousarray(cu_k, dtype=np.int32)
        if want_mat:
            ct.h_cu_klines = cu_k32.ctypes.data
            if B == 1:
                tb.mat = tb.mat.view(K, N)
        nbytes = self._L.linetr_describe_workspace_bytes(self._h, B, H, W, N, n_real)
        cu = np.ascontiguousarray(cu_n, dtype=np.int32)
        args = (self._h, d_recs.data_ptr(), K, N, n_real, nat.np_ptr(cu), d_cu.data_ptr() if d_cu is not None else None, B,
                float(token_distance), T, dense_desc.data_ptr(), dense_score.data_ptr(), H, W, int(bool(align_corners)), int(nhwc), ct,
                tb.sub2line.data_ptr(), ld.data_ptr())
        if pipeline_slot is None:
            ws = self._workspace("desc", nbytes)
            nat.check(self._L.linetr_describe(*args, ws.data_ptr(), ws.numel(), self._stream()), self._L)
        else:
            slot, n_slots = (int(v) for v in pipeline_slot)
            tag = f"desc_pipe{slot}"
            old = self._ws.get(tag)
            if old is not None and old.numel() < nbytes:
                # the slot's workspace has to grow: its previous batch may still be running on the library's streams, which the caching
                # allocator knows nothing about -- let the device drain before the old block goes back to the pool (rare: sizes settle)
                torch.cuda.synchronize(self.device)
            ws = self._workspace(tag, nbytes)      # one workspace per slot: n_slots batches are in flight
            tb.extra["dense"] = (dense_desc, dense_score)          # read by the stage streams after this call returns
            nat.check(self._L.linetr_describe_submit(*args, ws.data_ptr(), ws.numel(), slot, n_slots, self._stream()), self._L)
        return tb, ld

    def describe_join(self, slot: int):
        """The current stream waits for the batch last submitted to `slot` (linetr_describe_join)."""
        nat.check(self._L.linetr_describe_join(self._h, int(slot), self._stream()), self._L)

    def describe_lines(self, lines6, offsets, dense_desc, dense_score, *, remove_borders, min_length, max_keylines,
                       token_distance, max_tokens, align_corners=False, want_tokens=False,
                       dense_layout="nchw", angles="native", pipeline_slot=None):
        """prefilter + describe for a batch given as one [sum K,6] array + row offsets [B+1].

        angles: "native" -- (cos 2theta, sin 2theta) from the host pre-filter's libm (the throughput path: nothing of the step runs in
        Python); "numpy" -- recomputed with NumPy from the filtered lines in one vectorised call, exactly what get_angles of the
        per-image surface computes (models/line_process.py:28-41): the two can differ in the last float64 ulp (<= 1.2e-7 after the
        float32 cast), and the batched drop-in surface (Matching.forward_batch) asks for NumPy's so that it returns forward()'s tensors.

        pipeline_slot = (slot, n_slots): see describe() / DescribePipeline.
        Returns (TokenBatch, line_desc [N,256]) for the whole batch."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        B = len(offsets) - 1
        H, W = int(dense_score.shape[-2]), int(dense_score.shape[-1])
        kw = dict(remove_borders=remove_borders, min_length=min_length, max_keylines=max_keylines,
                  token_distance=token_distance, max_tokens=max_tokens)
        if angles not in ("native", "numpy"):
            raise ValueError("angles must be 'native' or 'numpy'")
        recs, cu_k, cu_n = self.prefilter(lines6, H, W, offsets=offsets, **kw)
        if angles == "numpy" and len(recs):
            from .line_process import get_angles
            recs["angle"] = get_angles(np.stack([recs["sp"], recs["ep"]], axis=1))     # (records live in the pinned upload slot)
        return self.describe(recs, cu_k, cu_n, dense_desc, dense_score, token_distance=token_distance,
                             max_tokens=max_tokens, align_corners=align_corners, want_tokens=want_tokens,
                             dense_layout=dense_layout, pipeline_slot=pipeline_slot)

    def _upload_recs(self, recs, K, B, tb):
        """H2D of the line records (+ the sub-line prefix sums when they sit in the same pinned blob)."""
        dev = self.device
        last = getattr(self, "_last_host", None)
        if last is not None and K > 0 and recs.ctypes.data == last["recs_ptr"] and last["B"] == B:
            nb = last["rec_bytes"] + 8 * (B + 1)
            d_blob = torch.empty(nb, dtype=torch.uint8, device=dev)
            d_blob.copy_(last["slot"]["buf"][:nb], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            last["slot"]["event"] = ev
            d_cu = d_blob[last["rec_bytes"] + 4 * (B + 1):].view(torch.int32)
            tb.extra["d_recs"], tb.extra["d_cu_n"] = d_blob, d_cu
            tb.extra["d_cu_k"] = d_blob[last["rec_bytes"]:last["rec_bytes"] + 4 * (B + 1)].view(torch.int32)
            return d_blob, d_cu
        d_recs = torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(dev)
        tb.extra["d_recs"] = d_recs
        return d_recs, None

    def forward_tensors(self, sublines, pnt, resp, angle_sub, desc, score, cu_n, out=None, d_cu_n=None) -> torch.Tensor:
        """LineTransformer.forward on flat tensors; returns line_desc [N,256] (row-major)."""
        N, T = int(pnt.shape[0]), int(pnt.shape[1])
        t = nat.Tokens()
        keep = [self._f32(x) for x in (sublines, pnt, resp, angle_sub, desc, score)]
        t.sublines, t.pnt, t.resp, t.angle_sub, t.desc, t.score = [x.data_ptr() for x in keep]
        cu = np.ascontiguousarray(cu_n, dtype=np.int32)
        if int(cu[-1]) != N:
            raise ValueError("cu_n does not match the number of sub-lines")
        if out is None:
            out = torch.empty((N, D), dtype=torch.float32, device=self.device)
        if N == 0:
            return out
        nbytes = self._L.linetr_forward_workspace_bytes(self._h, N, T)
        ws = self._workspace("fwd", nbytes)
        nat.check(self._L.linetr_forward(self._h, C.byref(t), nat.np_ptr(cu), d_cu_n.data_ptr() if d_cu_n is not None else None,
                                         len(cu) - 1, T, out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), self._L)
        return out

    def bn_stats_floats(self) -> int:
        return int(self._L.linetr_bn_stats_floats(self._h))

    def forward_train_tensors(self, sublines, pnt, resp, angle_sub, desc, score, cu_n, bn_running, momentum=0.1, want_batch_stats=False):
        """The training-time forward (train.py:127,163-164) on flat tensors, on an Engine made with bn_batch_stats=True: BatchNorm on
        the statistics of THIS batch; `bn_running` (float32 device tensor of bn_stats_floats() entries: per BatchNorm layer
        mean[C] | var[C]; word encoder's four layers, line encoder's four, then one per signature layer) is updated in place.  Returns line_desc [N,256], and with want_batch_stats the batch mean
        | biased variance packed the same way."""
        N, T = int(pnt.shape[0]), int(pnt.shape[1])
        t = nat.Tokens()
        keep = [self._f32(x) for x in (sublines, pnt, resp, angle_sub, desc, score)]
        t.sublines, t.pnt, t.resp, t.angle_sub, t.desc, t.score = [x.data_ptr() for x in keep]
        cu = np.ascontiguousarray(cu_n, dtype=np.int32)
        if int(cu[-1]) != N:
            raise ValueError("cu_n does not match the number of sub-lines")
        n_stats = self.bn_stats_floats()
        if (bn_running.dtype != torch.float32 or bn_running.device != self.device or not bn_running.is_contiguous()
                or bn_running.numel() != n_stats):
            raise ValueError(f"bn_running must be a contiguous float32 tensor of {n_stats} entries on {self.device}")
        out = torch.empty((N, D), dtype=torch.float32, device=self.device)
        batch = torch.empty((n_stats,), dtype=torch.float32, device=self.device) if want_batch_stats else None
        if N == 0:
            return (out, batch) if want_batch_stats else out
        ws = self._workspace("fwd", self._L.linetr_forward_train_workspace_bytes(self._h, N, T))
        nat.check(self._L.linetr_forward_train(self._h, C.byref(t), nat.np_ptr(cu), None, len(cu) - 1, T, float(momentum),
                                               bn_running.data_ptr(), batch.data_ptr() if batch is not None else None,
                                               out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), self._L)
        return (out, batch) if want_batch_stats else out

    def forward(self, tb: TokenBatch, out=None) -> torch.Tensor:
        return self.forward_tensors(tb.sublines, tb.pnt, tb.resp, tb.angle_sub, tb.desc, tb.score, tb.cu_n, out,
                                    tb.extra.get("d_cu_n"))

    def match(self, desc0, cu_n0, sub2line0, cu_k0, desc1, cu_n1, sub2line1, cu_k1, thr, mutual=True):
        """Match image i of side 0 with image i of side 1 for all i.  desc* are [N,256] row-major.
        Returns (Dk_flat float32 device, off_dk host int64 [P+1], match01 int32 device [K0_total])."""
        P = len(cu_n0) - 1
        assert len(cu_n1) - 1 == P
        i64 = np.int64
        if P == 1:                      # latency path of a single pair: no diffs, cumsums, reductions (~15 us of NumPy calls)
            n0, k0, n1, k1 = int(cu_n0[1] - cu_n0[0]), int(cu_k0[1] - cu_k0[0]), int(cu_n1[1] - cu_n1[0]), int(cu_k1[1] - cu_k1[0])
            dims = np.array([[n0, k0, n1, k1]], dtype=np.int32)
            off_n0, off_n1, off_k0 = (np.array([int(c[0])], dtype=i64) for c in (cu_n0, cu_n1, cu_k0))
            off_dk = np.array([0, k0 * k1], dtype=i64)
            sum_nn, sum_k = n0 * n1, k0 + k1
        else:
            dims = np.zeros((P, 4), dtype=np.int32)
            dims[:, 0] = np.diff(cu_n0); dims[:, 1] = np.diff(cu_k0)
            dims[:, 2] = np.diff(cu_n1); dims[:, 3] = np.diff(cu_k1)
            off_n0 = np.ascontiguousarray(cu_n0[:-1], dtype=i64)
            off_n1 = np.ascontiguousarray(cu_n1[:-1], dtype=i64)
            off_k0 = np.ascontiguousarray(cu_k0[:-1], dtype=i64)
            kk = dims[:, 1].astype(i64) * dims[:, 3].astype(i64)
            off_dk = np.zeros(P + 1, dtype=i64)
            np.cumsum(kk, out=off_dk[1:])
            sum_nn = int((dims[:, 0].astype(i64) * dims[:, 2].astype(i64)).sum())
            sum_k = int(dims[:, 1].sum() + dims[:, 3].sum())
        dk = torch.empty((max(int(off_dk[-1]), 1),), dtype=torch.float32, device=self.device)
        m01 = torch.empty((max(int(cu_k0[-1]), 1),), dtype=torch.int32, device=self.device)
        if P == 0:
            return dk[:0], off_dk, m01[:0]
        nbytes = self._L.linetr_match_workspace_bytes(P, sum_nn, 0, sum_k)
        ws = self._workspace("match", nbytes)
        d0, d1 = self._f32(desc0), self._f32(desc1)
        nat.check(self._L.linetr_match(self._h, P, nat.np_ptr(dims), d0.data_ptr(), nat.np_ptr(off_n0),
                                       sub2line0.data_ptr(), d1.data_ptr(), nat.np_ptr(off_n1), sub2line1.data_ptr(),
                                       float(thr), int(bool(mutual)), dk.data_ptr(), nat.np_ptr(off_dk[:-1].copy()),
                                       m01.data_ptr(), nat.np_ptr(off_k0), ws.data_ptr(), ws.numel(), self._stream()), self._L)
        return dk[:int(off_dk[-1])], off_dk, m01[:int(cu_k0[-1])]

    def match_offsets(self, desc_flat, s2l_flat, dims, off_n0, off_s0, off_n1, off_s1, thr, mutual=True):
        """linetr_match_gathered: P pairs whose descriptors (rows of desc_flat [R,256]) and key-line maps (elements of
        s2l_flat int32) sit at arbitrary offsets of the same two buffers -- the form global matching over the all-gathered
        set uses (parallel.global_match).  dims [P,4] = (n0,k0,n1,k1).
        Returns (Dk_flat, off_dk host [P+1], match01 int32 [sum k0], off_k0 host [P+1])."""
        P = len(dims)
        i64 = np.int64
        if P == 1:
            # latency path of a single pair: the seven small host tables of a given (dims, offsets) are built once and kept (a
            # NumPy array costs ~1 us to make; the call is ~45 us end to end)
            key1 = (int(dims[0][0]), int(dims[0][1]), int(dims[0][2]), int(dims[0][3]), int(off_n0[0]), int(off_s0[0]), int(off_n1[0]),
                    int(off_s1[0]))
            cache = self.__dict__.setdefault("_pair_tables", {})
            tabs = cache.get(key1)
            if tabs is None:
                if len(cache) > 256:
                    cache.clear()
                n0, k0, n1, k1 = key1[:4]
                tabs = cache[key1] = (np.array([key1[:4]], dtype=np.int32), np.array([0, k0 * k1], dtype=i64), np.array([0, k0], dtype=i64),
                                      np.array([key1[4]], dtype=i64), np.array([key1[6]], dtype=i64), np.array([key1[5]], dtype=i64),
                                      np.array([key1[7]], dtype=i64), self._L.linetr_match_workspace_bytes(1, n0 * n1, 0, k0 + k1))
            dims1, off_dk, off_k0, o0, o1, s0, s1, ws_bytes = tabs
            dk = torch.empty((max(int(off_dk[1]), 1),), dtype=torch.float32, device=self.device)
            m01 = torch.empty((max(int(off_k0[1]), 1),), dtype=torch.int32, device=self.device)
            ws = self._workspace("match", ws_bytes)
            d = desc_flat if (desc_flat.dtype == torch.float32 and desc_flat.is_contiguous() and desc_flat.device == self.device) \
                else self._f32(desc_flat)
            nat.check(self._L.linetr_match_gathered(self._h, 1, dims1.ctypes.data, d.data_ptr(), o0.ctypes.data, s2l_flat.data_ptr(),
                                                    s0.ctypes.data, d.data_ptr(), o1.ctypes.data, s2l_flat.data_ptr(),
                                                    s1.ctypes.data, float(thr), int(bool(mutual)), dk.data_ptr(),
                                                    off_dk.ctypes.data, m01.data_ptr(), off_k0.ctypes.data, ws.data_ptr(), ws.numel(),
                                                    self._stream()), self._L)
            return dk[:int(off_dk[1])], off_dk, m01[:int(off_k0[1])], off_k0
        dims = np.ascontiguousarray(dims, dtype=np.int32).reshape(P, 4)
        off_dk = np.zeros(P + 1, dtype=i64)
        np.cumsum(dims[:, 1].astype(i64) * dims[:, 3].astype(i64), out=off_dk[1:])
        off_k0 = np.zeros(P + 1, dtype=i64)
        np.cumsum(dims[:, 1].astype(i64), out=off_k0[1:])
        sum_nn = int((dims[:, 0].astype(i64) * dims[:, 2].astype(i64)).sum()) if P else 0
        sum_k = int(dims[:, 1].sum() + dims[:, 3].sum()) if P else 0
        dk = torch.empty((max(int(off_dk[-1]), 1),), dtype=torch.float32, device=self.device)
        m01 = torch.empty((max(int(off_k0[-1]), 1),), dtype=torch.int32, device=self.device)
        if P == 0:
            return dk[:0], off_dk, m01[:0], off_k0
        key = (P, sum_nn, sum_k)
        if self.__dict__.get("_match_ws_key") != key:     # the workspace query is a ctypes call: cache it per shape
            self._match_ws_key = key
            self._match_ws_bytes = self._L.linetr_match_workspace_bytes(P, sum_nn, 0, sum_k)
        ws = self._workspace("match", self._match_ws_bytes)
        c64 = lambda a: np.ascontiguousarray(a, dtype=i64)
        o0, o1, s0, s1 = c64(off_n0), c64(off_n1), c64(off_s0), c64(off_s1)
        d = desc_flat if (desc_flat.dtype == torch.float32 and desc_flat.is_contiguous() and desc_flat.device == self.device) \
            else self._f32(desc_flat)
        nat.check(self._L.linetr_match_gathered(self._h, P, dims.ctypes.data, d.data_ptr(), o0.ctypes.data, s2l_flat.data_ptr(),
                                                s0.ctypes.data, d.data_ptr(), o1.ctypes.data, s2l_flat.data_ptr(),
                                                s1.ctypes.data, float(thr), int(bool(mutual)), dk.data_ptr(),
                                                off_dk.ctypes.data, m01.data_ptr(), off_k0.ctypes.data, ws.data_ptr(), ws.numel(),
                                                self._stream()), self._L)
        return dk[:int(off_dk[-1])], off_dk, m01[:int(off_k0[-1])], off_k0

    def pool_distmat(self, dist: torch.Tensor, sub2line0: torch.Tensor, k0: int, sub2line1: torch.Tensor, k1: int):
        """subline2keyline on the device: dist [n0,n1] + the two sub-line -> key-line maps -> Dk [k0,k1]."""
        d = self._f32(dist)
        n0, n1 = int(d.shape[0]), int(d.shape[1])
        dk = torch.empty((k0, k1), dtype=torch.float32, device=self.device)
        if k0 == 0 or k1 == 0:
            return dk
        ws = self._workspace("pool", self._L.linetr_pool_distmat_workspace_bytes(k0, k1))
        s0 = sub2line0.to(device=self.device, dtype=torch.int32)
        s1 = sub2line1.to(device=self.device, dtype=torch.int32)
        with torch.cuda.device(self.device):
            nat.check(self._L.linetr_pool_distmat(self._h, d.data_ptr(), n0, n1, s0.data_ptr(), k0, s1.data_ptr(), k1,
                                                  dk.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), self._L)
        return dk

    def pool_distmat_dense(self, dist: torch.Tensor, mat0: torch.Tensor, mat1: torch.Tensor):
        """subline2keyline on the device from the two mat_klines2sublines MATRICES ([K,N] float32): a tokeniser's matrix is
        reduced to its map and pooled by the segmented-mean kernel, any other is multiplied out as given
        (linetr_pool_distmat_dense; decided on the device, asynchronous).  Returns Dk [k0,k1]."""
        d, a0, a1 = self._f32(dist), self._f32(mat0), self._f32(mat1)
        n0, n1 = int(d.shape[0]), int(d.shape[1])
        k0, k1 = int(a0.shape[0]), int(a1.shape[0])
        if tuple(a0.shape) != (k0, n0) or tuple(a1.shape) != (k1, n1):
            raise ValueError(f"subline2keyline: shapes {tuple(a0.shape)} @ {tuple(d.shape)} @ {tuple(a1.shape)}^T do not chain")
        dk = torch.empty((k0, k1), dtype=torch.float32, device=self.device)
        if k0 == 0 or k1 == 0:
            return dk
        ws = self._workspace("pool", self._L.linetr_pool_distmat_dense_workspace_bytes(k0, n0, k1, n1))
        with torch.cuda.device(self.device):
            nat.check(self._L.linetr_pool_distmat_dense(self._h, d.data_ptr(), n0, n1, a0.data_ptr(), k0, a1.data_ptr(), k1,
                                                        dk.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), self._L)
        return dk

    def match_distmat(self, dist: torch.Tensor, thr, mutual=True):
        """nn_matcher_distmat on a device matrix [n0,n1]: match01 [n0] (device, int32; -1 = no match)."""
        d = self._f32(dist)
        n0, n1 = int(d.shape[0]), int(d.shape[1])
        m01 = torch.full((n0,), -1, dtype=torch.int32, device=self.device)
        if n0 == 0 or n1 == 0:
            return m01
        ws = self._workspace("match_dm", self._L.linetr_match_distmat_workspace_bytes(n0, n1))
        with torch.cuda.device(self.device):
            nat.check(self._L.linetr_match_distmat(self._h, d.data_ptr(), n0, n1, float(thr), int(bool(mutual)), m01.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), self._stream()), self._L)
        return m01

    def sample_descriptors(self, points: torch.Tensor, dense_desc: torch.Tensor, *, align_corners=False, dense_layout="nchw"):
        """sample_descriptors (line_process.py:86-98) for n points [n,2] of one image; dense_desc [256,Hc,Wc] or
        [Hc,Wc,256]; returns [n,256]."""
        pts = self._f32(points.reshape(-1, 2))
        dd = self._f32(dense_desc)
        nhwc = dense_layout == "nhwc"
        Hc, Wc = (int(dd.shape[0]), int(dd.shape[1])) if nhwc else (int(dd.shape[1]), int(dd.shape[2]))
        n = int(pts.shape[0])
        out = torch.empty((n, D), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        ws = self._workspace("sample", self._L.linetr_sample_descriptors_workspace_bytes(Hc, Wc, int(nhwc)) + 256)
        with torch.cuda.device(self.device):
            nat.check(self._L.linetr_sample_descriptors(self._h, pts.data_ptr(), n, dd.data_ptr(), Hc, Wc, int(bool(align_corners)),
                                                        int(nhwc), out.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()), self._L)
        return out

    def to_host_async(self, *tensors):
        """Queues the device -> host copies of `tensors` into a pinned staging buffer on the current stream and returns a ticket for
        collect().  Nothing is waited for: a result that is ready early (the point matcher's, queued before the line branch) travels
        while the device works on what was queued behind it."""
        sizes = [t.numel() * t.element_size() for t in tensors]
        offs = [0]
        for b in sizes:
            offs.append(offs[-1] + (b + 255) // 256 * 256)
        ring = self.__dict__.setdefault("_host_ring", {"i": 0, "bufs": [None] * 4, "gen": [0] * 4})
        i = ring["i"] = (ring["i"] + 1) % len(ring["bufs"])
        ring["gen"][i] += 1                       # a ticket on this slot that was never collected is stale from here on
        stage = ring["bufs"][i]
        if stage is None or stage.numel() < offs[-1]:
            stage = ring["bufs"][i] = torch.empty(offs[-1] * 2 + 4096, dtype=torch.uint8, pin_memory=True)
        views = []
        for t, o, b in zip(tensors, offs[:-1], sizes):
            v = stage[o:o + b].view(t.dtype).view(t.shape)
            v.copy_(t.contiguous(), non_blocking=True)
            views.append(v)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return views, ev, ring["gen"], i, ring["gen"][i]

    @staticmethod
    def collect(ticket):
        """Waits for a to_host_async ticket and returns its tensors as NumPy arrays (copies: the staging buffer is reused).
        At most three younger tickets may be taken before a ticket is collected (a ring of four staging buffers): a ticket whose
        buffer has been handed out again raises instead of returning another call's data."""
        views, ev, gens, i, gen = ticket
        ev.synchronize()
        if gens[i] != gen:
            raise RuntimeError("to_host_async ticket collected too late: its staging buffer has been reused (collect within three "
                               "younger tickets)")
        return [v.numpy().copy() for v in views]

    def to_host(self, *tensors):
        """Device tensors -> NumPy arrays with ONE synchronisation: every tensor is copied asynchronously into a pinned staging
        buffer, then the copies are waited for once (`.cpu()` per tensor would synchronise per tensor)."""
        return self.collect(self.to_host_async(*tensors))

    def pair_tail(self, pdesc0_cn, pdesc1_cn, thr_p, ldesc0, s2l0, k0, ldesc1, s2l1, k1, thr_l, mutual=True):
        """The matching tail of Matching.forward in ONE native call (linetr_pair_tail): point matcher on the two [256,n] SuperPoint
        descriptor sets, line matcher on the two [N,256] line-descriptor sets with their sub-line -> key-line maps, and the device ->
        host copies of the four results into one pinned block.  Asynchronous; returns a ticket for collect_tail().  A branch is
        skipped when its descriptors are None."""
        pts = pdesc0_cn is not None and pdesc1_cn is not None
        lns = ldesc0 is not None and ldesc1 is not None and k0 > 0 and k1 > 0
        p0, p1 = (self._f32(pdesc0_cn), self._f32(pdesc1_cn)) if pts else (None, None)
        np0, np1 = (int(p0.shape[1]), int(p1.shape[1])) if pts else (0, 0)
        l0, l1 = (self._f32(ldesc0), self._f32(ldesc1)) if lns else (None, None)
        n0, n1 = (int(l0.shape[0]), int(l1.shape[0])) if lns else (0, 0)
        if not lns:
            k0 = k1 = 0
        key = (np0, np1, n0, int(k0), n1, int(k1))
        cache = self.__dict__.setdefault("_tail_tables", {})
        tab = cache.get(key)
        if tab is None:
            if len(cache) > 256:
                cache.clear()
            offs = (C.c_int64 * 4)()
            out_bytes = int(self._L.linetr_pair_tail_output_bytes(np0, np1, int(k0), int(k1), offs))
            tab = cache[key] = (out_bytes, tuple(int(v) for v in offs), int(self._L.linetr_pair_tail_workspace_bytes(*key)))
        out_bytes, offs, ws_bytes = tab
        ring = self.__dict__.setdefault("_host_ring", {"i": 0, "bufs": [None] * 4, "gen": [0] * 4})
        i = ring["i"] = (ring["i"] + 1) % len(ring["bufs"])
        ring["gen"][i] += 1
        stage = ring["bufs"][i]
        if stage is None or stage.numel() < out_bytes:
            stage = ring["bufs"][i] = torch.empty(out_bytes * 2 + 4096, dtype=torch.uint8, pin_memory=True)
        ws = self._workspace("tail", ws_bytes)
        s0 = s2l0 if (lns and s2l0.dtype == torch.int32 and s2l0.device == self.device and s2l0.is_contiguous()) else \
            (s2l0.to(device=self.device, dtype=torch.int32).contiguous() if lns else None)
        s1 = s2l1 if (lns and s2l1.dtype == torch.int32 and s2l1.device == self.device and s2l1.is_contiguous()) else \
            (s2l1.to(device=self.device, dtype=torch.int32).contiguous() if lns else None)
        with torch.cuda.device(self.device):
            nat.check(self._L.linetr_pair_tail(self._h, p0.data_ptr() if pts else None, np0, p1.data_ptr() if pts else None, np1, float(thr_p),
                                               l0.data_ptr() if lns else None, n0, s0.data_ptr() if lns else None, int(k0),
                                               l1.data_ptr() if lns else None, n1, s1.data_ptr() if lns else None, int(k1), float(thr_l),
                                               int(bool(mutual)), stage.data_ptr(), stage.numel(), ws.data_ptr(), ws.numel(), self._stream()),
                      self._L)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        return stage, offs, key, ev, ring["gen"], i, ring["gen"][i], (p0, p1, l0, l1, s0, s1)     # (inputs kept alive until collected)

    @staticmethod
    def collect_tail(ticket):
        """Waits for a pair_tail ticket: (point distances [np0,np1] f32, point match01 [np0] i32, Dk [k0,k1] f32, line match01 [k0] i32)
        as NumPy arrays (copies); the entries of a skipped branch are empty."""
        stage, offs, (np0, np1, _n0, k0, _n1, k1), ev, gens, i, gen, _keep = ticket
        ev.synchronize()
        if gens[i] != gen:
            raise RuntimeError("pair_tail ticket collected too late: its staging buffer has been reused")
        host = stage.numpy()
        f = lambda o, n, dt: host[o:o + 4 * n].view(dt).copy()
        return (f(offs[0], np0 * np1, np.float32).reshape(np0, np1), f(offs[1], np0, np.int32),
                f(offs[2], k0 * k1, np.float32).reshape(k0, k1), f(offs[3], k0, np.int32))

    def match_points(self, desc0_cn: torch.Tensor, desc1_cn: torch.Tensor, thr, mutual=True):
        """nn_matcher on [256,n] descriptors; returns (dist [n0,n1] device, match01 [n0] device)."""
        d0, d1 = self._f32(desc0_cn), self._f32(desc1_cn)
        n0, n1 = int(d0.shape[1]), int(d1.shape[1])
        dist = torch.empty((n0, n1), dtype=torch.float32, device=self.device)
        m01 = torch.full((n0,), -1, dtype=torch.int32, device=self.device)
        if n0 == 0 or n1 == 0:
            return dist, m01
        need = 4 * (n0 + n1) * (D + 1) + 2048 + self._L.linetr_match_workspace_bytes(1, n0 * n1, 0, n0 + n1)
        ws = self._workspace("match", need)
        nat.check(self._L.linetr_match_points(self._h, d0.data_ptr(), n0, d1.data_ptr(), n1, float(thr),
                                              int(bool(mutual)), dist.data_ptr(), m01.data_ptr(), ws.data_ptr(),
                                              ws.numel(), self._stream()), self._L)
        return dist, m01

    def superpoint_heads(self, score_logits: torch.Tensor = None, desc_raw: torch.Tensor = None, *, nhwc=True,
                         nchw=False):
        """Fused SuperPoint head post-processing (models/superpoint.py:161-167, 190-193).

        score_logits [B,65,Hc,Wc] (convPb output) -> dense_score [B,8Hc,8Wc];
        desc_raw [B,256,Hc,Wc] (convDb output) -> L2-normalised descriptors as [B,Hc,Wc,256] (`nhwc`, the layout
        `describe_lines(..., dense_layout='nhwc')` consumes without a transposition pass) and/or [B,256,Hc,Wc]
        (`nchw`, the reference's 'dense_descriptor').  Returns (dense_score, desc_nhwc, desc_nchw), None where not
        requested."""
        sl = self._f32(score_logits) if score_logits is not None else None
        dr = self._f32(desc_raw) if desc_raw is not None else None
        ref = sl if sl is not None else dr
        if ref is None:
            raise ValueError("superpoint_heads needs at least one head")
        B, _, Hc, Wc = (int(v) for v in ref.shape)
        if sl is not None and tuple(sl.shape) != (B, 65, Hc, Wc):
            raise ValueError(f"score_logits must be [B,65,Hc,Wc], got {tuple(sl.shape)}")
        if dr is not None and tuple(dr.shape) != (B, D, Hc, Wc):
            raise ValueError(f"desc_raw must be [B,{D},Hc,Wc], got {tuple(dr.shape)}")
        score = torch.empty((B, Hc * 8, Wc * 8), dtype=torch.float32, device=self.device) if sl is not None else None
        o_nhwc = torch.empty((B, Hc, Wc, D), dtype=torch.float32, device=self.device) if (dr is not None and nhwc) else None
        o_nchw = torch.empty((B, D, Hc, Wc), dtype=torch.float32, device=self.device) if (dr is not None and nchw) else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(self.device):     # a heads-only engine has no handle: the current device is used
            nat.check(self._L.linetr_superpoint_heads(self._h, ptr(sl), ptr(dr), B, Hc, Wc, ptr(score), ptr(o_nhwc),
                                                      ptr(o_nchw), self._stream()), self._L)
        return score, o_nhwc, o_nchw

    PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16x6": 2, "f16x3": 3}

    def set_precision(self, mode: str):
        """arithmetic mode of the dense contractions: 'f32' | 'bf16x6' (fp32-faithful, default) | 'bf16x3'."""
        nat.check(self._L.linetr_set_precision(self._h, self.PRECISIONS[mode]), self._L)

    def get_precision(self) -> str:
        code = self._L.linetr_get_precision(self._h)
        return {v: k for k, v in self.PRECISIONS.items()}[code]

    def debug_posenc(self, which: str, in0, in1, in2=None):
        """Layers 1-3 of the word ('word': points [rows,2], scores [rows]) or line ('line': sub-lines [rows,2,2],
        resp [rows], angles [rows,2]) positional encoder alone -> [rows,128] (unit tests of the fused MLP kernel)."""
        a, b = self._f32(in0), self._f32(in1)
        c = self._f32(in2) if in2 is not None else None
        rows = int(b.shape[0])
        out = torch.empty((rows, 128), dtype=torch.float32, device=self.device)
        nat.check(self._L.linetr_debug_posenc(self._h, {"word": 0, "line": 1}[which], a.data_ptr(), b.data_ptr(),
                                                 c.data_ptr() if c is not None else None, rows, out.data_ptr(),
                                                 self._stream()), self._L)
        return out

    def debug_gemm(self, A, W, bias=None, residual=None, act=0, cache_weights=False, out=None):
        """Y = act(A @ W.T + bias) (+ residual) on the library's MFMA GEMM (diagnostics / unit tests).
        A / out / residual may be row-strided views (stride(0) multiple of 4, stride(1) == 1)."""
        W = self._f32(W)
        if A.dtype != torch.float32 or A.device != self.device or A.stride(1) != 1:
            A = self._f32(A)
        M, K = A.shape
        N = W.shape[0]
        Y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=self.device)
        b = self._f32(bias) if bias is not None else None
        r = residual
        if r is not None and (r.stride(0) != Y.stride(0) or r.stride(1) != 1):
            r = self._f32(r) if Y.stride(0) == N else None
            assert r is not None, "residual must share the output's row stride"
        nat.check(self._L.linetr_debug_gemm(self._h, A.data_ptr(), A.stride(0), W.data_ptr(),
                                            b.data_ptr() if b is not None else None,
                                            r.data_ptr() if r is not None else None, Y.data_ptr(), Y.stride(0), M, N, K,
                                            int(act), int(cache_weights), self._stream()), self._L)
        return Y

    # ------------------------------------------------------------------ split-tile operands (csrc/lt_st_image.h; the GEMM on them: experiments/csrc/lt_gemm_st.h)
    def to_st(self, X):
        """fp32 [rows, K] -> ST image (uint8 tensor); K % 32 == 0."""
        X = self._f32(X)
        rows, K = X.shape
        out = torch.empty(int(self._L.linetr_st_bytes(rows, K)), dtype=torch.uint8, device=self.device)
        nat.check(self._L.linetr_debug_to_st(self._h, X.data_ptr(), X.stride(0), rows, K, out.data_ptr(), self._stream()), self._L)
        return out

    def from_st(self, st, rows, K):
        X = torch.empty((rows, K), dtype=torch.float32, device=self.device)
        nat.check(self._L.linetr_debug_from_st(self._h, st.data_ptr(), rows, K, X.data_ptr(), K, self._stream()), self._L)
        return X

    def gemm_st(self, A1, K1, W, M, N, A2=None, K2=0, bias=None, residual=None, act=0, out_st=None, out=None):
        """act([A1 | A2] W^T + bias) (+ residual) on ST images; returns the ST image `out_st` or the fp32 matrix `out`."""
        b = self._f32(bias) if bias is not None else None
        nat.check(self._L.linetr_debug_gemm_st(
            self._h, A1.data_ptr(), K1, A2.data_ptr() if A2 is not None else None, K2, W.data_ptr(),
            b.data_ptr() if b is not None else None, residual.data_ptr() if residual is not None else None,
            out_st.data_ptr() if out_st is not None else None, out.data_ptr() if out is not None else None,
            out.stride(0) if out is not None else 0, M, N, int(act), self._stream()), self._L)
        return out_st if out_st is not None else out

    # ------------------------------------------------------------------ profiling
    def set_profiling(self, on: bool):
        nat.check(self._L.linetr_set_profiling(self._h, int(on)), self._L)

    def get_profile(self):
        arr = (nat.ProfileEntry * 64)()
        n = C.c_int32()
        nat.check(self._L.linetr_get_profile(self._h, arr, 64, C.byref(n)), self._L)
        return [dict(name=arr[i].name.decode(), calls=arr[i].calls, ms=arr[i].ms, flops=arr[i].flops,
                     bytes=arr[i].bytes) for i in range(min(n.value, 64))]


class DescribePipeline:
    """Software pipeline over CONSECUTIVE batches of Engine.describe_lines (linetr_describe_submit / linetr_describe_join, SURVEY.md
    section 7 step 5): a batch is cut into stages that run on their own streams, so batch i + 1's front (layout pass, tokeniser, token
    MLP, pooling -- half of it HBM-bound) runs on the GPU under batch i's line-signature network (MFMA-bound).  Each batch is described
    whole -- every GEMM sees the full batch -- and its results are those of describe_lines bit for bit; what is traded is depth - 1
    batches of latency:

        pipe = DescribePipeline(engine)            # depth 2: two batches in flight
        for batch in batches:
            done = pipe.submit(lines6, offsets, dense_desc, dense_score, ...)   # -> (TokenBatch, line_desc) of an EARLIER batch, or None
            if done: consume(*done)
        for done in pipe.drain(): consume(*done)

    The tensors a submit returns are ordered on the current stream like describe_lines' own."""

    def __init__(self, engine: "Engine", depth: int = 2):
        if not 2 <= depth <= engine._L.linetr_pipeline_max_slots():
            raise ValueError(f"depth must be 2 .. {engine._L.linetr_pipeline_max_slots()}")
        self.eng, self.depth = engine, int(depth)
        self.i = 0
        self.inflight = []            # [(slot, (tb, ld))] submitted and not joined yet, oldest first

    def _join(self, entry):
        slot, (tb, ld) = entry
        if tb.K > 0 and tb.N > 0:     # an empty batch queues nothing
            self.eng.describe_join(slot)
        return tb, ld

    def submit(self, *args, **kw):
        slot = self.i % self.depth
        self.i += 1
        self.inflight.append((slot, self.eng.describe_lines(*args, pipeline_slot=(slot, self.depth), **kw)))
        return self._join(self.inflight.pop(0)) if len(self.inflight) >= self.depth else None

    def drain(self):
        """joins and returns every batch still in flight, oldest first."""
        out = [self._join(e) for e in self.inflight]
        self.inflight = []
        return out

    def __del__(self):
        # a pipeline dropped with batches in flight: their output tensors go back to torch's caching allocator, which orders re-use on
        # the CURRENT stream only -- join them first, so that whoever gets those blocks next is queued behind the library's streams
        try:
            self.drain()
        except Exception:
            pass
