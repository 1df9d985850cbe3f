"""Multi-GPU plumbing: image pairs shard embarrassingly over ranks (pair p -> rank p mod R); the only
data-path collective is ONE all-gather of a fixed-size slab per rank that carries the line descriptors
together with everything global matching needs about them (per-image sub-line / key-line counts and the
sub-line -> key-line map), so that every rank holds the global descriptor set for any-vs-any matching
(BASELINE.json cfg4).

Slab layout, float32 [header_rows + map_rows + rows_cap, 256]:
    header (int32 bit-cast):  [n_images, n_0 .. n_{cap-1}, k_0 .. k_{cap-1}]      sub-line / key-line counts per image
    map    (int32 bit-cast):  sub2line[rows_cap]                                  key-line index of every sub-line
    rows:                     line_desc[N,256], zero-padded to rows_cap

torch.distributed backend "nccl" is RCCL on ROCm (xGMI inside a node); the same code runs on "gloo"
with CPU tensors, which is how the N>1 path is covered without GPUs (tests/test_distributed_cpu.py).
The reference has no counterpart (it only uses nn.DataParallel for training, train.py:81).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

D = 256


def shard_pairs(n_pairs: int, rank: int, world: int):
    """Round-robin pair -> rank map used by bench.py / cfg4."""
    return list(range(rank, n_pairs, world))


def owner_of(pair: int, world: int):
    """(rank, local index) of a global pair under shard_pairs."""
    return pair % world, pair // world


def header_rows(n_images_cap: int) -> int:
    """Rows of the slab reserved for the int32 header (image count + per-image sub-line and key-line counts)."""
    return (1 + 2 * n_images_cap + D - 1) // D


def map_rows(rows_cap: int) -> int:
    """Rows of the slab that hold the sub-line -> key-line map."""
    return (rows_cap + D - 1) // D


def slab_rows(n_images_cap: int, rows_cap: int) -> int:
    return header_rows(n_images_cap) + map_rows(rows_cap) + rows_cap


def _as_i32(x, device):
    if torch.is_tensor(x):
        return x.to(device=device, dtype=torch.int32)
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(device)


def pack_descriptors(line_desc: torch.Tensor, cu_n, n_images_cap: int, rows_cap: int, out: torch.Tensor | None = None,
                     cu_k=None, sub2line: torch.Tensor | None = None, d_cu_n: torch.Tensor | None = None,
                     d_cu_k: torch.Tensor | None = None) -> torch.Tensor:
    """[N,256] descriptors + per-image counts (+ key-line counts and the sub-line map) -> one fixed-size slab.

    cu_n / cu_k are the host prefix sums [B+1]; when the same arrays already live on the device (d_cu_n / d_cu_k, as
    Engine.describe leaves them) the header is assembled there and nothing is copied from the host."""
    n_img = len(cu_n) - 1
    N = int(cu_n[-1])
    if n_img > n_images_cap or N > rows_cap:
        raise ValueError(f"capacity exceeded: {n_img}>{n_images_cap} images or {N}>{rows_cap} rows")
    hr, mr = header_rows(n_images_cap), map_rows(rows_cap)
    dev = line_desc.device
    if out is None:
        out = torch.zeros((hr + mr + rows_cap, D), dtype=torch.float32, device=dev)
    if dev.type == "cuda" and d_cu_n is not None and (cu_k is None or d_cu_k is not None) and out.is_contiguous():
        # ONE launch (linetr_pack_slab): header from the device prefix sums, sub-line map and descriptor rows
        import ctypes as C
        from . import _native as nat
        ld = line_desc if (line_desc.dtype == torch.float32 and line_desc.is_contiguous()) else line_desc.float().contiguous()
        s2l = sub2line.to(torch.int32).contiguous() if sub2line is not None else None
        # the kernel reads both prefix sums as contiguous int32 on `dev`: any other integer tensor is converted, never reinterpreted
        i32 = lambda t: t if (t.dtype == torch.int32 and t.device == dev and t.is_contiguous()) else t.to(device=dev, dtype=torch.int32).contiguous()
        d_cu_n = i32(d_cu_n)
        d_cu_k = i32(d_cu_k) if d_cu_k is not None else None
        if d_cu_n.numel() < n_img + 1 or (d_cu_k is not None and d_cu_k.numel() < n_img + 1):
            raise ValueError("pack_descriptors: device prefix sums shorter than n_images + 1")
        with torch.cuda.device(dev):
            nat.check(nat.lib().linetr_pack_slab(ld.data_ptr(), N, d_cu_n.data_ptr(), d_cu_k.data_ptr() if d_cu_k is not None else None,
                                                 n_img, s2l.data_ptr() if s2l is not None else None, n_images_cap, rows_cap, 0,
                                                 out.data_ptr(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return out
    hdr = out[:hr].view(torch.int32).view(-1)
    if d_cu_n is not None and (cu_k is None or d_cu_k is not None):
        hdr[0:1].fill_(n_img)
        hdr[1:1 + n_img] = d_cu_n[1:n_img + 1] - d_cu_n[:n_img]
        if d_cu_k is not None:
            hdr[1 + n_images_cap:1 + n_images_cap + n_img] = d_cu_k[1:n_img + 1] - d_cu_k[:n_img]
    else:
        h = np.zeros(1 + 2 * n_images_cap, dtype=np.int32)
        h[0] = n_img
        h[1:1 + n_img] = np.diff(cu_n)
        if cu_k is not None:
            h[1 + n_images_cap:1 + n_images_cap + n_img] = np.diff(cu_k)
        src = torch.from_numpy(h)
        if dev.type == "cuda":
            src = src.pin_memory()
        hdr[:h.size].copy_(src, non_blocking=True)
    if sub2line is not None:
        out[hr:hr + mr].view(torch.int32).view(-1)[:N].copy_(sub2line[:N])
    out[hr + mr:hr + mr + N].copy_(line_desc[:N])
    return out


def unpack_descriptors(buf: torch.Tensor, n_images_cap: int, rows_cap: int | None = None, with_lines: bool = False):
    """Inverse of pack_descriptors for one rank's slab: (line_desc [N,256] view, cu_n) or, with_lines=True,
    (line_desc, cu_n, sub2line [N] int32 view, cu_k).  One small D2H copy of the header (synchronises)."""
    hr = header_rows(n_images_cap)
    if rows_cap is None:                       # infer from the slab height:  rows = hr + ceil(cap/256) + cap
        rest = buf.shape[0] - hr
        rows_cap = rest - (rest + D) // (D + 1)
        while map_rows(rows_cap) + rows_cap < rest:
            rows_cap += 1
    mr = map_rows(rows_cap)
    hdr = buf[:hr].view(torch.int32).view(-1)[:1 + 2 * n_images_cap].cpu().numpy()
    n_img = int(hdr[0])
    cu = np.zeros(n_img + 1, dtype=np.int32)
    np.cumsum(hdr[1:1 + n_img], out=cu[1:])
    desc = buf[hr + mr:hr + mr + int(cu[-1])]
    if not with_lines:
        return desc, cu
    cu_k = np.zeros(n_img + 1, dtype=np.int32)
    np.cumsum(hdr[1 + n_images_cap:1 + n_images_cap + n_img], out=cu_k[1:])
    s2l = buf[hr:hr + mr].view(torch.int32).view(-1)[:int(cu[-1])]
    return desc, cu, s2l, cu_k


def allgather_descriptors(packed: torch.Tensor, group=None, async_op: bool = False):
    """ONE collective: every rank contributes its packed slab, receives [world, rows, 256].
    async_op=True returns (work, out): the caller overlaps the transfer with the next batch's compute and calls
    work.wait() before touching `out` or re-using `packed`."""
    world = dist.get_world_size(group)
    rows = packed.shape[0]
    out = torch.empty((world * rows,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    work = dist.all_gather_into_tensor(out, packed.contiguous(), group=group, async_op=async_op)  # dim-0 concatenation
    out = out.view((world, rows) + tuple(packed.shape[1:]))
    return (work, out) if async_op else out


class NativeAllGather:
    """The descriptor all-gather through the library's own C entry point: linetr_allgather_desc (ncclAllGather of the packed slabs,
    include/linetr_hip.h) over an RCCL communicator that spans the ranks of the initialised torch.distributed group and is created
    directly on the librccl PyTorch has loaded (ncclGetUniqueId on rank 0, the 128-byte id handed round with
    dist.broadcast_object_list, ncclCommInitRank on every rank).  Same result as allgather_descriptors; exists so that the C ABI's
    collective is exercised by the same command as the torch one (bench.py, LINETR_BENCH_COLLECTIVE=native).  One rank per DEVICE:
    RCCL refuses two ranks on one GPU, so the single-device gloo harness cannot use it."""

    def __init__(self, device, group=None):
        import ctypes as C
        import os

        from . import _native as nat
        self._nat, self._C = nat, C
        self.device = torch.device(device)
        inited = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if inited else 0
        self.world = dist.get_world_size(group) if inited else 1
        self._rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        self._rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        self._rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        self._rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        uid = UniqueId()
        if self.rank == 0 and self._rccl.ncclGetUniqueId(C.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        box = [bytes(C.string_at(C.addressof(uid), 128))]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
            C.memmove(C.addressof(uid), box[0], 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self._rccl.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank failed on rank {self.rank} (code {rc})")

    def __call__(self, packed: torch.Tensor) -> torch.Tensor:
        """[rows,256] slab of this rank -> [world, rows, 256]; enqueued on the current stream of the device."""
        packed = packed.contiguous()
        out = torch.empty((self.world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
        with torch.cuda.device(self.device):
            st = self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._nat.check(self._nat.lib().linetr_allgather_desc(self._comm, packed.data_ptr(), out.data_ptr(),
                                                                   packed.numel() * packed.element_size(), st))
        return out

    def close(self):
        if getattr(self, "_comm", None):
            self._rccl.ncclCommDestroy(self._comm)
            self._comm = None


class GatheredSet:
    """The global descriptor set after the all-gather: per-rank views + host tables, images addressed globally.

    Slab r holds the images of rank r in local order; with pairs sharded round-robin, the two images of global pair p
    are local images 2*(p // R) and 2*(p // R) + 1 of rank p % R."""

    def __init__(self, gathered: torch.Tensor, n_images_cap: int, rows_cap: int):
        self.world = int(gathered.shape[0])
        self.flat = gathered.reshape(-1, D)
        self.slab = int(gathered.shape[1])
        self.hr, self.mr = header_rows(n_images_cap), map_rows(rows_cap)
        hdr = gathered[:, :self.hr].reshape(self.world, -1).view(torch.int32)[:, :1 + 2 * n_images_cap].cpu().numpy()
        self.cu_n, self.cu_k = [], []
        for r in range(self.world):
            n_img = int(hdr[r, 0])
            self.cu_n.append(np.concatenate([[0], np.cumsum(hdr[r, 1:1 + n_img])]).astype(np.int64))
            self.cu_k.append(np.concatenate([[0], np.cumsum(hdr[r, 1 + n_images_cap:1 + n_images_cap + n_img])]).astype(np.int64))
        # one int32 view of the whole buffer: the sub-line maps are addressed with absolute element offsets
        self.flat_i32 = self.flat.view(torch.int32).view(-1)

    def image(self, rank: int, local_image: int):
        """(row offset into self.flat, n, element offset of the image's sub2line in flat_i32, k)."""
        cn, ck = self.cu_n[rank], self.cu_k[rank]
        n0 = int(cn[local_image])
        row = rank * self.slab + self.hr + self.mr + n0
        s2l = (rank * self.slab + self.hr) * D + n0
        return row, int(cn[local_image + 1]) - n0, s2l, int(ck[local_image + 1] - ck[local_image])

    def pair_image(self, pair: int, side: int):
        r, loc = owner_of(pair, self.world)
        return self.image(r, 2 * loc + side)


def global_match(eng, gs: GatheredSet, queries, candidates, thr: float, mutual: bool = True):
    """Match image `queries[i]` against image `candidates[i]` for all i in ONE linetr_match call, every image addressed
    inside the gathered buffer (so any rank's descriptors can be on either side).
    queries / candidates: lists of (rank, local_image).  Returns (dk_flat, off_dk, match01, off_k0) as Engine.match."""
    P = len(queries)
    dims = np.zeros((P, 4), dtype=np.int32)
    off0 = np.zeros(P, dtype=np.int64); off1 = np.zeros(P, dtype=np.int64)
    s0 = np.zeros(P, dtype=np.int64); s1 = np.zeros(P, dtype=np.int64)
    for i, (q, c) in enumerate(zip(queries, candidates)):
        r0, n0, m0, k0 = gs.image(*q)
        r1, n1, m1, k1 = gs.image(*c)
        dims[i] = (n0, k0, n1, k1)
        off0[i], off1[i], s0[i], s1[i] = r0, r1, m0, m1
    return eng.match_offsets(gs.flat, gs.flat_i32, dims, off0, s0, off1, s1, thr, mutual)
