"""Multi-GPU plumbing: image pairs shard embarrassingly over ranks (pair p -> rank p mod R); the only
data-path collective is ONE all-gather of the padded line descriptors (+ counts in the same buffer)
so that every rank holds the global descriptor set for any-vs-any matching.

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


def header_rows(n_images_cap: int) -> int:
    """Rows of the [rows,256] float32 buffer reserved for the int32 header (count + per-image sizes)."""
    return (1 + n_images_cap + D - 1) // D


def pack_descriptors(line_desc: torch.Tensor, cu_n: np.ndarray, n_images_cap: int, rows_cap: int,
                     out: torch.Tensor | None = None) -> torch.Tensor:
    """[N,256] + per-image sub-line counts -> fixed-size buffer [header + rows_cap, 256].

    The header stores int32 values bit-cast into the float32 buffer: [n_images, n_0, n_1, ...]."""
    n_img = len(cu_n) - 1
    N = int(cu_n[-1])
    if n_img > n_images_cap or N > rows_cap:
        raise ValueError(f"capacity exceeded: {n_img}>{n_images_cap} images or {N}>{rows_cap} rows")
    hr = header_rows(n_images_cap)
    if out is None:
        out = torch.zeros((hr + rows_cap, D), dtype=torch.float32, device=line_desc.device)
    hdr = np.zeros(hr * D, dtype=np.int32)
    hdr[0] = n_img
    hdr[1:1 + n_img] = np.diff(cu_n)
    out[:hr].view(torch.int32).view(-1).copy_(torch.from_numpy(hdr), non_blocking=True)
    out[hr:hr + N].copy_(line_desc[:N])
    return out


def unpack_descriptors(buf: torch.Tensor, n_images_cap: int):
    """Inverse of pack_descriptors for one rank's slab: returns (line_desc [N,256] view, cu_n)."""
    hr = header_rows(n_images_cap)
    hdr = buf[:hr].view(torch.int32).view(-1)[:1 + n_images_cap].cpu().numpy()
    n_img = int(hdr[0])
    cu = np.zeros(n_img + 1, dtype=np.int32)
    np.cumsum(hdr[1:1 + n_img], out=cu[1:])
    return buf[hr:hr + int(cu[-1])], cu


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
