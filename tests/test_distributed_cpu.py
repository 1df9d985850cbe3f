"""CPU, world_size=2, gloo: the N>1 path of bench.py (pair sharding + the single all-gather of padded
line descriptors with counts in the same buffer) round-trips uneven per-image sub-line counts."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from linetr_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_payload(rank, n_pairs_total, world):
    """deterministic fake descriptors for the pairs a rank owns (uneven image sizes)."""
    mine = parallel.shard_pairs(n_pairs_total, rank, world)
    sizes, chunks = [], []
    for p in mine:
        for side in (0, 1):
            n = 3 + (7 * p + 5 * side) % 11
            rs = np.random.RandomState(1000 * p + side)
            chunks.append(rs.standard_normal((n, 256)).astype(np.float32))
            sizes.append(n)
    cu = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ld = torch.from_numpy(np.concatenate(chunks)) if chunks else torch.zeros((0, 256))
    return mine, ld, cu


def _worker(rank, world, port, n_pairs_total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine, ld, cu = _rank_payload(rank, n_pairs_total, world)
        img_cap, rows_cap = 2 * ((n_pairs_total + world - 1) // world), 2 * 14 * ((n_pairs_total + world - 1) // world)
        packed = parallel.pack_descriptors(ld, cu, img_cap, rows_cap)
        allbuf = parallel.allgather_descriptors(packed)
        assert allbuf.shape == (world, parallel.header_rows(img_cap) + rows_cap, 256)
        ok = True
        for r in range(world):
            d, c = parallel.unpack_descriptors(allbuf[r], img_cap)
            _, want_ld, want_cu = _rank_payload(r, n_pairs_total, world)
            ok &= np.array_equal(c, want_cu) and torch.equal(d, want_ld)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [5, 8])
def test_allgather_roundtrip_world2(n_pairs):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_pairs, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_pairs_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(p for r in range(world) for p in parallel.shard_pairs(1024, r, world))
        assert seen == list(range(1024))
        assert max(len(parallel.shard_pairs(1024, r, world)) for r in range(world)) == 1024 // world


def test_pack_capacity_errors():
    with pytest.raises(ValueError):
        parallel.pack_descriptors(torch.zeros((10, 256)), np.array([0, 10], np.int32), 1, 5)
