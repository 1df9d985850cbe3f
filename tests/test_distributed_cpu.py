"""CPU, gloo, world_size 2 and 4: the N>1 path of bench.py -- pair sharding, the single all-gather of one slab per rank
(descriptors + per-image counts + key-line maps) and global matching on the gathered set -- without GPUs.

The world-4 test replays a frozen cfg4 job (tests/golden/cfg4_job.npz: the CPU oracle's descriptors of 8 homography
pairs) as the engine output of 4 ranks and checks every rank's global matches, computed from the GATHERED buffer with
the oracle's matcher, against the answers frozen from a single process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from linetr_amd import parallel

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg4_job.npz")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_payload(rank, n_pairs_total, world):
    """deterministic fake descriptors for the pairs a rank owns (uneven image sizes)."""
    mine = parallel.shard_pairs(n_pairs_total, rank, world)
    sizes, ks, chunks, maps = [], [], [], []
    for p in mine:
        for side in (0, 1):
            n = 3 + (7 * p + 5 * side) % 11
            k = max(1, n - (p + side) % 3)
            rs = np.random.RandomState(1000 * p + side)
            chunks.append(rs.standard_normal((n, 256)).astype(np.float32))
            maps.append(np.minimum(np.arange(n), k - 1).astype(np.int32))
            sizes.append(n)
            ks.append(k)
    cu = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    cu_k = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    ld = torch.from_numpy(np.concatenate(chunks)) if chunks else torch.zeros((0, 256))
    s2l = torch.from_numpy(np.concatenate(maps)) if maps else torch.zeros((0,), dtype=torch.int32)
    return mine, ld, cu, cu_k, s2l


def _worker(rank, world, port, n_pairs_total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)            # several ranks share this host's few cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine, ld, cu, cu_k, s2l = _rank_payload(rank, n_pairs_total, world)
        per = (n_pairs_total + world - 1) // world
        img_cap, rows_cap = 2 * per, 2 * 14 * per
        packed = parallel.pack_descriptors(ld, cu, img_cap, rows_cap, cu_k=cu_k, sub2line=s2l)
        allbuf = parallel.allgather_descriptors(packed)
        assert allbuf.shape == (world, parallel.slab_rows(img_cap, rows_cap), 256)
        ok = True
        for r in range(world):
            d, c, m, ck = parallel.unpack_descriptors(allbuf[r], img_cap, rows_cap, with_lines=True)
            _, want_ld, want_cu, want_ck, want_m = _rank_payload(r, n_pairs_total, world)
            ok &= np.array_equal(c, want_cu) and torch.equal(d, want_ld) and np.array_equal(ck, want_ck)
            ok &= torch.equal(m, want_m)
            d2, c2 = parallel.unpack_descriptors(allbuf[r], img_cap)          # rows_cap inferred from the slab height
            ok &= np.array_equal(c2, want_cu) and torch.equal(d2, want_ld)
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


def _sparse_worker(rank, world, port, n_pairs_total, ret):
    """fewer pairs than ranks: some ranks contribute an EMPTY slab; the slab height is agreed with one MAX all-reduce of the
    ranks' (very uneven) local row counts, as bench.py's cfg4 job does."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine, ld, cu, cu_k, s2l = _rank_payload(rank, n_pairs_total, world)
        img_cap = 2 * max(1, (n_pairs_total + world - 1) // world)
        rows_cap = torch.tensor([int(cu[-1])])
        dist.all_reduce(rows_cap, op=dist.ReduceOp.MAX)
        rows_cap = max(int(rows_cap.item()), 1)
        slab = parallel.pack_descriptors(ld, cu, img_cap, rows_cap, cu_k=cu_k, sub2line=s2l)
        gs = parallel.GatheredSet(parallel.allgather_descriptors(slab), img_cap, rows_cap)
        ok = True
        for r in range(world):
            r_mine, want_ld, want_cu, want_ck, want_m = _rank_payload(r, n_pairs_total, world)
            ok &= np.array_equal(gs.cu_n[r], want_cu) and np.array_equal(gs.cu_k[r], want_ck)
            ok &= (len(r_mine) == 0) == (len(gs.cu_n[r]) == 1)
            for li in range(len(want_cu) - 1):
                row, n, m, k = gs.image(r, li)
                ok &= torch.equal(gs.flat[row:row + n], want_ld[want_cu[li]:want_cu[li + 1]])
                ok &= torch.equal(gs.flat_i32[m:m + n], want_m[want_cu[li]:want_cu[li + 1]])
                ok &= k == int(want_ck[li + 1] - want_ck[li])
        for p in range(n_pairs_total):          # every pair is addressable from every rank, whoever owns it
            row, n, _, _ = gs.pair_image(p, 1)
            ok &= n == 3 + (7 * p + 5) % 11
        ret[rank] = (bool(ok), len(mine))
    finally:
        dist.destroy_process_group()


def test_rank_without_pairs_and_uneven_slabs_world4():
    world, n_pairs = 4, 3                        # rank 3 owns nothing
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sparse_worker, args=(world, _free_port(), n_pairs, ret), nprocs=world, join=True)
    assert dict(ret) == {0: (True, 1), 1: (True, 1), 2: (True, 1), 3: (True, 0)}


def _oracle_match(desc0, s2l0, k0, desc1, s2l1, k1, thr=0.8):
    """the oracle's matcher on [n,256] descriptors + key-line maps (what linetr_match_gathered computes on the GPU)."""
    from oracle import linetr_oracle as O

    def mat(s2l, k):
        n = len(s2l)
        a = torch.zeros((k, n))
        cnt = torch.bincount(s2l.long(), minlength=k).clamp(min=1)
        a[s2l.long(), torch.arange(n)] = (1.0 / cnt.double())[s2l.long()].float()
        return a
    M, Dk = O.match_lines(desc0.t()[None], desc1.t()[None], mat(s2l0, k0), mat(s2l1, k1), thr)
    return np.where(M[0].sum(1) > 0, M[0].argmax(1), -1).astype(np.int32), Dk[0]


def _cfg4_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)            # several ranks share this host's few cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(GOLD)
        P, S = int(g["P"]), int(g["S"])
        mine = parallel.shard_pairs(P, rank, world)
        imgs = [2 * p + side for p in mine for side in (0, 1)]
        ld = torch.from_numpy(np.concatenate([g[f"desc_{i}"] for i in imgs]))
        s2l = torch.from_numpy(np.concatenate([g[f"s2l_{i}"] for i in imgs]))
        cu_n = np.concatenate([[0], np.cumsum([len(g[f"s2l_{i}"]) for i in imgs])]).astype(np.int32)
        cu_k = np.concatenate([[0], np.cumsum([int(g[f"k_{i}"]) for i in imgs])]).astype(np.int32)
        per = (P + world - 1) // world
        img_cap = 2 * per
        rows_cap = torch.tensor([int(cu_n[-1])])
        dist.all_reduce(rows_cap, op=dist.ReduceOp.MAX)               # same slab height on every rank (bench.Cfg4Job)
        rows_cap = int(rows_cap.item())
        slab = parallel.pack_descriptors(ld, cu_n, img_cap, rows_cap, cu_k=cu_k, sub2line=s2l)
        gathered = parallel.allgather_descriptors(slab)               # THE single collective
        gs = parallel.GatheredSet(gathered, img_cap, rows_cap)
        ok, checked = True, 0
        for p in mine:
            for s in range(S):
                c = (p + s) % P
                r0, n0, m0, k0 = gs.pair_image(p, 0)
                r1, n1, m1, k1 = gs.pair_image(c, 1)
                assert parallel.owner_of(p, world)[0] == rank
                d0, d1 = gs.flat[r0:r0 + n0], gs.flat[r1:r1 + n1]
                a0, a1 = gs.flat_i32[m0:m0 + n0], gs.flat_i32[m1:m1 + n1]
                ok &= torch.equal(d1, torch.from_numpy(g[f"desc_{2 * c + 1}"])) and k1 == int(g[f"k_{2 * c + 1}"])
                m01, dk = _oracle_match(d0, a0, k0, d1, a1, k1)
                ok &= np.array_equal(m01, g[f"match_{p}_{s}"]) and np.abs(dk - g[f"dk_{p}_{s}"]).max() < 1e-5
                checked += 1
        ret[rank] = (bool(ok), checked)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_cfg4_global_matching_on_gathered_set(world):
    """8 pairs round-robin over the ranks, one all-gather, every rank matches its queries against candidates that other
    ranks own: all 8 x 3 pair-matches equal the single-process answers."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cfg4_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(v[0] for v in ret.values()) and sum(v[1] for v in ret.values()) == 24, dict(ret)


def test_cfg4_fixture_recall_against_homography():
    """the frozen job itself: own-partner matches (s = 0) recover the correspondences the known homography defines."""
    from workloads import synth
    g = np.load(GOLD)
    hit = tot = 0
    for p in range(int(g["P"])):
        k0, k1, m = g[f"klines_{2 * p}"].astype(np.float64), g[f"klines_{2 * p + 1}"].astype(np.float64), g[f"homography_{p}"]
        w = synth.warp_points(m, k0.reshape(-1, 2)).reshape(-1, 2, 2)
        m01 = g[f"match_{p}_0"]
        for i in range(len(k0)):
            d = np.minimum(np.abs(k1 - w[i][None]).reshape(len(k1), -1).max(1), np.abs(k1 - w[i][::-1][None]).reshape(len(k1), -1).max(1))
            j = int(d.argmin())
            if d[j] < 0.5:
                tot += 1
                hit += int(m01[i] == j)
    assert tot > 150 and hit / tot > 0.8, (hit, tot)


def test_shard_pairs_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(p for r in range(world) for p in parallel.shard_pairs(1024, r, world))
        assert seen == list(range(1024))
        assert max(len(parallel.shard_pairs(1024, r, world)) for r in range(world)) == 1024 // world
        for p in (0, 1, 7, 8, 1023):
            r, loc = parallel.owner_of(p, world)
            assert parallel.shard_pairs(1024, r, world)[loc] == p


def test_pack_capacity_errors():
    with pytest.raises(ValueError):
        parallel.pack_descriptors(torch.zeros((10, 256)), np.array([0, 10], np.int32), 1, 5)


def test_homography_sampler_properties():
    """cfg4 generator: deterministic per seed, warped end points of the common lines land on image-1 rows, every line
    inside the margin box, strength -> 0 tends to the identity."""
    from workloads import synth
    a = synth.homography_pair(7, 60)
    b = synth.homography_pair(7, 60)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    l0, l1, m, gt = a
    ok = gt >= 0
    assert ok.sum() == 51 and l0.shape == l1.shape == (60, 6)
    w = synth.warp_points(m, l0[ok][:, :4].reshape(-1, 2, 2)).reshape(-1, 4)
    assert np.abs(w - l1[gt[ok]][:, :4]).max() < 1e-3
    for l in (l0, l1):
        assert (l[:, [0, 2]] >= 10).all() and (l[:, [0, 2]] <= 630).all() and (l[:, [1, 3]] >= 10).all() and (l[:, [1, 3]] <= 470).all()
        assert np.array_equal(l[:, :5], l[:, :5].astype(np.float32).astype(np.float64))
    rs = np.random.RandomState(3)
    near = synth.pixel_homography(rs, 480, 640, strength=1e-4)
    assert np.abs(near - np.eye(3)).max() < 0.1
    full = synth.pixel_homography(np.random.RandomState(3), 480, 640)
    assert np.abs(full - np.eye(3)).max() > 0.01


def test_warped_dense_maps_follow_the_homography():
    """cfg4 generator: the dense maps of view 1 are view 0's maps seen through M (x1 ~ M x0).  At random interior points
    the descriptor sampled from view 1 at M x0 has cosine > 0.9 with view 0's descriptor at x0 (bilinear resampling +
    5 % noise), and the score maps agree the same way -- i.e. the warp goes in the direction the line end points do."""
    import torch.nn.functional as F
    from workloads import synth
    H, W = 240, 320
    rs = np.random.RandomState(5)
    m = synth.pixel_homography(rs, H, W, strength=0.3)
    g = torch.Generator().manual_seed(3)
    # smooth maps (upsampled low-resolution noise) so that bilinear resampling is meaningful
    dd0 = F.normalize(F.interpolate(torch.randn(1, 256, H // 32, W // 32, generator=g), size=(H // 8, W // 8), mode="bilinear",
                                    align_corners=False), dim=1)
    ds0 = F.interpolate(torch.rand(1, 1, H // 16, W // 16, generator=g), size=(H, W), mode="bilinear", align_corners=False)[0]
    dd1, ds1 = synth.warp_dense_maps(dd0, ds0, m, noise=0.05, seed=1)
    assert dd1.shape == dd0.shape and ds1.shape == ds0.shape
    assert ((dd1.norm(dim=1) - 1).abs() < 1e-5).all()
    pts0 = np.stack([rs.uniform(40, W - 40, 200), rs.uniform(40, H - 40, 200)], axis=1)
    pts1 = synth.warp_points(m, pts0)
    ok = (pts1[:, 0] > 16) & (pts1[:, 0] < W - 16) & (pts1[:, 1] > 16) & (pts1[:, 1] < H - 16)
    assert ok.sum() > 50

    def sample_desc(dd, pts):          # nearest cell centre of the 1/8 map
        cx = np.clip(np.round((pts[:, 0] - 3.5) / 8).astype(int), 0, W // 8 - 1)
        cy = np.clip(np.round((pts[:, 1] - 3.5) / 8).astype(int), 0, H // 8 - 1)
        return dd[0][:, cy, cx].t()
    cos = (sample_desc(dd0, pts0[ok]) * sample_desc(dd1, pts1[ok])).sum(1)
    assert cos.median().item() > 0.9 and (cos > 0.7).float().mean().item() > 0.9
    s0 = ds0[0][np.round(pts0[ok][:, 1]).astype(int), np.round(pts0[ok][:, 0]).astype(int)]
    s1 = ds1[0][np.round(pts1[ok][:, 1]).astype(int), np.round(pts1[ok][:, 0]).astype(int)]
    assert (s0 - s1).abs().median().item() < 0.05
