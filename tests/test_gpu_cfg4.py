"""GPU: BASELINE.json cfg4 -- homography-augmented pairs, pair sharding, one slab per rank, global matching on the
gathered set -- on one device (world = 1 runs the same code path as N ranks minus the collective itself, which the
gloo tests cover), against the frozen job of tests/golden/cfg4_job.npz and the known homographies."""
import os
import sys

import numpy as np
import pytest
import torch

from linetr_amd import parallel
from workloads import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(ROOT, "tests", "golden", "cfg4_job.npz")
KW = dict(remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21)


@pytest.fixture(scope="module")
def eng():
    from linetr_amd.engine import Engine
    return Engine(synth.calibrated_state_dict(), "cuda:0")


def _describe_fixture_images(eng, g, images):
    """HIP descriptors of the fixture's images (same lines, same dense maps as make_cfg4_fixture.py)."""
    H, W = (int(v) for v in g["hw"])
    lines, dds, dss = [], [], []
    cache = {}
    for i in images:
        p, side = divmod(i, 2)
        if p not in cache:
            dd0, ds0 = synth.synth_dense_maps_np(int(g["seed_base"]) + p, H, W)
            dd0, ds0 = torch.from_numpy(dd0), torch.from_numpy(ds0)
            cache[p] = ((dd0, ds0), synth.warp_dense_maps(dd0, ds0, g[f"homography_{p}"], seed=p))
        dd, ds = cache[p][side]
        lines.append(g[f"lines_{i}"]); dds.append(dd); dss.append(ds)
    off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)
    tb, ld = eng.describe_lines(np.concatenate(lines), off, torch.cat(dds).cuda(), torch.cat(dss).cuda(), **KW)
    return tb, ld


def test_fixture_job_as_four_ranks_on_one_device(eng):
    """The frozen 8-pair job computed by the HIP path, laid out exactly as 4 ranks would lay it out (round-robin shards,
    one slab each, slabs stacked as the all-gather stacks them), then linetr_match_gathered over the stacked buffer:
    descriptors within 1e-4 of the oracle's, all 24 global pair-matches identical to the frozen answers."""
    g = np.load(GOLD)
    P, S, world = int(g["P"]), int(g["S"]), 4
    per = P // world
    slabs, rows_caps, parts = [], [], []
    for r in range(world):
        mine = parallel.shard_pairs(P, r, world)
        tb, ld = _describe_fixture_images(eng, g, [2 * p + s for p in mine for s in (0, 1)])
        for j, i in enumerate(2 * p + s for p in mine for s in (0, 1)):
            have = ld[tb.cu_n[j]:tb.cu_n[j + 1]].cpu().numpy()
            assert have.shape == g[f"desc_{i}"].shape and np.abs(have - g[f"desc_{i}"]).max() < 1e-4
            assert np.array_equal(tb.sub2line[tb.cu_n[j]:tb.cu_n[j + 1]].cpu().numpy(), g[f"s2l_{i}"])
        parts.append((tb, ld))
        rows_caps.append(tb.N)
    rows_cap = max(rows_caps)
    for tb, ld in parts:
        slabs.append(parallel.pack_descriptors(ld, tb.cu_n, 2 * per, rows_cap, cu_k=tb.cu_k, sub2line=tb.sub2line,
                                               d_cu_n=tb.extra.get("d_cu_n"), d_cu_k=tb.extra.get("d_cu_k")))
    gathered = torch.stack(slabs)                      # what all_gather_into_tensor hands every rank
    gs = parallel.GatheredSet(gathered, 2 * per, rows_cap)
    for r in range(world):                             # every "rank" matches its own queries
        q, c, keys = [], [], []
        for p in parallel.shard_pairs(P, r, world):
            for s in range(S):
                cand = (p + s) % P
                q.append((r, 2 * (p // world)))
                rc, lc = parallel.owner_of(cand, world)
                c.append((rc, 2 * lc + 1))
                keys.append((p, s))
        dk, off_dk, m01, off_k0 = parallel.global_match(eng, gs, q, c, 0.8, True)
        m01, dk = m01.cpu().numpy(), dk.cpu().numpy()
        for i, (p, s) in enumerate(keys):
            assert np.array_equal(m01[off_k0[i]:off_k0[i + 1]], g[f"match_{p}_{s}"]), (p, s)
            want = g[f"dk_{p}_{s}"]
            assert np.abs(dk[off_dk[i]:off_dk[i + 1]].reshape(want.shape) - want).max() < 1e-4


def test_cfg4_job_world1_recall_and_phases(eng):
    """bench.py's Cfg4Job itself (world = 1): 12 mild-view pairs in batches of 5, 3 gathered candidates per query.
    Own-partner matches recover > 80 % of the correspondences the known homography defines; an unrelated candidate
    recovers next to none; the three phases are timed."""
    sys.path.insert(0, ROOT)
    import bench
    job = bench.Cfg4Job(eng, torch.device("cuda:0"), 0, 1, pairs_total=12, batch_pairs=5, candidates=3, strength=0.05)
    assert len(job.batches) == 3
    n = job.step()
    comp, gath, mat = job.phase_ms()
    assert n > 12 * 2 * 150 and comp > 0 and mat > 0 and gath >= 0
    rec = job.recall(max_pairs=12)
    assert rec["gt_pairs"] > 12 * 120 and rec["recall"] > 0.8, rec
    # candidate s = 1 is another scene: mutual-NN matches exist (threshold 0.8) but do not follow pair p's homography
    gs, ld, cu_n, cu_k, res, queries, cands = job.last
    assert len(queries) == 36 and queries[1][1] == queries[0][1] and cands[1] == (0, 3)
    n2 = job.step()
    assert n2 == n


def test_matcher_async_and_multiblock_equals_oracle(eng):
    """linetr_match (pinned staging ring, pooling + final kernels, no stream synchronisation inside) on ragged pairs --
    key-lines with 1..4 sub-lines, k not a multiple of the 8-row chunk, one empty side -- against the oracle, called
    back to back so that ring slots are reused while earlier launches are still in flight."""
    from oracle import linetr_oracle as O
    rs = np.random.RandomState(5)
    cases = []
    for (k0, k1) in [(1, 1), (7, 9), (8, 8), (41, 23), (130, 257)]:
        def side(k):
            reps = rs.randint(1, 5, size=k)
            s2l = np.repeat(np.arange(k), reps).astype(np.int32)
            d = rs.standard_normal((len(s2l), 256)).astype(np.float32)
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            return d, s2l
        cases.append((side(k0), side(k1), k0, k1))
    outs = []
    for _ in range(3):                      # 15 launches without a host wait in between: > 8 ring slots
        for (d0, s0), (d1, s1), k0, k1 in cases:
            t0, t1 = torch.from_numpy(d0).cuda(), torch.from_numpy(d1).cuda()
            outs.append(eng.match(t0, np.array([0, len(s0)]), torch.from_numpy(s0).cuda(), np.array([0, k0]), t1,
                                  np.array([0, len(s1)]), torch.from_numpy(s1).cuda(), np.array([0, k1]), 0.8, True))
    torch.cuda.synchronize()
    for i, (dk, off, m01) in enumerate(outs):
        (d0, s0), (d1, s1), k0, k1 = cases[i % len(cases)]

        def mat(s2l, k):
            a = torch.zeros((k, len(s2l)))
            cnt = torch.bincount(torch.from_numpy(s2l).long(), minlength=k)
            a[torch.from_numpy(s2l).long(), torch.arange(len(s2l))] = (1.0 / cnt.double())[torch.from_numpy(s2l).long()].float()
            return a
        M, Dk = O.match_lines(torch.from_numpy(d0).t()[None], torch.from_numpy(d1).t()[None], mat(s0, k0), mat(s1, k1), 0.8)
        want = np.where(M[0].sum(1) > 0, M[0].argmax(1), -1)
        assert np.array_equal(m01.cpu().numpy(), want), i
        assert np.abs(dk.cpu().numpy().reshape(k0, k1) - Dk[0]).max() < 1e-5


def test_match_empty_side(eng):
    d = torch.nn.functional.normalize(torch.randn(5, 256, device="cuda"), dim=1)
    s = torch.arange(5, dtype=torch.int32, device="cuda")
    dk, off, m01 = eng.match(d, np.array([0, 5]), s, np.array([0, 5]), d[:0], np.array([0, 0]), s[:0], np.array([0, 0]), 0.8, True)
    assert (m01.cpu().numpy() == -1).all() and dk.numel() == 0


def test_rccl_single_rank_allgather_and_global_match(eng):
    """The collective itself on the device: torch.distributed backend 'nccl' (= RCCL on ROCm) with ONE rank -- all this box
    has -- all-gathers the slab of a small batch and the global matcher reads the gathered buffer.  (World sizes 2 and 4
    are covered on CPU with gloo in tests/test_distributed_cpu.py; a real multi-GPU run has never been possible here.)"""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        lines = [synth.synth_lines(900 + i, 50, 480, 640) for i in range(4)]
        maps = [synth.synth_dense_maps(900 + i, 480, 640) for i in range(4)]
        off = np.concatenate([[0], np.cumsum([len(l) for l in lines])]).astype(np.int32)
        tb, ld = eng.describe_lines(np.concatenate(lines), off, torch.cat([m[0] for m in maps]).cuda(),
                                    torch.cat([m[1] for m in maps]).cuda(), **KW)
        slab = parallel.pack_descriptors(ld, tb.cu_n, 4, tb.N, cu_k=tb.cu_k, sub2line=tb.sub2line,
                                         d_cu_n=tb.extra.get("d_cu_n"), d_cu_k=tb.extra.get("d_cu_k"))
        work, gathered = parallel.allgather_descriptors(slab, async_op=True)
        work.wait()
        assert gathered.shape == (1, parallel.slab_rows(4, tb.N), 256) and torch.equal(gathered[0], slab)
        gs = parallel.GatheredSet(gathered, 4, tb.N)
        dk, off_dk, m01, off_k0 = parallel.global_match(eng, gs, [(0, 0), (0, 2)], [(0, 1), (0, 3)], 0.8, True)
        n, k = np.diff(tb.cu_n), np.diff(tb.cu_k)
        ref = eng.match(torch.cat([ld[tb.cu_n[0]:tb.cu_n[1]], ld[tb.cu_n[2]:tb.cu_n[3]]]), np.array([0, n[0], n[0] + n[2]]),
                        torch.cat([tb.sub2line[tb.cu_n[0]:tb.cu_n[1]], tb.sub2line[tb.cu_n[2]:tb.cu_n[3]]]),
                        np.array([0, k[0], k[0] + k[2]]),
                        torch.cat([ld[tb.cu_n[1]:tb.cu_n[2]], ld[tb.cu_n[3]:tb.cu_n[4]]]), np.array([0, n[1], n[1] + n[3]]),
                        torch.cat([tb.sub2line[tb.cu_n[1]:tb.cu_n[2]], tb.sub2line[tb.cu_n[3]:tb.cu_n[4]]]),
                        np.array([0, k[1], k[1] + k[3]]), 0.8, True)
        assert torch.equal(m01, ref[2]) and torch.equal(dk, ref[0])
    finally:
        dist.destroy_process_group()


def test_native_allgather_entry_point_single_rank():
    """linetr_allgather_desc (the C-ABI form of the descriptor all-gather) over a one-rank RCCL communicator created directly
    on the RCCL PyTorch has loaded: the gathered buffer must equal the slab.  (More than one rank needs more than one GPU;
    the multi-rank layout is covered by the gloo tests and by linetr_match_gathered on a four-rank layout above.)"""
    import ctypes as C
    import os

    from linetr_amd import _native as nat
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(0)
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        slab = torch.randn(1000, 256, device="cuda")
        out = torch.zeros_like(slab)
        st = torch.cuda.current_stream().cuda_stream
        nat.check(nat.lib().linetr_allgather_desc(comm, slab.data_ptr(), out.data_ptr(), slab.numel() * 4, st))
        torch.cuda.synchronize()
        assert torch.equal(out, slab)
    finally:
        rccl.ncclCommDestroy(comm)


def test_native_allgather_wrapper_single_rank():
    """parallel.NativeAllGather (what bench.py runs with LINETR_BENCH_COLLECTIVE=native): the RCCL communicator is created on the
    RCCL PyTorch loaded, the slab goes through linetr_allgather_desc, the result has torch's all-gather layout [world, rows, 256]."""
    nag = parallel.NativeAllGather(torch.device("cuda:0"))
    try:
        slab = torch.randn(777, 256, device="cuda")
        out = nag(slab)
        torch.cuda.synchronize()
        assert nag.world == 1 and out.shape == (1, 777, 256) and torch.equal(out[0], slab)
    finally:
        nag.close()
