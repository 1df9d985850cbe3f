"""GPU: the one-launch matcher of a single pair (pair_match_fused_kernel: distances + key-line pooling + mutual NN, column argmin
by a 64-bit packed atomicMin, last-arriving block finishes) against the CPU oracle and against the three-launch path it replaces
(the same pair inside a batch of two): models/line_process.py:198-201, models/line_transformer.py:277-282, models/nn_matcher.py:3-31."""
import numpy as np
import pytest
import torch

from oracle import linetr_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def eng():
    from linetr_amd.engine import Engine
    return Engine.heads_only("cuda:0")


def make_side(rs, k, max_sub, dup=()):
    """k key-lines with 1..max_sub sub-lines each; returns (desc [n,256] f32, sub2line [n] i32, A [k,n] f32)."""
    n_sub = rs.randint(1, max_sub + 1, k)
    s2l = np.repeat(np.arange(k), n_sub).astype(np.int32)
    n = len(s2l)
    d = rs.standard_normal((n, 256))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    for a, b in dup:                                   # key-line b gets key-line a's sub-line descriptors (same count forced)
        ia, ib = np.nonzero(s2l == a)[0], np.nonzero(s2l == b)[0]
        m = min(len(ia), len(ib))
        d[ib[:m]] = d[ia[:m]]
    A = np.zeros((k, n), np.float32)
    for i in range(k):
        idx = np.nonzero(s2l == i)[0]
        A[i, idx] = np.float32(1.0 / len(idx))
    return d, s2l, A


def run(eng, d0, s0, k0, d1, s1, k1, thr, mutual=True):
    dk, _, m01 = eng.match(torch.from_numpy(d0).cuda(), np.array([0, len(d0)]), torch.from_numpy(s0).cuda(), np.array([0, k0]),
                           torch.from_numpy(d1).cuda(), np.array([0, len(d1)]), torch.from_numpy(s1).cuda(), np.array([0, k1]), thr, mutual)
    torch.cuda.synchronize()
    return dk.cpu().numpy().reshape(k0, k1), m01.cpu().numpy()


def run_in_a_batch_of_two(eng, d0, s0, k0, d1, s1, k1, thr, mutual=True):
    """the same pair twice in ONE linetr_match call: a batch always takes the three launches (pair_dist / pair_pool / pair_final)"""
    t = lambda a: torch.from_numpy(np.concatenate([a, a])).cuda()
    dk, off, m01 = eng.match(t(d0), np.array([0, len(d0), 2 * len(d0)]), t(s0), np.array([0, k0, 2 * k0]),
                             t(d1), np.array([0, len(d1), 2 * len(d1)]), t(s1), np.array([0, k1, 2 * k1]), thr, mutual)
    torch.cuda.synchronize()
    dk, m01 = dk.cpu().numpy(), m01.cpu().numpy()
    assert np.array_equal(dk[:k0 * k1], dk[k0 * k1:]) and np.array_equal(m01[:k0], m01[k0:])
    return dk[:k0 * k1].reshape(k0, k1), m01[:k0]


@pytest.mark.parametrize("k0,k1,max_sub,seed", [(199, 199, 1, 0), (37, 53, 3, 1), (1, 7, 2, 2), (16, 16, 1, 3), (17, 300, 4, 4),
                                                 (600, 599, 2, 5), (130, 1, 1, 6), (250, 400, 1, 7),
                                                 (1024, 1024, 1, 8),      # the largest image 1 of the one-launch path: 64 column tiles over 8 blocks per row chunk
                                                 (1030, 1030, 1, 9),      # one past it: the three launches
                                                 (512, 700, 1, 10)])      # point-matcher shape (identity maps, n0 != n1)
def test_fused_single_pair_matcher_vs_oracle_and_three_launches(eng, k0, k1, max_sub, seed):
    rs = np.random.RandomState(seed)
    dup0 = [(0, min(3, k0 - 1))] if k0 > 3 else []     # two identical key-lines in image 0: a COLUMN-argmin tie -> first index
    dup1 = [(1, min(5, k1 - 1))] if k1 > 5 else []     # two identical key-lines in image 1: a ROW-argmin tie -> first index
    d0, s0, A0 = make_side(rs, k0, max_sub, dup0 if max_sub == 1 else [])
    d1, s1, A1 = make_side(rs, k1, max_sub, dup1 if max_sub == 1 else [])
    D = O.dist_matrix(d0.T[None], d1.T[None])[0]
    Dk = O.subline2keyline(D, torch.from_numpy(A0), torch.from_numpy(A1))
    for thr, mutual in ((0.8, True), (2.5, True), (1.2, False)):
        want = O.mutual_nn(Dk, thr, mutual)[0]
        dk, m01 = run(eng, d0, s0, k0, d1, s1, k1, thr, mutual)
        got = np.zeros_like(want)
        got[np.nonzero(m01 >= 0)[0], m01[m01 >= 0]] = 1
        assert np.abs(dk - Dk[0]).max() < 5e-6
        assert np.array_equal(got, want), (thr, mutual)
        dk3, m3 = run_in_a_batch_of_two(eng, d0, s0, k0, d1, s1, k1, thr, mutual)
        assert np.array_equal(m01, m3) and np.abs(dk - dk3).max() < 5e-6
    dk_a, m_a = run(eng, d0, s0, k0, d1, s1, k1, 0.8, True)
    for _ in range(25):                                 # deterministic to the bit, and the slot is left clean every time
        dk_r, m_r = run(eng, d0, s0, k0, d1, s1, k1, 0.8, True)
        assert np.array_equal(m_r, m_a) and np.array_equal(dk_r, dk_a)


def test_fused_matcher_on_several_streams(eng):
    """one scratch slot per stream: calls queued on different streams must not see each other's column keys"""
    rs = np.random.RandomState(9)
    sets = [make_side(rs, 120 + 10 * i, 2) + make_side(rs, 140 - 5 * i, 2) for i in range(4)]
    ref = [run(eng, s[0], s[1], s[2].shape[0], s[3], s[4], s[5].shape[0], 0.9)[1] for s in sets]
    streams = [torch.cuda.Stream() for _ in sets]
    for rep in range(5):
        outs = []
        for st, s in zip(streams, sets):
            with torch.cuda.stream(st):
                outs.append(eng.match(torch.from_numpy(s[0]).cuda(), np.array([0, len(s[0])]), torch.from_numpy(s[1]).cuda(),
                                      np.array([0, s[2].shape[0]]), torch.from_numpy(s[3]).cuda(), np.array([0, len(s[3])]),
                                      torch.from_numpy(s[4]).cuda(), np.array([0, s[5].shape[0]]), 0.9, True)[2])
        torch.cuda.synchronize()
        for o, r in zip(outs, ref):
            assert np.array_equal(o.cpu().numpy(), r)


def test_pair_tail_equals_the_separate_calls(eng):
    """linetr_pair_tail (the matching tail of Matching.forward in one call: point matcher + line matcher + the four device -> host
    copies into one pinned block) returns exactly what linetr_match_points and linetr_match return; either branch can be skipped."""
    rs = np.random.RandomState(21)
    d0, s0, A0 = make_side(rs, 90, 3)
    d1, s1, A1 = make_side(rs, 120, 2)
    p0 = rs.standard_normal((256, 301)).astype(np.float32); p0 /= np.linalg.norm(p0, axis=0, keepdims=True)
    p1 = rs.standard_normal((256, 257)).astype(np.float32); p1 /= np.linalg.norm(p1, axis=0, keepdims=True)
    t = lambda a: torch.from_numpy(a).cuda()
    dist_ref, m_ref = eng.match_points(t(p0), t(p1), 0.7, True)
    dk_ref, m01_ref = run(eng, d0, s0, 90, d1, s1, 120, 0.8)
    got = eng.collect_tail(eng.pair_tail(t(p0), t(p1), 0.7, t(d0), t(s0), 90, t(d1), t(s1), 120, 0.8, True))
    assert np.array_equal(got[0], dist_ref.cpu().numpy()) and np.array_equal(got[1], m_ref.cpu().numpy())
    assert np.array_equal(got[2], dk_ref) and np.array_equal(got[3], m01_ref)
    only_pts = eng.collect_tail(eng.pair_tail(t(p0), t(p1), 0.7, None, None, 0, None, None, 0, 0.8, True))
    assert np.array_equal(only_pts[1], got[1]) and only_pts[2].shape == (0, 0) and only_pts[3].shape == (0,)
    only_lines = eng.collect_tail(eng.pair_tail(None, None, 0.7, t(d0), t(s0), 90, t(d1), t(s1), 120, 0.8, True))
    assert only_lines[0].shape == (0, 0) and np.array_equal(only_lines[2], got[2]) and np.array_equal(only_lines[3], got[3])
    # tickets collected out of order within the ring of four staging buffers
    a = eng.pair_tail(t(p0), t(p1), 0.7, t(d0), t(s0), 90, t(d1), t(s1), 120, 0.8, True)
    b = eng.pair_tail(t(p1), t(p0), 0.7, t(d1), t(s1), 120, t(d0), t(s0), 90, 0.8, True)
    rb, ra = eng.collect_tail(b), eng.collect_tail(a)
    assert np.array_equal(ra[3], got[3]) and rb[2].shape == (120, 90)
