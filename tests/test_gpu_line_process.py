"""GPU: the module-level surface of the reference's ``models.line_process`` -- what
``dataloaders/build_homography_dataset.py:19,198-206`` imports and calls -- served by the HIP library, against the golden
fixtures of the real reference; plus the smaller entry points added with it (sub-line pooling alone, slab packing)."""
import numpy as np
import pytest
import torch

from helpers import TOK_KEYS, load
from workloads import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

REF_KEYS = ["klines", "length_klines", "angles", "sublines", "pnt_sublines", "mask_sublines", "resp_sublines",
            "angle_sublines", "desc_sublines", "score_sublines", "mat_klines2sublines"]
CONF = {"min_length": 16, "max_sublines": -1, "token_distance": 8, "max_tokens": 21, "remove_borders": 8}


def test_preprocess_and_line_tokenizer_as_the_dataset_builder_imports_them():
    from models.line_process import preprocess, line_tokenizer      # dataloaders/build_homography_dataset.py:19
    g = load("cfg2_pair")
    for t in "ab":
        dd, ds = synth.synth_dense_maps(int(g[f"{t}_seed"]), 480, 640)
        pred = {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}
        kl = synth.array_to_keylines(g[f"{t}_lines"])
        out = preprocess(kl, (480, 640), pred, mask=None, conf=CONF)                     # :198-199
        assert list(out.keys()) == REF_KEYS
        for k in TOK_KEYS:
            want, have = g[f"{t}_{k}"], out[k].cpu().numpy()
            assert have.shape == want.shape and have.dtype == np.float32, k
            assert np.abs(have - want).max() <= (1.2e-7 if "angle" in k else 0), k
        idx = g[f"{t}_desc_sample_idx"]                                                   # [64,2] (sub-line, token) of the frozen samples
        assert np.abs(out["desc_sublines"][0].cpu().numpy()[idx[:, 0], idx[:, 1]] - g[f"{t}_desc_sample"]).max() <= 1e-6
        # line_tokenizer on its own, the way conv_fixed_size calls it (dataloaders/utils/util_lines.py:713): float64 arrays in
        from models.line_process import change_cv2_T_np, filter_by_length, remove_borders
        lines = filter_by_length(remove_borders(change_cv2_T_np(kl), 8, 480, 640, None), 16, -1)
        out2 = line_tokenizer(lines, 8, 21, pred, (480, 640))
        for k in REF_KEYS:
            assert torch.equal(out2[k], out[k]), k
        # `image_shape` only sets the end-point clip (line_process.py:101,115-116): one that clips nothing gives the same tensors
        out3 = line_tokenizer(filter_by_length(remove_borders(change_cv2_T_np(kl), 8, 480, 640, None), 16, -1), 8, 21, pred, (4800, 6400))
        for k in REF_KEYS:
            assert torch.equal(out3[k], out[k]), k
    # an ndarray mask is honoured by the module-level function too (line_process.py:76-80)
    vm = np.ones((480, 640)); vm[:, :320] = 0
    masked = preprocess(synth.array_to_keylines(g["a_lines"]), (480, 640), pred, mask=vm, conf=CONF)
    assert 0 < masked["klines"].shape[1] < 199
    assert len(preprocess([], (480, 640), pred, conf=CONF)["klines"]) == 0


def test_scalar_helpers_and_sample_descriptors():
    from models.line_process import get_line_dist, point_on_line, sample_descriptors
    from oracle import linetr_oracle as O
    line = np.array([[10.0, 20.0], [110.0, 95.0]])
    assert get_line_dist(line) == 125.0
    p = point_on_line(line, 50.0)
    assert np.array_equal(p, O.walk_along(line[0], line[1], np.array([50.0]))[0])
    assert np.array_equal(point_on_line(np.array([[5.0, 30.0], [5.0, 10.0]]), 8.0), np.array([5.0, 22.0]))
    with pytest.raises(AssertionError):
        point_on_line(line, 126.0)
    dd, _ = synth.synth_dense_maps(5, 480, 640)
    rs = np.random.RandomState(0)
    pts = torch.from_numpy(np.stack([rs.uniform(0, 639, 300), rs.uniform(0, 479, 300)], 1).astype(np.float32))
    got = sample_descriptors(pts[None].cuda(), dd.cuda(), 8)
    want = O.sample_token_desc(pts[None], dd, 8)
    assert got.shape == want.shape == (1, 256, 300)
    assert (got.cpu() - want).abs().max().item() <= 1e-6
    with pytest.raises(ValueError):
        sample_descriptors(pts[None].cuda(), dd.cuda(), 4)


def test_subline2keyline_pooled_natively_and_as_given():
    """subline2keyline reads the matrices by their CONTENTS, on the device (linetr_pool_distmat_dense): a tokeniser's matrix is
    reduced to its map and pooled by the matcher's segmented-mean kernel, anything else is multiplied out as given
    (models/line_transformer.py:277-282).  No torch matmul behind it."""
    from models.line_transformer import LineTransformer
    from linetr_amd.line_process import sub2line_of
    m = LineTransformer({"mode": "train", "max_keylines": -1, "min_length": 16, "token_distance": 8, "max_tokens": 3}).eval()
    m.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
    m = m.to("cuda")
    pre = []
    for seed in (21, 22):
        dd, ds = synth.synth_dense_maps(seed, 480, 640)
        pre.append(m.preprocess(synth.array_to_keylines(synth.synth_lines(seed, 60, 480, 640)), (1, 1, 480, 640),
                                {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}))
    A0, A1 = pre[0]["mat_klines2sublines"][0], pre[1]["mat_klines2sublines"][0]      # what matching.py:80 passes
    assert sub2line_of(pre[0]["mat_klines2sublines"]) is not None and sub2line_of(A0) is None
    assert (A0 > 0).sum(1).max() > 1, "max_tokens=3 must give multi-sub-line key-lines"
    rs = np.random.RandomState(1)
    D = rs.rand(A0.shape[1], A1.shape[1]).astype(np.float32)
    ref = lambda a0, a1: (a0.double().cpu().numpy() @ D.astype(np.float64) @ a1.double().cpu().numpy().T)[None]
    for a0, a1 in ((A0, A1), (A0.cpu(), A1.cpu())):                                  # device and host tensors alike
        got = m.subline2keyline(D, a0, a1)
        assert got.shape == (1, A0.shape[0], A1.shape[0]) and got.dtype == np.float32
        assert np.abs(got - ref(a0, a1)).max() < 1e-6
    # the segmented-mean kernel served the call above: its result equals the map-driven entry point's bit for bit
    eng = m.engine()
    want = eng.pool_distmat(torch.from_numpy(D).cuda(), sub2line_of(pre[0]["mat_klines2sublines"]), A0.shape[0],
                            sub2line_of(pre[1]["mat_klines2sublines"]), A1.shape[0])
    assert np.array_equal(m.subline2keyline(D, A0, A1)[0], want.cpu().numpy())
    # matrices that are NOT a tokeniser's are multiplied out as given: a wrong weight, two non-zeros in a column, rows out of
    # order, a key-line without sub-lines, a dense random matrix, more key-lines than sub-lines
    B = A0.clone(); B[0, int(torch.nonzero(A0[0])[0])] *= 0.5
    C2 = A0.clone(); C2[1, 0] = 0.25
    Pm = A0.clone()[torch.randperm(A0.shape[0], generator=torch.Generator().manual_seed(0)).cuda()]
    Z = A0.clone(); Z[3] = 0
    R = torch.from_numpy(rs.standard_normal(tuple(A0.shape)).astype(np.float32)).cuda()
    for a0 in (B, C2, Pm, Z, R):
        assert np.abs(m.subline2keyline(D, a0, A1) - ref(a0, A1)).max() < 2e-5
        assert np.abs(m.subline2keyline(D.T.copy(), A1, a0) - np.transpose(ref(a0, A1), (0, 2, 1))).max() < 2e-5
    tall = torch.from_numpy(rs.standard_normal((A0.shape[1] + 5, A0.shape[1])).astype(np.float32)).cuda()
    assert np.abs(m.subline2keyline(D, tall, A1) - ref(tall, A1)).max() < 2e-5
    with pytest.raises(ValueError):
        m.subline2keyline(D, A0[:, :-1], A1)


def test_edited_matrix_is_matched_by_its_contents():
    """A mat_klines2sublines written to after the tokeniser made it loses its attached map (version stamp): Matching.match_lines then
    reads the matrix itself, like the reference's `A0 @ D @ A1.T` does (models/matching.py:77-84)."""
    from models.matching import Matching
    from linetr_amd.line_process import sub2line_of
    from models.line_transformer import LineTransformer
    m = LineTransformer({"mode": "train", "max_keylines": -1, "min_length": 16, "token_distance": 8, "max_tokens": 3,
                         "nn_threshold": 0.8}).eval()
    m.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
    m = m.to("cuda")
    outs = []
    for seed in (31, 32):
        dd, ds = synth.synth_dense_maps(seed, 480, 640)
        outs.append(m(m.preprocess(synth.array_to_keylines(synth.synth_lines(seed, 50, 480, 640)), (1, 1, 480, 640),
                                   {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()})))
    mt = Matching.__new__(Matching)
    torch.nn.Module.__init__(mt)
    mt.linetransformer = m
    args = lambda: (outs[0]["line_desc"], outs[0]["mat_klines2sublines"], outs[1]["line_desc"], outs[1]["mat_klines2sublines"], 0.8)
    M_fast, Dk_fast = mt.match_lines(*args())
    # same matrices without their maps (cloned: what an .npz reload gives): the contents path must agree exactly
    a, b, c, d, thr = args()
    M_slow, Dk_slow = mt.match_lines(a, b.clone(), c, d.clone(), thr)
    # (same matches; the distances agree to fp32 round-off -- the contents path forms D in the point matcher's launch, whose MFMA
    # accumulation order is not the fused line matcher's)
    assert np.array_equal(M_fast, M_slow) and np.abs(Dk_fast - Dk_slow).max() < 2e-6
    # edit in place: key-line 0 of image 0 now ignores its sub-lines (row zeroed) -> Dk row 0 becomes 0, as A0 @ D @ A1^T says
    outs[0]["mat_klines2sublines"][0, 0].zero_()
    assert sub2line_of(outs[0]["mat_klines2sublines"]) is None
    M_ed, Dk_ed = mt.match_lines(*args())
    assert np.all(Dk_ed[0, 0] == 0) and np.abs(Dk_ed[0, 1:] - Dk_fast[0, 1:]).max() < 1e-6


def test_pack_slab_kernel_equals_the_host_packing():
    from linetr_amd import parallel
    rs = np.random.RandomState(3)
    cu_n = np.array([0, 5, 5, 47, 300], np.int32)            # an image without sub-lines in the middle
    cu_k = np.array([0, 4, 4, 40, 250], np.int32)
    N = int(cu_n[-1])
    ld = torch.from_numpy(rs.randn(N, 256).astype(np.float32))
    s2l = torch.from_numpy(rs.randint(0, 50, N).astype(np.int32))
    cap_img, cap_rows = 6, 333
    want = parallel.pack_descriptors(ld, cu_n, cap_img, cap_rows, cu_k=cu_k, sub2line=s2l)            # CPU tensors: host form
    out = torch.full((parallel.slab_rows(cap_img, cap_rows), 256), 7.0, device="cuda")
    got = parallel.pack_descriptors(ld.cuda(), cu_n, cap_img, cap_rows, out=out, cu_k=cu_k, sub2line=s2l.cuda(),
                                    d_cu_n=torch.from_numpy(cu_n).cuda(), d_cu_k=torch.from_numpy(cu_k).cuda())
    assert got.data_ptr() == out.data_ptr()
    # prefix sums of another integer dtype / on the host / non-contiguous are converted, never reinterpreted
    got64 = parallel.pack_descriptors(ld.cuda(), cu_n, cap_img, cap_rows, out=torch.full_like(out, 7.0), cu_k=cu_k, sub2line=s2l.cuda(),
                                      d_cu_n=torch.from_numpy(cu_n.astype(np.int64)).cuda(),
                                      d_cu_k=torch.from_numpy(np.repeat(cu_k, 2)).cuda()[::2])
    assert torch.equal(got64, got)
    hr, mr = parallel.header_rows(cap_img), parallel.map_rows(cap_rows)
    g, w = got.cpu(), want
    assert torch.equal(g[:hr].view(torch.int32).view(-1)[:1 + 2 * cap_img], w[:hr].view(torch.int32).view(-1)[:1 + 2 * cap_img])
    assert torch.equal(g[hr:hr + mr].view(torch.int32).view(-1)[:N], w[hr:hr + mr].view(torch.int32).view(-1)[:N])
    assert torch.equal(g[hr + mr:hr + mr + N], w[hr + mr:hr + mr + N])
    d, cu, m, ck = parallel.unpack_descriptors(got, cap_img, cap_rows, with_lines=True)
    assert np.array_equal(cu, cu_n) and np.array_equal(ck, cu_k) and torch.equal(d.cpu(), ld) and torch.equal(m.cpu(), s2l)
    with pytest.raises(Exception):
        parallel.pack_descriptors(ld.cuda(), cu_n, 2, cap_rows, d_cu_n=torch.from_numpy(cu_n).cuda())
