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
        with pytest.raises(ValueError):
            line_tokenizer(filter_by_length(remove_borders(change_cv2_T_np(kl), 8, 480, 640, None), 16, -1), 8, 21, pred, (481, 640))
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
    """Matrices from this package's tokeniser are pooled by the matcher's segmented-mean kernel; a foreign matrix is multiplied
    out as given (models/line_transformer.py:277-282)."""
    from models.line_transformer import LineTransformer
    m = LineTransformer({"mode": "train", "max_keylines": -1, "min_length": 16, "token_distance": 8, "max_tokens": 3}).eval()
    m.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
    m = m.to("cuda")
    pre = []
    for seed in (21, 22):
        dd, ds = synth.synth_dense_maps(seed, 480, 640)
        pre.append(m.preprocess(synth.array_to_keylines(synth.synth_lines(seed, 60, 480, 640)), (1, 1, 480, 640),
                                {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}))
    A0, A1 = pre[0]["mat_klines2sublines"][0], pre[1]["mat_klines2sublines"][0]
    assert hasattr(pre[0]["mat_klines2sublines"], "_linetr_sub2line")
    assert (A0 > 0).sum(1).max() > 1, "max_tokens=3 must give multi-sub-line key-lines"
    D = np.random.RandomState(1).rand(A0.shape[1], A1.shape[1]).astype(np.float32)
    want = (A0.double().cpu().numpy() @ D.astype(np.float64) @ A1.double().cpu().numpy().T)[None]
    native = m.subline2keyline(D, pre[0]["mat_klines2sublines"], pre[1]["mat_klines2sublines"])      # carries the map
    plain = m.subline2keyline(D, A0.clone(), A1.clone())                                              # foreign tensors
    assert native.shape == plain.shape == want.shape and native.dtype == np.float32
    assert np.abs(native - want).max() < 1e-6 and np.abs(plain - want).max() < 1e-6


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
    hr, mr = parallel.header_rows(cap_img), parallel.map_rows(cap_rows)
    g, w = got.cpu(), want
    assert torch.equal(g[:hr].view(torch.int32).view(-1)[:1 + 2 * cap_img], w[:hr].view(torch.int32).view(-1)[:1 + 2 * cap_img])
    assert torch.equal(g[hr:hr + mr].view(torch.int32).view(-1)[:N], w[hr:hr + mr].view(torch.int32).view(-1)[:N])
    assert torch.equal(g[hr + mr:hr + mr + N], w[hr + mr:hr + mr + N])
    d, cu, m, ck = parallel.unpack_descriptors(got, cap_img, cap_rows, with_lines=True)
    assert np.array_equal(cu, cu_n) and np.array_equal(ck, cu_k) and torch.equal(d.cpu(), ld) and torch.equal(m.cpu(), s2l)
    with pytest.raises(Exception):
        parallel.pack_descriptors(ld.cuda(), cu_n, 2, cap_rows, d_cu_n=torch.from_numpy(cu_n).cuda())
