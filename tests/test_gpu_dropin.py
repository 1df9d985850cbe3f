"""GPU: the reference's Python call surface (LineTransformer / Matching / nn_matcher / get_dist_matrix)
served by the HIP library, checked against the golden fixtures of the real reference."""
import numpy as np
import pytest
import torch

from helpers import BASE_CFG, TOK_KEYS, load, tiny_maps
from linetr_amd import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
LT_CFG = {"mode": "train", "max_keylines": -1, "min_length": 16, "token_distance": 8, "nn_threshold": 0.8}
REF_KEYS = ["klines", "length_klines", "angles", "sublines", "pnt_sublines", "mask_sublines", "resp_sublines",
            "angle_sublines", "desc_sublines", "score_sublines", "mat_klines2sublines"]


def make_lt(**cfg):
    from models.line_transformer import LineTransformer      # the import path the reference's scripts use
    m = LineTransformer({**LT_CFG, **cfg}).eval()
    m.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
    return m.to("cuda")


def test_linetransformer_preprocess_forward_cfg2():
    g = load("cfg2_pair")
    m = make_lt()
    outs = []
    for t in "ab":
        dd, ds = synth.synth_dense_maps(int(g[f"{t}_seed"]), 480, 640)
        sp = {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}
        kl = synth.array_to_keylines(g[f"{t}_lines"])
        pre = m.preprocess(kl, (1, 1, 480, 640), sp, torch.ones(1, 1, 480, 640).cuda())   # tensor mask: ignored
        assert list(pre.keys()) == REF_KEYS
        assert m.config["image_shape"] == (1, 1, 480, 640)                                  # quirk: config mutated
        for k in TOK_KEYS:
            want = g[f"{t}_{k}"]
            have = pre[k].cpu().numpy()
            assert have.shape == want.shape and have.dtype == np.float32, k
            assert np.abs(have - want).max() <= (1.2e-7 if "angle" in k else 0), k
        out = m(pre)
        assert out is pre and out["line_desc"].shape == (1, 256, 199)
        assert np.abs(out["line_desc"].cpu().numpy() - g[f"{t}_line_desc"]).max() < 1e-4
        outs.append(out)
    # the matching tail exactly as models/matching.py:77-84 writes it, on the drop-in functions
    from models.line_transformer import get_dist_matrix
    from models.nn_matcher import nn_matcher_distmat
    D = get_dist_matrix(outs[0]["line_desc"].cpu().numpy(), outs[1]["line_desc"].cpu().numpy())[0]
    assert D.dtype == np.float32 and np.abs(D - g["pair_D"]).max() < 1e-4
    Dk = m.subline2keyline(D, outs[0]["mat_klines2sublines"][0], outs[1]["mat_klines2sublines"][0])
    assert Dk.shape == (1, 199, 199) and np.abs(Dk - g["pair_Dk"]).max() < 1e-4
    M = nn_matcher_distmat(Dk, 0.8, True)
    assert M.dtype == np.float64 and np.array_equal(M, g["pair_M"])


def test_linetransformer_quirks_and_empty():
    g = load("tiny_validmask")
    dd, ds, hw = tiny_maps(g)
    sp = {"dense_descriptor": dd.cuda(), "dense_score": ds.cuda()}
    m = make_lt()
    vm = np.ones(hw)
    vm[:, :int(g["valid_mask_cols"])] = 0
    out = m(m.preprocess(synth.array_to_keylines(g["lines"]), (1, 1, *hw), sp, vm))       # ndarray mask honoured
    assert np.array_equal(out["klines"].cpu().numpy(), g["klines"])
    assert np.array_equal(out["mat_klines2sublines"].cpu().numpy(), g["mat_klines2sublines"])
    assert np.abs(out["line_desc"].cpu().numpy() - g["line_desc"]).max() < 1e-4
    g1 = load("tiny_single_line")
    pre = m.preprocess(synth.array_to_keylines(g1["lines"]), (1, 1, *hw), sp)
    assert len(pre["klines"]) == 0
    ret = m(pre)
    for k, v in ret.items():
        assert tuple(v.shape) == tuple(g1[f"ret_{k}_shape"])
    assert len(m.preprocess([], (1, 1, *hw), sp)["klines"]) == 0                          # zero detections: no crash
    bad = [synth.KeyLine(100, 100, 150, 100, length=400.0), synth.KeyLine(100, 200, 300, 200)]   # detector length > geometry
    with pytest.raises(AssertionError):
        m.preprocess(bad, (1, 1, *hw), sp)


class FakeSuperPoint(torch.nn.Module):
    """Synthetic SuperPoint stand-in: seeded dense maps + random unit point descriptors."""
    config = {"nn_threshold": 0.7}

    def __init__(self, seeds):
        super().__init__()
        self.seeds = list(seeds)

    def forward(self, data):
        seed = self.seeds.pop(0)
        dd, ds = synth.synth_dense_maps(seed, 480, 640)
        rs = np.random.RandomState(seed)
        n = 50 + seed % 7
        desc = torch.from_numpy(rs.standard_normal((256, n)).astype(np.float32))
        desc = torch.nn.functional.normalize(desc, dim=0)
        kp = torch.from_numpy(rs.uniform(8, 400, (n, 2)).astype(np.float32))
        dev = data["image"].device
        return {"keypoints": [kp.to(dev)], "scores": [torch.rand(n).to(dev)], "descriptors": [desc.to(dev)],
                "dense_descriptor": dd.to(dev), "dense_score": ds.to(dev)}


class FakeLSD:
    def __init__(self, line_sets):
        self.sets = list(line_sets)

    def detect_torch(self, image):
        return synth.array_to_keylines(self.sets.pop(0))


def test_matching_pipeline_outputs():
    from models.matching import Matching
    g = load("cfg2_pair")
    mt = Matching({"auto_min_length": True, "superpoint": {}, "lsd": {},
                   "linetransformer": {**LT_CFG}},
                  superpoint=FakeSuperPoint([int(g["a_seed"]), int(g["b_seed"])]), lsd=FakeLSD([g["a_lines"], g["b_lines"]]))
    mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
    mt = mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    pred = mt({"image0": img, "image1": img.clone()})
    for k in ("keypoints0", "keypoints1", "klines0", "klines1", "matches_p", "matching_scores_p", "matches_l",
              "matching_scores_l", "line_desc0", "line_desc1", "mat_klines2sublines0", "mat_klines2sublines1"):
        assert k in pred, k
    for k, v in pred.items():
        v[0]                                       # match_line_pairs.py:91-92 indexes every value
    assert mt.linetransformer.config["min_length"] == 16.0 and mt.linetransformer.config["token_distance"] == 8.0
    assert pred["matches_l"].dtype == torch.float64 and pred["matching_scores_l"].dtype == torch.float32
    assert np.array_equal(pred["matches_l"].numpy(), g["pair_M"])
    assert np.abs(pred["matching_scores_l"].numpy() - g["pair_Dk"]).max() < 1e-4
    assert pred["matches_p"].shape[0] == 1 and pred["matches_p"].dtype == torch.float64
    # np.savez payload of match_line_pairs.py:94-104
    out = {k: v[0].cpu().numpy() for k, v in pred.items() if torch.is_tensor(v[0])}
    assert out["klines0"].shape == (199, 2, 2) and out["matches_l"].shape == (199, 199)


def test_matching_forward_batch_equals_per_pair():
    """forward_batch (one fused describe + one match call for all pairs) vs forward() pair by pair."""
    from models.matching import Matching
    seeds = [31, 32, 33, 34, 35, 36]
    line_sets = [synth.synth_lines(s, 150 + 10 * i, 480, 640) for i, s in enumerate(seeds)]

    def build():
        mt = Matching({"auto_min_length": True, "linetransformer": {**LT_CFG}}, superpoint=FakeSuperPoint(seeds),
                      lsd=FakeLSD(line_sets))
        mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
        return mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    pairs = [{"image0": img, "image1": img.clone()} for _ in range(3)]
    batch = build().forward_batch(pairs)
    single = build()
    assert len(batch) == 3
    for p in range(3):
        ref = single({"image0": img, "image1": img.clone()})
        got = batch[p]
        for k in ("matches_l", "matches_p"):
            assert torch.equal(got[k], ref[k]), (p, k)
        assert (got["matching_scores_l"] - ref["matching_scores_l"]).abs().max().item() < 1e-5
        for k in ("klines0", "klines1", "mat_klines2sublines0", "sublines1"):
            assert torch.equal(got[k].cpu(), ref[k].cpu()), (p, k)
        assert (got["line_desc0"] - ref["line_desc0"]).abs().max().item() < 5e-6
        for v in got.values():
            v[0]


def test_forward_batch_uses_producer_layout():
    """A SuperPoint that also hands out 'dense_descriptor_nhwc' (FusedHeadSuperPoint): forward_batch feeds that map to
    linetr_describe (dense_layout='nhwc') and returns the same line descriptors and matches."""
    from models.matching import Matching
    seeds = [51, 52, 53, 54]
    line_sets = [synth.synth_lines(s, 120 + 15 * i, 480, 640) for i, s in enumerate(seeds)]

    class NhwcSuperPoint(FakeSuperPoint):
        def forward(self, data):
            out = super().forward(data)
            out["dense_descriptor_nhwc"] = out["dense_descriptor"].permute(0, 2, 3, 1).contiguous()
            return out

    def build(cls):
        mt = Matching({"auto_min_length": True, "linetransformer": {**LT_CFG}}, superpoint=cls(seeds), lsd=FakeLSD(line_sets))
        mt.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
        return mt.eval().to("cuda")
    img = torch.zeros(1, 1, 480, 640, device="cuda")
    pairs = [{"image0": img, "image1": img.clone()} for _ in range(2)]
    a = build(FakeSuperPoint).forward_batch(pairs)
    b = build(NhwcSuperPoint).forward_batch(pairs)
    for pa, pb in zip(a, b):
        assert torch.equal(pa["line_desc0"], pb["line_desc0"]) and torch.equal(pa["line_desc1"], pb["line_desc1"])
        assert torch.equal(pa["matches_l"], pb["matches_l"])
        assert "dense_descriptor_nhwc0" in pb and "dense_descriptor_nhwc0" not in pa


def test_nhwc_dense_layout_matches_nchw():
    from linetr_amd.engine import Engine
    eng = Engine(synth.calibrated_state_dict(), "cuda:0")
    rows = [synth.synth_lines(41, 120, 480, 640), synth.synth_lines(42, 90, 480, 640)]
    maps = [synth.synth_dense_maps(s, 480, 640) for s in (41, 42)]
    dd = torch.cat([m[0] for m in maps]).cuda()
    ds = torch.cat([m[1] for m in maps]).cuda()
    off = np.array([0, 120, 210], np.int32)
    kw = dict(remove_borders=8, min_length=16, max_keylines=-1, token_distance=8, max_tokens=21)
    _, a = eng.describe_lines(np.concatenate(rows), off, dd, ds, **kw)
    _, b = eng.describe_lines(np.concatenate(rows), off, dd.permute(0, 2, 3, 1).contiguous(), ds, dense_layout="nhwc", **kw)
    assert torch.equal(a, b)


def test_nn_matcher_known_answers():
    from models.nn_matcher import nn_matcher, nn_matcher_distmat
    g = load("matcher_cases")
    for k in ("ties", "big", "empty0", "empty1"):
        assert np.array_equal(nn_matcher_distmat(g[f"{k}_dist"], 0.8, True), g[f"{k}_mutual"]), k
        assert np.array_equal(nn_matcher_distmat(g[f"{k}_dist"], 0.8, False), g[f"{k}_oneway"]), k
    M, D = nn_matcher(g["point_desc0"], g["point_desc1"], 0.7, True)
    assert np.array_equal(M, g["point_M"]) and np.abs(D - g["point_D"]).max() < 1e-5
