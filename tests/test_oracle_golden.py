"""CPU: pin oracle/linetr_oracle.py against the fixtures frozen from the real reference
(tests/golden/make_golden.py).  Tokeniser entries bit-exact, descriptors <= 2e-5, matches identical."""
import os

import numpy as np
import pytest
import torch

from workloads import synth
from oracle import linetr_oracle as O
from helpers import BASE_CFG, TOK_KEYS, golden_cfg, load, oracle_image, tiny_maps, train_mode_batches, weights_for

torch.set_grad_enabled(False)
DESC_TOL = 2e-5


def test_cfg2_pair_tokenizer_forward_match():
    g = load("cfg2_pair")
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    outs = []
    for tag in "ab":
        dd, ds = synth.synth_dense_maps(int(g[f"{tag}_seed"]), 480, 640)
        out = oracle_image(sd, g[f"{tag}_lines"], dd, ds, (480, 640), BASE_CFG)
        for k in TOK_KEYS:
            assert np.array_equal(out[k].numpy(), g[f"{tag}_{k}"]), k
        desc = out["desc_sublines"].numpy()[0]
        ii, jj = g[f"{tag}_desc_sample_idx"].T
        assert np.array_equal(desc[ii, jj], g[f"{tag}_desc_sample"])
        assert np.abs(desc.astype(np.float64).sum(-1) - g[f"{tag}_desc_checksum"]).max() == 0
        assert np.abs(out["line_desc"].numpy() - g[f"{tag}_line_desc"]).max() < DESC_TOL
        assert out["line_desc"].shape == (1, 256, 199)          # the [:-1] drop (quirk 5)
        outs.append(out)
    M, Dk = O.match_lines(outs[0]["line_desc"], outs[1]["line_desc"], outs[0]["mat_klines2sublines"][0],
                          outs[1]["mat_klines2sublines"][0], 0.8)
    assert np.array_equal(M, g["pair_M"])
    assert M.dtype == np.float64 and Dk.dtype == np.float32
    assert np.abs(Dk - g["pair_Dk"]).max() < 1e-5


def test_mixed_size_pair_per_image_thresholds():
    """A 480 x 640 + 960 x 1280 pair through the reference's Matching.forward line branch with auto_min_length (matching.py:29-32,
    :45-48: thresholds per image; normalisation keeps the constructor's 480 x 640) -- tests/golden/make_golden_mixed.py."""
    g = load("mixed_size_pair")
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    outs = []
    for s in "01":
        hw = tuple(int(v) for v in g["hw" + s])
        dd, ds = synth.synth_dense_maps(int(g["map_seed" + s]), *hw)
        cfg = dict(BASE_CFG, min_length=max(16, max(hw) / 40), token_distance=max(8, max(hw) / 80))
        assert cfg["min_length"] == float(g["min_length" + s]) and cfg["token_distance"] == float(g["token_distance" + s])
        out = oracle_image(sd, g["lines" + s], dd, ds, hw, cfg, image_shape=(480, 640))
        for k in TOK_KEYS:
            assert np.array_equal(out[k].numpy(), g[k + s]), (k, s)
        assert np.abs(out["line_desc"].numpy() - g["line_desc" + s]).max() < DESC_TOL
        outs.append(out)
    M, Dk = O.match_lines(outs[0]["line_desc"], outs[1]["line_desc"], outs[0]["mat_klines2sublines"][0],
                          outs[1]["mat_klines2sublines"][0], 0.8)
    assert np.array_equal(M, g["matches_l"]) and np.abs(Dk - g["matching_scores_l"]).max() < 1e-5


def test_asset_pair_real_image_statistics():
    """The first pair of the reference's demo list (assets/input_pairs.txt:1) through the REFERENCE's SuperPoint (seeded weights) and the
    line branch of its Matching.forward (tests/golden/make_golden_asset_pair.py): dense maps with real-image statistics -- neighbouring
    descriptor cells at cosine 0.98 -- instead of the i.i.d. maps of every other fixture.  The oracle must reproduce tokens bit for bit,
    descriptors, Dk and both match sets."""
    g = load("asset_pair")
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    dd0 = g["dense_descriptor0"][0]
    assert float((dd0[:, :, 1:] * dd0[:, :, :-1]).sum(0).mean()) > 0.95         # this IS the correlated regime
    outs = []
    for s in "01":
        out = oracle_image(sd, g["lines" + s], torch.from_numpy(g["dense_descriptor" + s]), torch.from_numpy(g["dense_score" + s]),
                           (480, 640), BASE_CFG)
        for k in TOK_KEYS:
            assert np.array_equal(out[k].numpy(), g[k + s]), (k, s)
        desc = out["desc_sublines"].numpy()[0]
        ii, jj = g["desc_sample_idx" + s].T
        assert np.array_equal(desc[ii, jj], g["desc_sample" + s])
        assert np.abs(desc.astype(np.float64).sum(-1) - g["desc_checksum" + s]).max() == 0
        assert np.abs(out["line_desc"].numpy() - g["line_desc" + s]).max() < DESC_TOL
        outs.append(out)
    assert outs[0]["sublines"].shape[1] > outs[0]["klines"].shape[1]             # lines longer than 168 px: several sub-lines per line
    M, Dk = O.match_lines(outs[0]["line_desc"], outs[1]["line_desc"], outs[0]["mat_klines2sublines"][0],
                          outs[1]["mat_klines2sublines"][0], 0.8)
    assert np.array_equal(M, g["matches_l"]) and M.sum() >= 40
    assert np.abs(Dk - g["matching_scores_l"]).max() < 1e-5
    Mp, Dp = O.point_nn(g["descriptors0"], g["descriptors1"], 0.7, True)
    idx = np.where(Mp[0].sum(1) > 0, Mp[0].argmax(1), -1)
    assert np.array_equal(idx, g["matches_p_index"]) and int(Mp.sum()) == int(g["matches_p_count"])
    assert np.abs(Dp[0].min(1) - g["matching_scores_p_rowmin"]).max() < 1e-5


def test_jitter_pair_recovers_permutation():
    g = load("cfg2_jitter_pair")
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    dd, ds = synth.synth_dense_maps(int(g["map_seed"]), 480, 640)
    o0 = oracle_image(sd, g["a_lines"], dd, ds, (480, 640), BASE_CFG)
    o1 = oracle_image(sd, g["b_lines"], dd, ds, (480, 640), BASE_CFG)
    M, _ = O.match_lines(o0["line_desc"], o1["line_desc"], o0["mat_klines2sublines"][0],
                         o1["mat_klines2sublines"][0], 0.8)
    assert np.array_equal(M, g["pair_M"])
    # every matched pair is the same physical line (start points within the jitter)
    i, j = np.nonzero(M[0])
    assert len(i) >= 195
    k0, k1 = o0["klines"][0].numpy(), o1["klines"][0].numpy()
    err = np.minimum(np.abs(k0[i] - k1[j]).reshape(-1, 4).max(1),
                     np.abs(k0[i] - k1[j][:, ::-1]).reshape(-1, 4).max(1))   # near-vertical lines may swap ends
    assert (err < 2.0).sum() >= 195, err


@pytest.mark.parametrize("name", ["tiny_default", "tiny_float_td", "tiny_align_true", "tiny_two_layers",
                                  "tiny_max3", "tiny_noborder"])
def test_tiny_cases(name):
    g = load(name)
    dd, ds, hw = tiny_maps(g)
    sd = synth.to_torch_state_dict(weights_for(g))
    cfg = golden_cfg(g)
    out = oracle_image(sd, g["lines"].copy(), dd, ds, hw, cfg, align_corners=bool(g["align_corners"]))
    for k in TOK_KEYS:
        assert np.array_equal(out[k].numpy(), g[k]), k
    assert np.abs(out["desc_sublines"].numpy() - g["desc_sublines"]).max() < 1e-6
    assert np.abs(out["line_desc"].numpy() - g["line_desc"]).max() < DESC_TOL


def test_tiny_validmask_and_single_line():
    g = load("tiny_validmask")
    dd, ds, hw = tiny_maps(g)
    sd = synth.to_torch_state_dict(weights_for(g))
    vm = np.ones(hw)
    vm[:, :int(g["valid_mask_cols"])] = 0
    out = oracle_image(sd, g["lines"].copy(), dd, ds, hw, BASE_CFG, valid_mask=vm)
    for k in ("klines", "sublines", "mat_klines2sublines"):
        assert np.array_equal(out[k].numpy(), g[k]), k
    assert np.abs(out["line_desc"].numpy() - g["line_desc"]).max() < DESC_TOL
    g = load("tiny_single_line")
    out = O.preprocess(synth.array_to_keylines(g["lines"]), (1, 1, *hw), dd, ds, BASE_CFG)
    assert len(out["klines"]) == int(g["n_klines"]) == 0
    ret = O.forward(sd, out, hw)
    for k, v in ret.items():
        assert tuple(v.shape) == tuple(g[f"ret_{k}_shape"])


def test_cfg5_small_long_tokens():
    g = load("cfg5_small")
    dd, ds, hw = tiny_maps(g)
    sd = synth.to_torch_state_dict(weights_for(g))
    cfg = dict(BASE_CFG, max_tokens=int(g["max_tokens"]))
    out = oracle_image(sd, g["lines"], dd, ds, hw, cfg)
    for k in ("klines", "sublines", "pnt_sublines", "mask_sublines", "resp_sublines", "angle_sublines",
              "score_sublines"):
        assert np.array_equal(out[k].numpy(), g[k]), k
    assert out["pnt_sublines"].shape[2] == 41
    desc = out["desc_sublines"].numpy()[0].astype(np.float64)
    assert np.abs(desc.sum(-1) - g["desc_checksum"]).max() < 1e-5
    assert np.abs(out["line_desc"].numpy() - g["line_desc"]).max() < DESC_TOL


def test_matcher_known_answers():
    g = load("matcher_cases")
    for k in ("ties", "big", "empty0", "empty1"):
        d = g[f"{k}_dist"]
        assert np.array_equal(O.mutual_nn(d, 0.8, True), g[f"{k}_mutual"]), k
        assert np.array_equal(O.mutual_nn(d, 0.8, False), g[f"{k}_oneway"]), k
    M, D = O.point_nn(g["point_desc0"], g["point_desc1"], 0.7, True)
    assert np.array_equal(M, g["point_M"])
    assert np.abs(D - g["point_D"]).max() < 1e-6


def test_superpoint_heads_oracle_matches_reference_fixture():
    """8(f) row 2: the oracle's head post-processing vs the dense maps the real SuperPoint.forward returned."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "superpoint_heads.npz"))
    score, desc = O.superpoint_heads(g["score_logits"], g["desc_raw"])
    assert score.shape == g["dense_score"].shape == (2, 64, 96)
    assert np.array_equal(score, g["dense_score"])
    assert np.array_equal(desc, g["dense_descriptor"])
    # what the maps mean: every 8x8 patch + its dustbin is a probability distribution; descriptors are unit vectors
    patch = score.reshape(2, 8, 8, 12, 8).sum(axis=(2, 4))
    assert (patch <= 1 + 1e-6).all() and (patch > 0).all()
    assert np.abs(np.linalg.norm(desc, axis=1) - 1).max() < 1e-6


def test_training_time_batched_forward():
    """8(f) row 4: the model called as train.py:163-164 calls it -- B = 3 images, a fixed 40 sub-lines each, 12
    line-descriptive layers -- frozen from the real reference (make_golden_train.py)."""
    g = load("train_batch")
    hw = tuple(int(v) for v in g["hw"])
    n_fix, nl = int(g["n_fix"]), int(g["n_desc_layers"])
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict(nl))
    keys = ("sublines", "pnt_sublines", "desc_sublines", "score_sublines", "mask_sublines", "resp_sublines", "angle_sublines", "klines")
    outs = []
    for b in range(3):
        dd, ds = synth.synth_dense_maps_np(int(g[f"map_seed_{b}"]), *hw)
        outs.append(O.preprocess(synth.array_to_keylines(g[f"lines_{b}"]), (1, 1, *hw), torch.from_numpy(dd), torch.from_numpy(ds),
                                 dict(BASE_CFG)))
    batch = {k: torch.cat([o[k][:, :n_fix] for o in outs], dim=0) for k in keys}
    got = O.forward_batch(sd, batch, hw)["line_desc"].numpy()
    assert got.shape == g["line_desc"].shape == (3, 256, n_fix)
    assert np.abs(got - g["line_desc"]).max() < 2e-6


def _oracle_train_batches(g):
    hw = tuple(int(v) for v in g["hw"])
    pre = lambda rows, pred: O.preprocess(synth.array_to_keylines(rows), (1, 1, *hw), pred["dense_descriptor"], pred["dense_score"],
                                          dict(BASE_CFG))
    tok = lambda lines, pred: O.tokenize(lines, 8, 21, pred["dense_descriptor"], pred["dense_score"], (640, 480))
    return train_mode_batches(g, pre, tok)


def test_pseudo_lines_tokenised_with_the_dataset_builders_swapped_image_shape():
    """conv_fixed_size hands conf['data']['resize'] = (640, 480) = (width, height) to line_tokenizer as `image_shape`
    (dataloaders/utils/util_lines.py:682,703), which reads it as (height, width) and clips end points at x <= 479.4
    (models/line_process.py:101,115-116): 29 of the 111 pseudo lines of the fixture are bent by it."""
    g = load("train_mode")
    dd, ds = synth.synth_dense_maps_np(int(g["map_seed_0_0"]), 480, 640)
    lines = {k: g[f"pseudo_{k}_0_0"].copy() for k in ("klines", "length_klines", "angles")}
    assert (lines["klines"][:, 1, 0] > 479.4).sum() == 29
    out = O.tokenize(lines, 8, 21, torch.from_numpy(dd), torch.from_numpy(ds), (640, 480))
    for k in TOK_KEYS:
        assert np.array_equal(out[k].numpy(), g[f"pseudo_tok_{k}"]), k


def test_train_mode_forward_and_running_statistics():
    """8(f) row 4, train mode: the reference in .train() with dropout probability 0, called twice on batches of 3 x 250 sub-lines
    padded / truncated by its own conv_fixed_size (make_golden_train_mode.py): BatchNorm on batch statistics, running statistics and
    num_batches_tracked after each call."""
    g = load("train_mode")
    hw = tuple(int(v) for v in g["hw"])
    nl = int(g["n_desc_layers"])
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict(nl))
    sd = {k: v.clone() for k, v in sd.items()}
    bn_keys = [k for k in sd if k.endswith("running_mean")]
    assert len(bn_keys) == 15
    for c, batch in enumerate(_oracle_train_batches(g)):
        got = O.forward_train(sd, batch, hw)["line_desc"].numpy()
        want = g[f"line_desc_{c}"]
        assert np.abs((got if c == 0 else got[:, :, ::5]) - want).max() < 5e-6
        for k in bn_keys:
            for kk in (k, k.replace("running_mean", "running_var")):
                assert np.abs(sd[kk].numpy() - g[f"bn{c}.{kk}"]).max() <= 1e-6 * max(1.0, np.abs(g[f"bn{c}.{kk}"]).max()), (c, kk)
            nbt = k.replace("running_mean", "num_batches_tracked")
            assert int(sd[nbt]) == int(g[f"bn{c}.{nbt}"])
