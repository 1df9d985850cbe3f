"""CPU: host-side logic of the drop-in surface (no device compute): state_dict layout, config handling,
loud failure without a HIP device, NumPy pre-filter glue vs the oracle."""
import numpy as np
import pytest
import torch

from helpers import BASE_CFG, load
from workloads import synth
from linetr_amd import line_transformer as LT
from oracle import linetr_oracle as O


def test_state_dict_layout_and_config():
    m = LT.LineTransformer({"mode": "train", "nn_threshold": 0.8, "n_line_descriptive_layers": 2})
    sd = synth.make_state_dict(1, 2)
    assert list(m.state_dict().keys()) == list(sd.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    m.load_state_dict(synth.to_torch_state_dict(sd), strict=True)
    assert sum(v.numel() for v in LT.LineTransformer({"mode": "train"}).state_dict().values()) == 5692879
    assert m.config["max_tokens"] == 21 and m.config["nn_threshold"] == 0.8 and m.image_shape == [480, 640]
    assert set(LT.LineTransformer.default_config) == {"mode", "image_shape", "min_length", "token_distance",
                                                      "max_tokens", "remove_borders", "max_keylines", "descriptor_dim",
                                                      "keyline_encoder", "n_heads", "n_line_descriptive_layers", "d_inner"}
    ret = m.default_ret()
    assert ret["line_desc"].shape == (1, 256, 0) and ret["mat_klines2sublines"].shape == (1, 0, 0)


def test_mode_test_requires_weight_file():
    with pytest.raises(FileNotFoundError):
        LT.LineTransformer({})           # mode == 'test' -> weights/LineTR_weight.pth (absent blob)


def test_mode_test_finds_the_checkpoint_where_the_reference_keeps_it(tmp_path, monkeypatch, capsys):
    """models/line_transformer.py:220-223: mode 'test' loads <models package>/weights/LineTR_weight.pth strictly and prints a message.
    After the drop-in that directory belongs to the `models/` shim, so that is where the file is looked for first; $LINETR_WEIGHTS last."""
    import models
    sd = LT.LineTransformer({"mode": "train"}).state_dict()
    w = tmp_path / "models" / "weights"
    w.mkdir(parents=True)
    torch.save(sd, w / "LineTR_weight.pth")
    monkeypatch.setattr(models, "__path__", [str(tmp_path / "models")] + list(models.__path__))
    m = LT.LineTransformer({})
    assert "Loaded Line-Transformer model" in capsys.readouterr().out
    assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
    monkeypatch.setattr(models, "__path__", list(models.__path__)[1:])
    monkeypatch.setenv("LINETR_WEIGHTS", str(w / "LineTR_weight.pth"))
    LT.LineTransformer({"mode": "test"})


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_cpu_fallback():
    m = LT.LineTransformer({"mode": "train"})
    data = {"klines": torch.zeros(1, 3, 2, 2), "sublines": torch.zeros(1, 3, 2, 2), "pnt_sublines": torch.zeros(1, 3, 21, 2),
            "resp_sublines": torch.zeros(1, 3, 1), "angle_sublines": torch.zeros(1, 3, 2),
            "desc_sublines": torch.zeros(1, 3, 21, 256), "score_sublines": torch.zeros(1, 3, 21, 1),
            "mask_sublines": torch.ones(1, 3, 22, 1)}
    with pytest.raises(RuntimeError, match="train mode"):      # a module left in train mode (train.py:127) is refused, not silently
        m(data)                                                # run with BatchNorm(eval) and without dropout
    m.eval()
    with pytest.raises(RuntimeError, match="HIP device"):
        m(data)
    from linetr_amd.nn_matcher import nn_matcher_distmat
    with pytest.raises(RuntimeError, match="HIP device"):
        nn_matcher_distmat(np.ones((1, 2, 2), np.float32), 0.8)
    from linetr_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(synth.make_state_dict(0), "cpu")


@pytest.mark.parametrize("name", ["tiny_default", "tiny_noborder", "tiny_max3"])
def test_numpy_prefilter_glue_matches_oracle(name):
    g = load(name)
    cfg = dict(BASE_CFG)
    for k in g.files:
        if k.startswith("cfg_"):
            cfg[k[4:]] = g[k].item()
    hw = tuple(int(v) for v in g["hw"])
    kl = synth.array_to_keylines(g["lines"])
    a = LT.filter_by_length(LT.remove_borders(LT.change_cv2_T_np(kl), cfg["remove_borders"], hw[0], hw[1], np.ones(hw)),
                            cfg["min_length"], cfg["max_keylines"])
    b = O.keep_long_lines(O.drop_border_lines(O.cv2_to_arrays(kl), cfg["remove_borders"], hw[0], hw[1], np.ones(hw)),
                          cfg["min_length"], cfg["max_keylines"])
    for k in ("klines", "length_klines", "angles"):
        assert np.array_equal(a[k], b[k]), k
    assert len(LT.change_cv2_T_np([])["klines"]) == 0
