"""cProfile of the reference call surface for one pair (LineTransformer.preprocess + forward x2 + Matching.match_lines through
the models.* shim): where the HOST time of the drop-in path goes.   python tools/dropin_hostprof.py   (on the GPU box)"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from workloads import synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    from linetr_amd.engine import Engine
    eng = Engine(synth.calibrated_state_dict(), dev)
    lines, dd, nhwc, ds, hw, T = bench.make_inputs("cfg2", 1, 0, dev, eng)
    from models.matching import Matching
    H, W = hw
    klines = [synth.array_to_keylines(l) for l in lines[:2]]
    sp = [{"dense_descriptor": dd[i:i + 1], "dense_score": ds[i:i + 1]} for i in range(2)]
    m = Matching({"auto_min_length": False, "linetransformer": {"mode": "train", "max_tokens": T, "image_shape": [H, W],
                                                                "min_length": 16, "token_distance": 8, "remove_borders": 8,
                                                                "max_keylines": -1, "nn_threshold": 0.8}},
                 superpoint=bench._StubSuperPoint([{}]), lsd=bench._StubLSD(klines))
    m.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
    m = m.to(dev).eval()
    lt = m.linetransformer
    img = torch.zeros(1, 1, H, W, device=dev)

    def pair():
        o = [lt(lt.preprocess(klines[i], (1, 1, H, W), sp[i], img)) for i in range(2)]
        return m.match_lines(o[0]["line_desc"], o[0]["mat_klines2sublines"], o[1]["line_desc"], o[1]["mat_klines2sublines"], 0.8)

    with torch.no_grad():
        for _ in range(20):
            pair()
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(200):
            pair()
        pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
