"""Wall time and cProfile of `Matching.forward` (the call the reference's scripts make per pair: match_line_pairs.py:90, demo_LineTR.py:207)
through the models.* shim, with KeyLines and SuperPoint outputs injected.   python tools/matching_forward_hostprof.py   (on the GPU box)"""
import cProfile, pstats, sys, os, torch, numpy as np, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from workloads import synth
from linetr_amd.engine import Engine
dev = torch.device("cuda:0")
eng = Engine(synth.calibrated_state_dict(), dev)
lines, dd, nhwc, ds, hw, T = bench.make_inputs("cfg2", 1, 0, dev, eng)
from models.matching import Matching
H, W = hw
klines = [synth.array_to_keylines(l) for l in lines[:2]]
g = torch.Generator(device=dev).manual_seed(5)
def sp_out(i):
    kp = torch.rand(512, 2, device=dev, generator=g) * 400
    de = torch.nn.functional.normalize(torch.randn(256, 512, device=dev, generator=g), dim=0)
    return {"keypoints": [kp], "scores": (torch.rand(512, device=dev, generator=g),), "descriptors": [de], "dense_descriptor": dd[i:i+1], "dense_score": ds[i:i+1]}
m = Matching({"auto_min_length": False, "linetransformer": {"mode": "train", "max_tokens": T, "image_shape": [H, W], "min_length": 16, "token_distance": 8, "remove_borders": 8, "max_keylines": -1, "nn_threshold": 0.8}},
             superpoint=bench._StubSuperPoint([sp_out(0), sp_out(1)]), lsd=bench._StubLSD(klines))
m.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
m = m.to(dev).eval()
img = torch.zeros(1, 1, H, W, device=dev)
with torch.no_grad():
    for _ in range(20): m({"image0": img, "image1": img})
    torch.cuda.synchronize()
    ts=[]
    for _ in range(50):
        t0=time.perf_counter(); m({"image0": img, "image1": img}); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    print("median ms", np.median(ts)*1e3)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): m({"image0": img, "image1": img})
    pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22); st.sort_stats("cumulative").print_stats(16)
