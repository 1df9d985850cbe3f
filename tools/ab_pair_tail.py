"""Same-box A/B of Matching.forward with and without the one-call matching tail (linetr_pair_tail; LINETR_NO_PAIR_TAIL=1 = the four
separate calls it replaces), alternating blocks of calls in ONE process:   python tools/ab_pair_tail.py   (on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
sys.argv = ["bench.py"]
import bench
from workloads import synth
from linetr_amd import matching as M
from linetr_amd.engine import Engine
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
eng = Engine(synth.calibrated_state_dict(), dev, image_shape=[480, 640])
lines, dd, nhwc, ds, hw, T = bench.make_inputs("cfg2", 1, 0, dev, eng)
H, W = hw
g = torch.Generator(device=dev).manual_seed(0)
def sp_out(i):
    de = torch.nn.functional.normalize(torch.randn(256, 512, device=dev, generator=g), dim=0)
    kp = torch.rand(512, 2, device=dev, generator=g) * 400
    return {"keypoints": [kp], "scores": (torch.rand(512, device=dev, generator=g),), "descriptors": [de], "dense_descriptor": dd[i:i+1], "dense_score": ds[i:i+1]}
outs = [sp_out(0), sp_out(1)]
m = M.Matching({"auto_min_length": False, "linetransformer": {"mode": "train", "max_tokens": T, "image_shape": [H, W], "min_length": 16, "token_distance": 8, "remove_borders": 8, "max_keylines": -1, "nn_threshold": 0.8}},
               superpoint=bench._StubSuperPoint(outs), lsd=bench._StubLSD([synth.array_to_keylines(l) for l in lines]))
m.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()))
m = m.eval().to(dev)
img = torch.zeros(1, 1, H, W, device=dev)
def block(n):
    t0 = time.perf_counter()
    for _ in range(n): m({"image0": img, "image1": img})
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for _ in range(50): m({"image0": img, "image1": img})
res = {"tail": [], "separate": []}
for rep in range(6):
    for mode in ("tail", "separate"):
        if mode == "separate": os.environ["LINETR_NO_PAIR_TAIL"] = "1"
        else: os.environ.pop("LINETR_NO_PAIR_TAIL", None)
        block(20)
        res[mode].append(block(200))
os.environ.pop("LINETR_NO_PAIR_TAIL", None)
for k, v in res.items():
    print(f"{k:9s} median {np.median(v):.4f} ms  (blocks: {' '.join(f'{x:.3f}' for x in v)})")
