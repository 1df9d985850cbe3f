"""Where the wall time of one `Matching.forward` goes, without a profiler's overhead: the functions of the drop-in path are wrapped
with perf_counter accumulators (exclusive of nothing: nested entries are listed under their own names too).
    python tools/matching_forward_stages.py      (on the GPU box)"""
import os, sys, time, collections
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from workloads import synth
from linetr_amd import engine as E, line_process as LP, matching as M, nn_matcher as NM, line_transformer as LT

acc = collections.defaultdict(float)
def wrap(owner, name, label=None):
    f = getattr(owner, name)
    lab = label or name
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[lab] += time.perf_counter() - t0
    setattr(owner, name, g)

for n in ("describe", "prefilter", "pack_many", "match", "match_points", "to_host_async", "_upload_recs", "pair_tail"):
    wrap(E.Engine, n, "Engine." + n)
for n in ("lines_from_rows", "keylines_to_array", "remove_borders", "filter_by_length", "get_angles"):
    wrap(LP, n)
wrap(M, "match01_to_matrix")
wrap(M.Matching, "_describe_fused", "Matching._describe_fused")
wrap(M.Matching, "_queue_line_match", "Matching._queue_line_match")
wrap(LT.LineTransformer, "engine", "LineTransformer.engine (weight-version check)")
wrap(torch, "ones_like", "torch.ones_like (valid_mask)")
wrap(torch, "cat", "torch.cat (the two dense maps side by side)")
wrap(torch, "empty", "torch.empty")
from linetr_amd import _native as nat
for n in ("linetr_describe", "linetr_prefilter_batch", "linetr_match", "linetr_match_points", "linetr_pair_tail"):
    wrap(nat.lib(), n, "native " + n)

dev = torch.device("cuda:0")
eng = E.Engine(synth.calibrated_state_dict(), dev)
lines, dd, nhwc, ds, hw, T = bench.make_inputs("cfg2", 1, 0, dev, eng)
H, W = hw
klines = [synth.array_to_keylines(l) for l in lines[:2]]
g = torch.Generator(device=dev).manual_seed(5)
def sp_out(i):
    kp = torch.rand(512, 2, device=dev, generator=g) * 400
    de = torch.nn.functional.normalize(torch.randn(256, 512, device=dev, generator=g), dim=0)
    return {"keypoints": [kp], "scores": (torch.rand(512, device=dev, generator=g),), "descriptors": [de], "dense_descriptor": dd[i:i+1], "dense_score": ds[i:i+1]}
m = M.Matching({"auto_min_length": False, "linetransformer": {"mode": "train", "max_tokens": T, "image_shape": [H, W], "min_length": 16, "token_distance": 8, "remove_borders": 8, "max_keylines": -1, "nn_threshold": 0.8}},
               superpoint=bench._StubSuperPoint([sp_out(0), sp_out(1)]), lsd=bench._StubLSD(klines))
m.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
m = m.to(dev).eval()
img = torch.zeros(1, 1, H, W, device=dev)
R = 300
with torch.no_grad():
    for _ in range(30): m({"image0": img, "image1": img})
    torch.cuda.synchronize(); acc.clear()
    t0 = time.perf_counter()
    for _ in range(R): m({"image0": img, "image1": img})
    total = time.perf_counter() - t0
print(f"Matching.forward: {total / R * 1e3:.3f} ms per call (mean of {R})")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {v / R * 1e6:8.1f} us  {k}")
