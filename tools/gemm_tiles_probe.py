"""Tile table: every split-GEMM tile on a list of (M, N, K) shapes, one process per tile (LINETR_GEMM_TILE is read once).
    python tools/gemm_tiles_probe.py [bf16x6]            -> one line per (shape, tile): us and TF-equivalent
Used to set the dispatcher's thresholds (lt_gemm_split.h: split_tile_name)."""
import os
import subprocess
import sys

TILES = ["128x256", "64x256", "256x128", "128x128", "128x128s", "64x128", "64x64"]
SHAPES = [(9584, 256, 256), (9584, 256, 512), (9584, 512, 512), (9584, 768, 256), (9584, 1024, 256), (9584, 256, 1024),
          (4792, 256, 512), (4792, 512, 512), (4792, 768, 256), (25472, 256, 512), (25472, 512, 512), (25472, 768, 256),
          (2396, 512, 512), (2396, 768, 256), (1198, 512, 512), (1198, 768, 256), (291208, 256, 128), (25472, 1024, 256),
          (25472, 256, 256), (25472, 256, 1024), (25472, 256, 768)]

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    # tuning switches and the non-shipped kernels live in the experiments build of the library
    os.environ.setdefault("LINETR_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments", "liblinetr_hip_experiments.so"))
    from workloads import synth
    from linetr_amd.engine import Engine
    eng = Engine(synth.make_state_dict(0), "cuda:0")
    eng.set_precision(sys.argv[2])
    g = torch.Generator(device="cuda").manual_seed(1)
    keep = []       # debug_gemm(cache_weights=True) keys its split copy by the weight POINTER: never let one be reused
    for M, N, K in SHAPES:
        A = torch.randn(M, K, device="cuda", generator=g)
        W = torch.randn(N, K, device="cuda", generator=g)
        R = torch.randn(M, N, device="cuda", generator=g)
        keep.append(W)
        try:
            for _ in range(5):
                eng.debug_gemm(A, W, None, R, 1, cache_weights=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                eng.debug_gemm(A, W, None, R, 1, cache_weights=True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 30
            print(f"{M:6d} {N:5d} {K:5d}  {os.environ.get('LINETR_GEMM_TILE', 'auto'):8s} {us:8.1f} us {2 * M * N * K / us / 1e6:7.1f} TF", flush=True)
        except Exception as e:  # a tile that does not support the shape
            print(f"{M:6d} {N:5d} {K:5d}  {os.environ.get('LINETR_GEMM_TILE', 'auto'):8s} failed: {str(e)[:60]}", flush=True)
else:
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
    if len(sys.argv) > 2:
        TILES = sys.argv[2].split(",")
    for tile in ["auto"] + TILES:
        env = dict(os.environ)
        if tile != "auto":
            env["LINETR_GEMM_TILE"] = tile
        env["LINETR_NO_SMALL_GEMM"] = "1"
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], env=env)
