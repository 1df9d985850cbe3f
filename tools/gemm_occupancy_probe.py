"""Is the split GEMM limited per CU or chip-wide?  Same tile work per block, different number of busy CUs."""
import torch, time, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
# tuning switches and the non-shipped kernels live in the experiments build of the library
os.environ.setdefault("LINETR_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiments", "liblinetr_hip_experiments.so"))
from workloads import synth
from linetr_amd.engine import Engine
eng = Engine(synth.make_state_dict(0), 'cuda:0'); eng.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'bf16x6')
N, K = 4096, 4096                      # 32 column tiles of 128
W = torch.randn(N, K, device='cuda')
for M in (256, 512, 1024, 2048, 4096, 8192):   # 32, 64, 128, 256, 512, 1024 blocks of 256x128
    A = torch.randn(M, K, device='cuda')
    for _ in range(3): eng.debug_gemm(A, W, cache_weights=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.debug_gemm(A, W, cache_weights=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    blocks = (M // 256) * (N // 128)
    print(f"M={M:5d} blocks={blocks:5d}  {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:6.1f} TF   per-block-round {dt*1e6/max(1,-(-blocks//256)):7.1f} us")
