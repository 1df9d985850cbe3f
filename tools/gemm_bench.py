import torch, time, sys
sys.path.insert(0,'.'); sys.path.insert(0,'..')
from linetr_amd import synth
from linetr_amd.engine import Engine
eng = Engine(synth.make_state_dict(0), 'cuda:0')
def bench(M,N,K,reps=20):
    A = torch.randn(M,K,device='cuda'); W = torch.randn(N,K,device='cuda')
    Y = eng.debug_gemm(A,W)
    ref = A@W.t()
    err = (Y-ref).abs().max().item()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): eng.debug_gemm(A,W)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/reps
    t1=time.perf_counter()
    for _ in range(reps): torch.mm(A,W.t())
    torch.cuda.synchronize(); dt2=(time.perf_counter()-t1)/reps
    print(f"M={M:7d} N={N:5d} K={K:5d}  {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:6.1f} TF   torch.mm {2*M*N*K/dt2/1e12:6.1f} TF  err {err:.2e}")
for s in [(8192,4096,4096),(65536,1024,1024),(25472,512,512),(25472,768,256),(25472,256,256),(25472,256,512),(32768,256,256),(32768,512,512),(534912,256,128),(534912,256,256)]:
    bench(*s)
