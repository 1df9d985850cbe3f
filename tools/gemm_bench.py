import torch, time, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from workloads import synth
from linetr_amd.engine import Engine
eng = Engine(synth.make_state_dict(0), 'cuda:0')
keep=[]
def bench(M,N,K,reps=20):
    PAD = int(os.environ.get('PAD','0'))
    A = torch.randn(M,K+PAD,device='cuda')[:, :K]; W = torch.randn(N,K,device='cuda'); keep.append(W)
    Ybuf = torch.empty(M,N+PAD,device='cuda')[:, :N]
    ref = (A.double()@W.double().t()).float()
    line = f"M={M:7d} N={N:5d} K={K:5d} "
    for mode in os.environ.get('MODES', 'f32,bf16x6,bf16x3,f16x3').split(','):
        eng.set_precision(mode)
        Y = eng.debug_gemm(A,W,cache_weights=True,out=Ybuf)
        err = ((Y-ref).abs().max()/ref.abs().max()).item()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(reps): eng.debug_gemm(A,W,cache_weights=True,out=Ybuf)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/reps
        line += f"| {mode}: {dt*1e6:7.1f} us {2*M*N*K/dt/1e12:6.1f} TF err {err:.1e} "
    t1=time.perf_counter()
    for _ in range(reps): torch.mm(A,W.t())
    torch.cuda.synchronize(); dt2=(time.perf_counter()-t1)/reps
    print(line + f"| torch.mm {2*M*N*K/dt2/1e12:6.1f} TF", flush=True)
_env = os.environ.get('SHAPES')
_shapes = [tuple(map(int, t.split('x'))) for t in _env.split(',')] if _env else None
for s in _shapes or [(8192,4096,4096),(65536,1024,1024),(25472,512,512),(25472,768,256),(25472,256,256),(25472,256,512),(32768,512,512),(534912,256,128),(534912,128,64)]:
    bench(*s)
