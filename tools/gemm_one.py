import torch, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from linetr_amd import synth
from linetr_amd.engine import Engine
mode = sys.argv[1]; M,N,K = map(int, sys.argv[2:5])
eng = Engine(synth.make_state_dict(0), 'cuda:0'); eng.set_precision(mode)
A = torch.randn(M,K,device='cuda'); W = torch.randn(N,K,device='cuda')
for _ in range(5): eng.debug_gemm(A,W,cache_weights=True)
torch.cuda.synchronize()
