"""Shader clock and power while one GEMM shape runs back to back (rocm-smi sampled from a side thread).
usage: python tools/gemm_clock_probe.py bf16x6 8192 4096 4096 [seconds]"""
import os, subprocess, sys, threading, time, re
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from workloads import synth
from linetr_amd.engine import Engine
mode = sys.argv[1]; M, N, K = map(int, sys.argv[2:5]); secs = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
eng = Engine(synth.make_state_dict(0), 'cuda:0')
A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda')
samples = []; stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(['rocm-smi', '-d', '0', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
            s = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', o); p = re.search(r'Power \(W\): ([\d.]+)', o)
            samples.append((int(s.group(1)) if s else None, float(p.group(1)) if p else None))
        except Exception as e:
            samples.append((None, None))
t = threading.Thread(target=poll); 
def run(label, fn):
    global stop, samples
    samples = []; stop = False
    th = threading.Thread(target=poll); th.start()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    dt = (time.perf_counter() - t0) / n
    stop = True; th.join()
    clk = [c for c, _ in samples[1:] if c]; pw = [p for _, p in samples[1:] if p]
    import statistics
    tail_c = clk[len(clk)//2:] or [0]; tail_p = pw[len(pw)//2:] or [0]
    print(f"{os.environ.get('LABEL', label)}: {dt*1e6:.1f} us, {2*M*N*K/dt/1e12:.1f} TF | sclk {statistics.median(tail_c):.0f} MHz | power {statistics.median(tail_p):.0f} W (settled half of {len(clk)} samples)", flush=True)
for md in ([mode] if mode != "all" else ["f32", "bf16x6", "bf16x3"]):
    eng.set_precision(md)
    eng.debug_gemm(A, W, cache_weights=True)
    run(f"{md} {M}x{N}x{K}", lambda: eng.debug_gemm(A, W, cache_weights=True))
if not os.environ.get("NO_TORCH_MM"):
    run("torch.mm bf16", (lambda a, w: (lambda: torch.mm(a, w.t())))(A.bfloat16(), W.bfloat16()))
