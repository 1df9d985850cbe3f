import json,sys,subprocess
out = subprocess.run([sys.executable,'bench.py','--steps','5','--warmup','2','--no-cpu-baseline']+sys.argv[1:],capture_output=True,text=True).stdout.strip().splitlines()[-1]
j = json.loads(out)
print('value', j['value'], 'ms/step', j['ms_per_step'], 'gpu ms', j['gpu_ms_per_step_profiled'], 'latency', j['pair_latency_ms'])
for k,v in j['kernels'].items(): print(f"{v['ms']:8.4f} ms  x{v['calls']:2d}  {str(v['tflops']):>6} TF  {k}")
