for P in 32 16; do for mode in classic st; do
  if [ $mode = classic ]; then export LINETR_SIG_PATH=classic; else unset LINETR_SIG_PATH; fi
  python bench.py --pairs $P --steps 20 --warmup 3 --no-cpu-baseline --no-sub-workloads 2>/dev/null | tail -1 > gpurun_out/ab_p${P}_$mode.json
  python - $P $mode <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_p{sys.argv[1]}_{sys.argv[2]}.json"))
print(f"== pairs {sys.argv[1]} {sys.argv[2]}: {d['value']/1e6:.3f} M desc/s, {d['ms_per_step']:.3f} ms/step")
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:4]:
    print(f"   {k:28s} {v['calls']:3d} x  {v['ms']:.4f} ms  {v['tflops']}")
PY
done; done
