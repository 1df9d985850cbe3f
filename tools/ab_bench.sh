#!/bin/bash
# Same-box A/B of bench.py under different environment settings:  tools/ab_bench.sh "NAME=ENV=1 ..." ...
# each argument is "label:VAR=val,VAR2=val2" (label alone = default environment); prints value, ms/step and the kernel table.
# Switches other than LINETR_PRECISION / LINETR_HOST_THREADS only exist in the experiments build: add
# LINETR_LIB=experiments/liblinetr_hip_experiments.so to the spec (and to the baseline, for a like-for-like A/B).
mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%:*}"; envs=""
  if [[ "$spec" == *:* ]]; then envs="${spec#*:}"; fi
  ( IFS=','; for kv in $envs; do export "$kv"; done
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sub-workloads 2>/dev/null | tail -1 > gpurun_out/ab_$label.json )
  python - "$label" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}.json"))
print(f"== {sys.argv[1]}: {d['value']/1e6:.3f} M desc/s, {d['ms_per_step']:.3f} ms/step (median {d['ms_per_step_median']:.3f})")
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
    print(f"   {k:28s} {v['calls']:3d} x  {v['ms']:.4f} ms  {v['tflops']}")
PY
done
