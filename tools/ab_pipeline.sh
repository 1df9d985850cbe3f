#!/bin/bash
# Same-box A/B of the cfg3 step: one stream (--pipeline 0) against the two-stream software pipeline (--pipeline 1), alternating.
#   gpurun -- 'bash tools/ab_pipeline.sh out_tag [reps]'
# each spec is "label:bench args:ENV=val,ENV2=val"
tag=${1:-ab}; reps=${2:-2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
specs=("serial:--pipeline=0:" "pipe:--pipeline=1:" ${AB_EXTRA_SPECS})
for r in $(seq 1 $reps); do
  for spec in "${specs[@]}"; do
    label="${spec%%:*}"; rest="${spec#*:}"; bargs="${rest%%:*}"; envs="${rest#*:}"
    ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done; IFS=' '
      timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-sub-workloads $bargs 2>gpurun_out/${tag}_${label}_$r.log | tail -1 > gpurun_out/${tag}_${label}_$r.json )
    python - gpurun_out/${tag}_${label}_$r.json "$label#$r" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"== {sys.argv[2]:16s} {d['value']/1e6:7.3f} M desc/s  {d['ms_per_step']:.4f} ms/step  median {d['ms_per_step_median']:.4f} p10 {d['ms_per_step_p10']:.4f} p90 {d['ms_per_step_p90']:.4f}  host {d['host_ms_per_step']:.3f}")
except Exception as e:
    print("== ", sys.argv[2], "FAILED", e)
PY
  done
done
