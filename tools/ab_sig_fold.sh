#!/bin/bash
# Same-box A/B of the single-pair signature fold (SigLayer::Wnext: W2 + residual + the next q/k/v projection as one contraction):
#   gpurun -- 'bash tools/ab_sig_fold.sh'      (experiments build: LINETR_NO_SIG_FOLD=1 switches it off)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LINETR_LIB=$PWD/experiments/liblinetr_hip_experiments.so
for rep in 1 2 3; do for mode in fold nofold; do
  if [ $mode = nofold ]; then export LINETR_NO_SIG_FOLD=1; else unset LINETR_NO_SIG_FOLD; fi
  python bench.py --workload cfg2 --steps 200 --warmup 20 --settle-s 1 --no-cpu-baseline --no-sub-workloads 2>/dev/null | tail -1 > gpurun_out/ab_fold_$mode.json
  python - $mode <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_fold_{sys.argv[1]}.json"))
k = d["kernels"]
print(f"{sys.argv[1]:7s} {d['ms_per_step']:.4f} ms/step (median {d['ms_per_step_median']:.4f}), {sum(v['calls'] for v in k.values())} launches, small GEMMs {k['gemm_bf16x6_32x32k4']['calls']} x = {k['gemm_bf16x6_32x32k4']['ms']:.4f} ms")
PY
done; done
