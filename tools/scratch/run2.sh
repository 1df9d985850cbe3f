cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5
run() {  # label, bench args, env
  label=$1; bargs=$2; envs=$3
  ( [ -n "$envs" ] && export "$envs"
    timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-sub-workloads $bargs 2>gpurun_out/r06c_$label.log | tail -1 > gpurun_out/r06c_$label.json )
  python - gpurun_out/r06c_$label.json "$label" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"== {sys.argv[2]:22s} {d['value']/1e6:7.3f} M desc/s  {d['ms_per_step']:.4f} ms/step  median {d['ms_per_step_median']:.4f} p10 {d['ms_per_step_p10']:.4f} p90 {d['ms_per_step_p90']:.4f}  host {d['host_ms_per_step']:.3f}")
except Exception as e:
    print("== ", sys.argv[2], "FAILED", e)
PY
}
for wl in cfg3 cfg5; do
run ${wl}_serial "--workload $wl --pipeline 0" ""
run ${wl}_d2_c6 "--workload $wl --pipeline 2" "LINETR_PIPE_CUTS=6"
run ${wl}_d2_c7 "--workload $wl --pipeline 2" "LINETR_PIPE_CUTS=7"
run ${wl}_d2_c8 "--workload $wl --pipeline 2" "LINETR_PIPE_CUTS=8"
run ${wl}_d3_c2_6 "--workload $wl --pipeline 3" "LINETR_PIPE_CUTS=2,6"
run ${wl}_d3_c3_7 "--workload $wl --pipeline 3" "LINETR_PIPE_CUTS=3,7"
run ${wl}_d3_c4_7 "--workload $wl --pipeline 3" "LINETR_PIPE_CUTS=4,7"
run ${wl}_d3_c5_8 "--workload $wl --pipeline 3" "LINETR_PIPE_CUTS=5,8"
run ${wl}_d4_c2_5_8 "--workload $wl --pipeline 4" "LINETR_PIPE_CUTS=2,5,8"
run ${wl}_d4_c3_6_8 "--workload $wl --pipeline 4" "LINETR_PIPE_CUTS=3,6,8"
run ${wl}_d4_c4_6_8 "--workload $wl --pipeline 4" "LINETR_PIPE_CUTS=4,6,8"
run ${wl}_d3_slot "--workload $wl --pipeline 3" "LINETR_PIPE_CUTS=slot"
done
