cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {  # label, bench args, env
  label=$1; bargs=$2; envs=$3
  ( [ -n "$envs" ] && export "$envs"
    timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-sub-workloads $bargs 2>gpurun_out/r06e_$label.log | tail -1 > gpurun_out/r06e_$label.json )
  python - gpurun_out/r06e_$label.json "$label" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"== {sys.argv[2]:22s} {d['value']/1e6:7.3f} M desc/s  {d['ms_per_step']:.4f} ms/step  median {d['ms_per_step_median']:.4f} p10 {d['ms_per_step_p10']:.4f} p90 {d['ms_per_step_p90']:.4f}  host {d['host_ms_per_step']:.3f}")
except Exception as e:
    print("== ", sys.argv[2], "FAILED", e)
PY
}
for wl in cfg3 cfg5 cfg2; do
run ${wl}_serial "--workload $wl --pipeline 0" ""
run ${wl}_d2 "--workload $wl --pipeline 2" ""
run ${wl}_d3 "--workload $wl --pipeline 3" ""
run ${wl}_d4 "--workload $wl --pipeline 4" ""
run ${wl}_d4_c3_6_8 "--workload $wl --pipeline 4" "LINETR_PIPE_CUTS=3,6,8"
run ${wl}_d3_c3_6 "--workload $wl --pipeline 3" "LINETR_PIPE_CUTS=3,6"
run ${wl}_d4_slot "--workload $wl --pipeline 4" "LINETR_PIPE_CUTS=slot"
done
run cfg5x4_serial "--workload cfg5 --pairs 32 --pipeline 0" ""
run cfg5x4_d3 "--workload cfg5 --pairs 32 --pipeline 3" ""
run cfg3_nhwc_serial "--workload cfg3 --dense-layout nhwc --pipeline 0" ""
run cfg3_nhwc_d3 "--workload cfg3 --dense-layout nhwc --pipeline 3" ""
