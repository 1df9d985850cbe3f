cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {  # label, bench args, env
  label=$1; bargs=$2; envs=$3
  ( [ -n "$envs" ] && export "$envs"
    timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-sub-workloads $bargs 2>gpurun_out/r06i_$label.log | tail -1 > gpurun_out/r06i_$label.json )
  python - gpurun_out/r06i_$label.json "$label" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d["kernels"]
    print(f"== {sys.argv[2]:22s} {d['value']/1e6:7.3f} M desc/s  {d['ms_per_step']:.4f} ms/step  median {d['ms_per_step_median']:.4f} p10 {d['ms_per_step_p10']:.4f} p90 {d['ms_per_step_p90']:.4f} | one-stream gemm256 {k.get('gemm_bf16x6_128x256',{}).get('ms')} gemm128s {k.get('gemm_bf16x6_128x128s',{}).get('ms')}")
except Exception as e:
    print("== ", sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
run p3_base_$rep "--pipeline 3" ""
run p3_w1_$rep "--pipeline 3" "LINETR_PIPE_TILES=1"
run p3_w2_$rep "--pipeline 3" "LINETR_PIPE_TILES=2"
run p3_w12_$rep "--pipeline 3" "LINETR_PIPE_TILES=3"
done
run p0_base "--pipeline 0" ""
run p0_w12 "--pipeline 0" "LINETR_PIPE_TILES=3"
