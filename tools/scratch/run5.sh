cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -3
for r in 1 2; do
for p in 0 3; do
  timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-sub-workloads --pipeline $p 2>/dev/null | tail -1 > gpurun_out/r06g_p${p}_$r.json
  python - gpurun_out/r06g_p${p}_$r.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"== {sys.argv[1]} {d['value']/1e6:7.3f} M desc/s  {d['ms_per_step']:.4f} ms/step  median {d['ms_per_step_median']:.4f} p10 {d['ms_per_step_p10']:.4f} p90 {d['ms_per_step_p90']:.4f}")
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:6]:
    print(f"   {k:28s} {v['calls']:3d} x  {v['ms']:.4f} ms  {v['tflops']}")
PY
done
done
bash tools/pmc_kernels.sh r06g cfg3 2>&1 | tail -9
