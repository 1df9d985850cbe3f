cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LINETR_LIB=$PWD/experiments/liblinetr_hip_experiments.so LINETR_PROFILE_SHAPES=1 timeout 300 python bench.py --workload cfg5 --pairs 8 --dense-layout nhwc --steps 10 --no-cpu-baseline --no-sub-workloads > gpurun_out/r04x_cfg5_shapes.json 2> gpurun_out/r04x_cfg5_shapes.log
python - gpurun_out/r04x_cfg5_shapes.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:30]:
    print(f"{v['ms']:8.4f} ms x{v['calls']:3d} {v['tflops']}  {k}")
PY
