#!/bin/bash
# One gpurun call = every measurement of a round, stage by stage (a failing stage does not stop the next one).
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a "tests bench shapes prof pmc cfg4 two"'
tag=${1:-rXX}; stages=${2:-"tests bench"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/$tag
has() { [[ " $stages " == *" $1 "* ]]; }
if has tests; then
  timeout 900 python -m pytest tests -m gpu -q --maxfail=12 > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${O}_pytest.log
fi
if has xtests; then   # the measured-and-rejected kernels (experiments/): their own marker, their own library
  timeout 600 python -m pytest experiments -m experiments -q > ${O}_pytest_experiments.log 2>&1; echo "experiments rc=$?"; tail -3 ${O}_pytest_experiments.log
fi
if has bench; then
  timeout 600 python bench.py --steps 20 --warmup 5 > ${O}_bench.json 2> ${O}_bench.log; echo "bench rc=$?"; cut -c1-900 ${O}_bench.json
fi
if has shapes; then   # per-GEMM-shape HIP-event profile of the cfg3 step
  LINETR_LIB=$PWD/experiments/liblinetr_hip_experiments.so LINETR_PROFILE_SHAPES=1 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-sub-workloads > ${O}_bench_shapes.json 2> ${O}_bench_shapes.log
  python - ${O}_bench_shapes.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:24]:
    print(f"{v['ms']:8.4f} ms x{v['calls']:3d} {v['tflops']}  {k}")
PY
fi
if has hostprof; then   # where the host side of a cfg3 step goes (cProfile over 30 steps)
  timeout 300 python - > ${O}_hostprof.txt 2>&1 <<'PY'
import cProfile, pstats, sys, torch
sys.argv = ["bench.py"]
import bench
from workloads import synth
from linetr_amd.engine import Engine
dev = torch.device("cuda:0")
eng = Engine(synth.calibrated_state_dict(), dev, image_shape=[480, 640])
lines, dd, nhwc, ds, hw, T = bench.make_inputs("cfg3", 64, 0, dev, eng)
pipe = bench.Pipeline(eng, lines, nhwc, ds, hw, T, 1, 64)      # (one stream: the host side of a step, not the pipeline's waits)
for _ in range(50):
    pipe.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    pipe.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
# the single-pair call (cfg2): what the host does before / between the 44 launches
lines, dd, nhwc, ds, hw, T = bench.make_inputs("cfg2", 1, 0, dev, eng)
one = bench.Pipeline(eng, lines, nhwc, ds, hw, T, 1, 1)
for _ in range(50):
    one.step(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    one.step(); torch.cuda.synchronize()
pr.disable()
print("==== cfg2, 100 synchronous steps")
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
PY
  grep -A26 "==== cfg2" ${O}_hostprof.txt | cut -c1-150
fi
if has prof; then     # rocprofv3 kernel stats, one run per workload.  --pipeline 0: every kernel ALONE on the chip, like the HIP-event profile
  # bench.py's roofline block is made from (under the describe pipeline two or three kernels share the CUs and every duration stretches;
  # that schedule has its own stage below: trace)
  for wl in cfg3 cfg2 cfg5; do
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$wl -o p -- \
      python bench.py --workload $wl --pipeline 0 --steps 5 --warmup 2 --settle-s 0.5 --no-cpu-baseline --no-sub-workloads \
      > ${O}_${wl}_bench_under_rocprof.json 2> ${O}_${wl}_prof.log
    f=$(ls gpurun_out/prof_${tag}_$wl/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && cp "$f" ${O}_${wl}_kernel_stats.csv && head -7 ${O}_${wl}_kernel_stats.csv | cut -c1-150
    rm -rf gpurun_out/prof_${tag}_$wl
  done
fi
if has trace; then    # the describe pipeline's steady state: which kernels execute concurrently (rocprofv3 --kernel-trace + tools/pipeline_overlap.py)
  for spec in "cfg3 3" "cfg5 3" "cfg3 2"; do
    set -- $spec; wl=$1; d=$2
    rm -rf gpurun_out/trace_$wl
    timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$wl -o p -- python tools/pipeline_trace_run.py $wl $d 400 > ${O}_${wl}_d${d}_trace.log 2>&1
    f=$(find gpurun_out/trace_$wl -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/pipeline_overlap.py "$f" --last-ms 30 --timeline-ms 6 > ${O}_${wl}_pipeline_overlap_d${d}.txt && head -16 ${O}_${wl}_pipeline_overlap_d${d}.txt | cut -c1-170
    rm -rf gpurun_out/trace_$wl
  done
fi
if has sweep8; then   # bench.py --gpus {1,2,4,8} for both sharded workloads with every rank on THIS device and gloo as the collective: plumbing only
  LINETR_BENCH_ONE_DEVICE=1 LINETR_BENCH_BACKEND=gloo SWEEP_PAIRS=8 CFG4_PAIRS=64 timeout 1500 bash tools/scale_sweep.sh gpurun_out/sweep_$tag 1 2 4 8 > ${O}_scale_sweep_one_device_gloo.txt 2>&1
  echo "sweep8 rc=$?"; cat ${O}_scale_sweep_one_device_gloo.txt
fi
if has soak; then     # parity soaks against the oracle / the reference-made asset fixture
  { timeout 900 python tools/parity_soak.py 96 cfg3; timeout 900 python tools/parity_soak.py 16 cfg5; timeout 300 python tools/parity_soak.py asset; } > ${O}_parity_soak.txt 2>/dev/null
  cat ${O}_parity_soak.txt
  timeout 900 python tools/tokenizer_fuzz_soak.py > ${O}_tokenizer_soak.txt 2>/dev/null; tail -1 ${O}_tokenizer_soak.txt
fi
if has pmc; then      # counters per workload: bench.py only attaches counters taken on the same workload
  for wl in cfg3 cfg2 cfg5; do bash tools/pmc_kernels.sh $tag $wl; done
fi
if has cfg4; then
  timeout 600 python bench.py --workload cfg4 --steps 3 --warmup 1 --settle-s 1 > ${O}_cfg4.json 2> ${O}_cfg4.log; echo "cfg4 rc=$?"; cut -c1-1200 ${O}_cfg4.json
  timeout 300 python bench.py --workload cfg4 --pairs-total 128 --homography-strength 0.05 --steps 3 --warmup 1 --settle-s 0.5 > ${O}_cfg4_mild.json 2> ${O}_cfg4_mild.log; cut -c1-400 ${O}_cfg4_mild.json
fi
if has two; then      # the N>1 code path on one device (gloo carries the collective): correctness of the plumbing only
  LINETR_BENCH_ONE_DEVICE=1 LINETR_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --settle-s 0.5 > ${O}_two_ranks.json 2> ${O}_two_ranks.log
  echo "two rc=$?"; cut -c1-600 ${O}_two_ranks.json; tail -3 ${O}_two_ranks.log
  LINETR_BENCH_ONE_DEVICE=1 LINETR_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload cfg4 --pairs-total 64 --pairs 16 --steps 2 --warmup 1 --settle-s 0.3 \
    > ${O}_two_ranks_cfg4.json 2> ${O}_two_ranks_cfg4.log
  echo "two cfg4 rc=$?"; cut -c1-600 ${O}_two_ranks_cfg4.json; tail -3 ${O}_two_ranks_cfg4.log
fi
