#!/bin/bash
# One-shot profile refresh on the GPU box (run through gpurun): rocprofv3 kernel stats of the default bench command,
# the PMC traffic passes, and a plain bench line.  Results land in gpurun_out/; copy what should be judged to profiles/.
#   gpurun --timeout 900 -- 'tools/profile_round.sh r01_v4'
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o p -- \
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-precisions > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/prof_$tag.log
f=$(ls gpurun_out/prof_$tag/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv && head -8 gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/prof_$tag
bash tools/pmc_traffic.sh --no-alt-precisions | tail -6
cp gpurun_out/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log
tail -c 600 gpurun_out/${tag}_bench.json
