cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.log; echo "bench rc=$?"; cut -c1-300 gpurun_out/r06_bench.json
