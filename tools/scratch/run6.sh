cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r06h_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r06h_pytest.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.log; echo "bench rc=$?"; cut -c1-600 gpurun_out/r06h_bench.json; tail -5 gpurun_out/r06h_bench.log
