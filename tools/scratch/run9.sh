cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/pipeline_clock_probe.py 5 2>/dev/null | tee gpurun_out/r06_pipeline_clock_probe.txt
