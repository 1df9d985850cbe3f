cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -15
AB_EXTRA_SPECS="pipe_cut1:--pipeline=1:LINETR_PIPE_CUT=1 pipe_prio1:--pipeline=1:LINETR_PIPE_PRIO=1 pipe_prio2:--pipeline=1:LINETR_PIPE_PRIO=2" bash tools/ab_pipeline.sh r06a 2
