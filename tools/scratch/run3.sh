cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for spec in "cfg3 3 5,8" "cfg3 2 3" "cfg5 3 4,7"; do
  set -- $spec
  wl=$1; d=$2; cuts=$3; lab=${wl}_d${d}_c${cuts//,/_}
  rm -rf gpurun_out/trace_$lab
  LINETR_PIPE_CUTS=$cuts timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$lab -o p -- \
     python tools/pipeline_trace_run.py $wl $d 400 > gpurun_out/r06d_${lab}.log 2>&1
  f=$(find gpurun_out/trace_$lab -name "*kernel_trace.csv" | head -1)
  python tools/pipeline_overlap.py "$f" --last-ms 30 --timeline-ms 5 > gpurun_out/r06d_${lab}_overlap.txt
  head -30 gpurun_out/r06d_${lab}_overlap.txt | cut -c1-200
  rm -rf gpurun_out/trace_$lab
done
