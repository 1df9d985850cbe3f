cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|FAILED" gpurun_out/t.log | head
for v in merged plain; do
  if [ $v = plain ]; then export LINETR_NO_MERGED_QKV=1; fi
  python bench.py --workload cfg2 --steps 50 --no-cpu-baseline --no-alt-precisions --no-sub-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['ms_per_step_median'], d['pair_latency_sync_ms'] if 'pair_latency_sync_ms' in d else '', d['config']['descriptors_per_step'])"
done
unset LINETR_NO_MERGED_QKV
python bench.py --steps 10 --no-cpu-baseline --no-alt-precisions 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'cfg2', d['cfg2']['ms_per_step'], d['cfg2']['pair_latency_sync_ms'], d['cfg2']['pair_match_latency_ms'], d['cfg2']['launches_per_step'])"
