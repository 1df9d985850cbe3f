cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_cfg4.py tests/test_gpu_dropin.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -2
python bench.py --workload cfg2 --steps 50 --no-cpu-baseline --no-alt-precisions --no-sub-workloads 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'match_lat', d['pair_match_latency_ms'], 'pair_match_ms', d['pair_match_ms'])"
python bench.py --steps 10 --no-cpu-baseline --no-alt-precisions 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'pair_match_ms', d['pair_match_ms'], 'lat', d['pair_match_latency_ms'], d['pair_latency_sync_ms'])"
