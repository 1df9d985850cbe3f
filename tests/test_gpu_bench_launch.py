"""GPU: `python bench.py --gpus 2` with no launcher around it must start two ranks by itself and print ONE JSON line
for the two-rank job.  On the single-GPU test box both ranks share cuda:0 and the collective runs over gloo
(LINETR_BENCH_ONE_DEVICE / LINETR_BENCH_BACKEND test hooks): the plumbing is what is checked, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(extra, expect_failure=False, gpus=2, **more_env):
    env = dict(os.environ, LINETR_BENCH_ONE_DEVICE="1", LINETR_BENCH_BACKEND="gloo", **more_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
                        "--settle-s", "0.2", *extra], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert (p.returncode != 0) if expect_failure else (p.returncode == 0), p.stderr[-2000:]
    _run.returncode = p.returncode
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_default_workload_self_launch():
    d = _run(["--pairs", "4"])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["descriptors_per_step"] > 2 * 4 * 2 * 150      # both ranks' descriptors are counted
    assert d["gathered_rows_checked"] is True and d["gather_ms"] > 0 and d["collective_backend"] == "gloo"
    assert d["global_match"]["matches"] > 0
    # pre-flight: both ranks saw each other through the benchmark's own backend; one gather time per rank
    assert d["preflight"]["ranks_seen"] == 2 and d["preflight"]["backend"] == "gloo" and len(d["gather_ms_per_rank"]) == 2


def test_a_bad_gather_fails_loudly():
    """A rank that received rows differing from what their owner packed must end the run with a non-zero exit code and ONE JSON line
    that carries "error" and a null value -- never a normal-looking line (the failure is injected on the last rank)."""
    d = _run(["--pairs", "4"], expect_failure=True, LINETR_BENCH_INJECT="bad_gather")
    assert d["value"] is None and "gathered_rows_checked" in d["error"] and d["n_gpus"] == 2


def test_cfg4_self_launch():
    d = _run(["--workload", "cfg4", "--pairs-total", "16", "--pairs", "4"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"


# ---- world size 8: the only size the driver's scaling run ends on (one node, eight MI355X).  Same one-device / gloo plumbing mode,
# tiny workloads: eight ranks, pre-flight, uneven shards, the slab gather, global matching, the one-JSON-line contract.

def test_eight_ranks_default_workload():
    d = _run(["--pairs", "2"], gpus=8)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["descriptors_per_step"] > 8 * 2 * 2 * 150            # all eight ranks' descriptors are counted
    assert d["gathered_rows_checked"] is True and d["collective_backend"] == "gloo"
    assert d["preflight"]["ranks_seen"] == 8 and len(d["preflight"]["devices"]) == 8 and len(d["gather_ms_per_rank"]) == 8
    assert d["global_match"]["matches"] > 0 and "rank 1" in d["global_match"]["against"]
    assert "cpu_baseline" not in d               # (the CPU baseline is reported at N = 1 only, as the contract asks)


def test_eight_ranks_cfg4_uneven_shards():
    """19 pairs over 8 ranks: ranks 0-2 own 3 pairs, ranks 3-7 own 2 -- slab heights agreed by the MAX all-reduce, every rank's
    queries matched against candidates other ranks own."""
    d = _run(["--workload", "cfg4", "--pairs-total", "19", "--pairs", "2", "--candidates", "3"], gpus=8)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["config"]["pairs_total"] == 19 and d["config"]["collective"] == "all_gather(slab)"
    assert d["config"]["pair_matches_per_step"] == 19 * 3
    assert d["config"]["descriptors_per_step"] > 19 * 2 * 150
    assert d["preflight"]["ranks_seen"] == 8
    assert d["recall_vs_homography_rank0"]["matches"] > 0


def test_eight_ranks_a_bad_rank_fails_with_exit_code_3():
    d = _run(["--pairs", "2"], expect_failure=True, gpus=8, LINETR_BENCH_INJECT="bad_gather")
    assert d["value"] is None and "gathered_rows_checked" in d["error"] and d["n_gpus"] == 8
    assert _run.returncode != 0
