#!/bin/bash
# 1 -> 8 GPU sweep of both sharded workloads (BASELINE.json cfg3 weak-scaled, cfg4 strong-scaled), one line per N with
# compute / gather / match milliseconds, and the checks a scaling run must pass before its numbers mean anything:
#   gathered_rows_checked  every rank's slab arrived intact (unit-norm rows, own rows bit-identical)
#   global_match.matches   matching out of the GATHERED buffer found matches against another rank's descriptors
#   collective_backend     "nccl" (= RCCL on ROCm) -- unless the sweep runs in the one-device test mode
#
#   bash tools/scale_sweep.sh [out_dir] [N ...]            # on a multi-GPU node:   N defaults to "1 2 4 8"
#   LINETR_BENCH_ONE_DEVICE=1 LINETR_BENCH_BACKEND=gloo bash tools/scale_sweep.sh gpurun_out/sweep 1 2
#       (one-device test mode: every rank on cuda:0, gloo carries the collective -- plumbing only, the times mean nothing;
#        SWEEP_PAIRS=8 CFG4_PAIRS=64 keep an 8-rank rehearsal on one GPU short)
# bench.py starts its own ranks (`python bench.py --gpus N` re-executes itself under torch.distributed.run).
out=${1:-gpurun_out/scale_sweep}; shift
ns=${*:-"1 2 4 8"}
cd "$(dirname "$0")/.." && mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
rc=0
for n in $ns; do
  timeout 900 python bench.py --gpus $n ${SWEEP_PAIRS:+--pairs $SWEEP_PAIRS} --steps 10 --warmup 3 --settle-s 1 --no-cpu-baseline --no-sub-workloads \
      > "$out/cfg3_n$n.json" 2> "$out/cfg3_n$n.log" || { echo "cfg3 N=$n: bench.py failed (see $out/cfg3_n$n.log)"; rc=1; continue; }
  timeout 900 python bench.py --gpus $n --workload cfg4 --pairs-total ${CFG4_PAIRS:-1024} ${SWEEP_PAIRS:+--pairs $SWEEP_PAIRS} --steps 3 --warmup 1 --settle-s 1 \
      > "$out/cfg4_n$n.json" 2> "$out/cfg4_n$n.log" || { echo "cfg4 N=$n: bench.py failed (see $out/cfg4_n$n.log)"; rc=1; continue; }
done
python - "$out" $ns <<'PY' || rc=1
import json, os, sys
out, ns = sys.argv[1], [int(v) for v in sys.argv[2:]]
one_dev = bool(os.environ.get("LINETR_BENCH_ONE_DEVICE"))
bad = 0
def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None
base = {}
print(f"{'workload':8s} {'N':>2s} {'desc/s':>12s} {'ms/step':>9s} {'compute':>9s} {'gather':>8s} {'match':>8s}  eff   checks")
for wl in ("cfg3", "cfg4"):
    for n in ns:
        p = os.path.join(out, f"{wl}_n{n}.json")
        d = last_json(p) if os.path.exists(p) else None
        if d is None:
            print(f"{wl:8s} {n:2d}  -- no result"); bad += 1; continue
        if d.get("error"):           # r05: a run that failed its pre-flight / gathered-rows check prints ONE line with "error" and a null value
            print(f"{wl:8s} {n:2d}  -- {d['error']}"); bad += 1; continue
        checks = []
        if d["n_gpus"] != n:
            checks.append(f"n_gpus={d['n_gpus']}")
        if n > 1:
            want = os.environ.get("LINETR_BENCH_BACKEND", "nccl") if one_dev else "nccl"
            if wl == "cfg3":
                if d.get("collective_backend") != want:
                    checks.append(f"backend={d.get('collective_backend')}")
                if d.get("gathered_rows_checked") is not True:
                    checks.append("gathered rows NOT verified")
                if not (d.get("global_match") or {}).get("matches", 0) > 0:
                    checks.append("no global matches")
            else:
                if d["config"]["collective"] != "all_gather(slab)":
                    checks.append("no collective ran")
        if wl == "cfg4" and not d.get("recall_vs_homography_rank0", {}).get("matches", 0) > 0:
            checks.append("cfg4: no matches")
        bad += len(checks)
        if wl == "cfg3":
            comp, gat, mat = d["ms_per_step"], d.get("gather_ms") or 0.0, (d.get("global_match") or {}).get("ms_per_batch") or d["pair_match_ms"] * d["config"]["pairs_per_gpu"]
        else:
            comp, gat, mat = d["compute_ms"], d["gather_ms"], d["global_match_ms"]
        base.setdefault(wl, (n, d["value"]))
        n0, v0 = base[wl]
        eff = d["value"] / (v0 * n / n0)
        per_rank = d.get("gather_ms_per_rank")
        extra = f"  gather/rank {min(per_rank):.3f}-{max(per_rank):.3f} ms" if per_rank else ""
        print(f"{wl:8s} {n:2d} {d['value']:12.0f} {d['ms_per_step']:9.3f} {comp:9.3f} {gat:8.3f} {mat:8.3f}  {eff:4.2f}  {'ok' if not checks else '; '.join(checks)}{extra}")
sys.exit(1 if bad else 0)
PY
exit $rc
