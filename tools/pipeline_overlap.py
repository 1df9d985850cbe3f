#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of `bench.py --pipeline D` and reports what ran CONCURRENTLY with what: per kernel class the
share of its execution time during which a kernel of another queue was also executing (and which), the chip's idle time between the
first and the last dispatch of the steady state, and a text timeline of one steady-state step.

    rocprofv3 --kernel-trace --output-format csv -d out -o p -- python bench.py --steps 20 --no-cpu-baseline --no-sub-workloads
    python tools/pipeline_overlap.py out/**/p_kernel_trace.csv [--last-ms 40] > profiles/r06_cfg3_pipeline_overlap.txt
"""
import argparse
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"lt::", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^(]*)?", name)
    base = m.group(1) if m else name
    if base == "gemm_split_kernel":
        t = re.search(r"<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false)", name)
        if t:
            base += f"<{t.group(1)}x{t.group(2)}{'' if t.group(6) == 'true' else 's'}>"
    return base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--last-ms", type=float, default=30.0, help="analyse the last this-many ms of the trace's busiest queue set (steady state)")
    ap.add_argument("--timeline-ms", type=float, default=5.0)
    args = ap.parse_args()
    rows = []
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            ks = {k.lower(): v for k, v in r.items()}
            s = int(ks.get("start_timestamp") or ks.get("start"))
            e = int(ks.get("end_timestamp") or ks.get("end"))
            rows.append((s, e, ks.get("queue_id", "?"), short(ks.get("kernel_name", "?"))))
    rows.sort()
    # steady state of the headline step: the last window in which the pipeline's kernels (tok_mlp / sig_qkv_attn) run
    idx = [i for i, r in enumerate(rows) if r[3].startswith("sig_qkv_attn") or r[3].startswith("sig_attn")]
    if not idx:
        print("no signature-attention kernels in the trace")
        return 1
    # the longest run of dispatches without a > 2 ms gap that contains signature kernels: take the LAST --last-ms of the largest such run
    t_end = rows[idx[-1]][1]
    t0 = t_end - int(args.last_ms * 1e6)
    win = [r for r in rows if r[0] >= t0 and r[1] <= t_end]
    queues = sorted({r[2] for r in win})
    print(f"# {args.csv}")
    print(f"# window: last {args.last_ms} ms before the final signature kernel; {len(win)} dispatches on queues {queues}")
    # sweep line over start/end points
    ev = []
    for i, (s, e, q, n) in enumerate(win):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active = set()
    alone = collections.Counter()       # ns a class ran with no kernel of another queue
    withq = collections.defaultdict(collections.Counter)   # class -> other class -> ns
    total = collections.Counter()
    idle = 0
    conc_hist = collections.Counter()
    last_t = ev[0][0]
    for t, kind, i in ev:
        dt = t - last_t
        if dt > 0:
            conc_hist[len(active)] += dt
            if not active:
                idle += dt
            for a in active:
                na, qa = win[a][3], win[a][2]
                total[na] += dt
                others = {win[b][3] for b in active if win[b][2] != qa}
                if not others:
                    alone[na] += dt
                for o in others:
                    withq[na][o] += dt
        last_t = t
        if kind == 1:
            active.add(i)
        else:
            active.discard(i)
    span = ev[-1][0] - ev[0][0]
    print(f"# span {span / 1e6:.3f} ms; no kernel executing: {idle / 1e6:.3f} ms ({100 * idle / span:.1f} %)")
    print("# kernels executing at once: " + ", ".join(f"{k}: {100 * v / span:.1f} %" for k, v in sorted(conc_hist.items())))
    calls = collections.Counter(r[3] for r in win)
    print(f"\n{'kernel class':44s} {'calls':>6s} {'avg us':>8s} {'busy ms':>8s} {'overlapped':>10s}   mostly with")
    for n, tns in sorted(total.items(), key=lambda kv: -kv[1]):
        ov = 1 - alone[n] / tns
        top = ", ".join(f"{o} {100 * v / tns:.0f}%" for o, v in withq[n].most_common(3))
        print(f"{n:44s} {calls[n]:6d} {tns / calls[n] / 1e3:8.1f} {tns / 1e6:8.3f} {100 * ov:9.1f}%   {top}")
    # timeline: one line per dispatch of the last --timeline-ms, columns per queue
    tl0 = t_end - int(args.timeline_ms * 1e6)
    print(f"\n# timeline of the last {args.timeline_ms} ms (us relative to its start; one column per hardware queue)")
    cols = {q: i for i, q in enumerate(queues)}
    for s, e, q, n in win:
        if s < tl0:
            continue
        pad = " " * (46 * cols[q])
        print(f"{(s - tl0) / 1e3:9.1f} {(e - tl0) / 1e3:9.1f}  {pad}{n[:34]:34s} {(e - s) / 1e3:7.1f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
