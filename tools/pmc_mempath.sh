#!/bin/bash
# Memory-path counters (texture addresser, vector L1, L2) of every kernel of one bench step: own rocprofv3 --pmc passes,
# counters only.  Writes gpurun_out/<tag>_mempath_pmc.json.
#   usage (on the GPU box, via gpurun): bash tools/pmc_mempath.sh r02 [extra bench.py flags]
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BENCH="python bench.py --steps 2 --warmup 1 --settle-s 0 --no-cpu-baseline --no-sub-workloads $*"
pass() {
  name=$1; shift
  timeout -k 5 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -o p -- $BENCH > gpurun_out/pmc_$name.log 2>&1
}
# at most two counters of a block per pass: a wider request fails with "exceeds the capabilities of the hardware"
pass ta TA_BUSY_avr GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
pass tcc2 TCC_REQ_sum TCC_BUSY_avr GRBM_GUI_ACTIVE
pass vmem SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
python - "$tag" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
out = {"note": "averages per launch, one rocprofv3 --pmc pass per counter group over `bench.py --steps 2` (cfg3); *_sum counters are "
               "summed over the chip, GRBM_GUI_ACTIVE over the 8 XCDs; SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM_RD = average "
               "cycles a vector load is outstanding", "kernels": {}}
for name in ("ta", "tcc", "tcc2", "vmem"):
    f = glob.glob(f"gpurun_out/pmc_{name}/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if not k.startswith(("void lt::", "lt::")):
            continue
        rec = out["kernels"].setdefault(k, {})
        for c, v in cs.items():
            rec[c] = sum(v) / len(v)
        rec["launches"] = max(rec.get("launches", 0), max(len(v) for v in cs.values()))
for k, rec in out["kernels"].items():
    cyc = rec.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc:
        rec["cycles_per_xcd"] = cyc
        if "TCC_REQ_sum" in rec:
            rec["l2_req_bytes_per_cycle_per_cu"] = round(rec["TCC_REQ_sum"] * 128.0 / cyc / 256.0, 2)   # upper bound: 128 B per request
            rec["l2_hit_rate"] = round(rec["TCC_HIT_sum"] / max(rec["TCC_HIT_sum"] + rec["TCC_MISS_sum"], 1.0), 4)
    if rec.get("SQ_INSTS_VMEM_RD"):
        rec["vmem_rd_latency_cycles"] = round(rec.get("SQ_INST_LEVEL_VMEM", 0.0) / rec["SQ_INSTS_VMEM_RD"], 1)
    if rec.get("SQ_INSTS_LDS"):
        rec["lds_latency_cycles"] = round(rec.get("SQ_INST_LEVEL_LDS", 0.0) / rec["SQ_INSTS_LDS"], 1)
json.dump(out, open(f"gpurun_out/{tag}_mempath_pmc.json", "w"), indent=1)
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1].get("launches", 0))[:7]
for k, v in top:
    print(k[:70], {a: v[a] for a in ("launches", "TA_BUSY_avr", "TCC_BUSY_avr", "l2_hit_rate", "l2_req_bytes_per_cycle_per_cu",
                                      "vmem_rd_latency_cycles", "lds_latency_cycles") if a in v})
PY
