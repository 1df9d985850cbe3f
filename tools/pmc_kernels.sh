#!/bin/bash
# MFMA-busy / issue-stall / LDS counters of every kernel of one bench step (rocprofv3 --pmc, counters only, own pass),
# plus the HBM traffic passes.  Writes gpurun_out/<tag>_<workload>_gemm_pmc.json and gpurun_out/<tag>_<workload>_pmc_traffic.json
# (the names bench.py looks up for the roofline's `traffic` / `mfma_busy`: counters of the SAME workload and binary only).
#   usage (on the GPU box, via gpurun): bash tools/pmc_kernels.sh r03 cfg3 [extra bench.py flags]
tag=${1:-rXX}; wl=${2:-cfg3}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BENCH="python bench.py --workload $wl --pipeline 0 --steps 2 --warmup 1 --settle-s 0 --no-cpu-baseline --no-sub-workloads $*"
pass() {  # name counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -o p -- $BENCH > gpurun_out/pmc_$name.log 2>&1
}
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass sq2 SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
python - "$tag" "$wl" <<'PY'
import collections, csv, glob, json, sys
tag, wl = sys.argv[1], sys.argv[2]
def load(name):
    f = glob.glob(f"gpurun_out/pmc_{name}/*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
out = {"workload": wl,
       "note": "averages per launch over one rocprofv3 --pmc pass of `bench.py --workload " + wl + " --steps 2`.  mfma_busy = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8): SQ_VALU_MFMA_BUSY_CYCLES is summed over all "
               "SIMDs (= 32 cycles x the number of 32x32x16 MFMAs, checked against the GEMM's 6*2MNK flops) and "
               "GRBM_GUI_ACTIVE over the 8 XCDs (GRBM_GUI_ACTIVE / 8 / kernel duration = 2.2 GHz).  SQ_WAVE_CYCLES and the "
               "SQ_WAIT_* / SQ_ACTIVE_* counters are in quad-cycles (MI355X_MICROARCH.md)", "kernels": {}}
for name in ("sq", "sq2"):
    for k, cs in load(name).items():
        rec = out["kernels"].setdefault(k, {})
        for c, v in cs.items():
            rec[c] = sum(v) / len(v)
        rec["launches"] = max(rec.get("launches", 0), max(len(v) for v in cs.values()))
for k, rec in out["kernels"].items():
    if rec.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in rec:
        rec["mfma_busy"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (rec["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        rec["cycles_per_xcd"] = rec["GRBM_GUI_ACTIVE"] / 8.0
    if rec.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in rec:
                rec[c + "_frac_of_wave_cycles"] = round(rec[c] / rec["SQ_WAVE_CYCLES"], 4)
    if rec.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_frac"] = round(rec.get("SQ_LDS_BANK_CONFLICT", 0.0) / rec["SQ_LDS_IDX_ACTIVE"], 4)
json.dump(out, open(f"gpurun_out/{tag}_{wl}_gemm_pmc.json", "w"), indent=1)
tr = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, cs in load(c).items():
        if c in cs:
            tr[k][c + "_KiB_avg"] = sum(cs[c]) / len(cs[c])
            tr[k]["launches"] = len(cs[c])
json.dump(tr, open(f"gpurun_out/{tag}_{wl}_pmc_traffic.json", "w"), indent=1)
top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1].get("launches", 0))[:8]
for k, v in top:
    print(k[:60], {a: v[a] for a in ("launches", "mfma_busy", "SQ_WAIT_INST_ANY_frac_of_wave_cycles", "SQ_WAIT_ANY_frac_of_wave_cycles", "lds_bank_conflict_frac") if a in v})
PY
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
