# HBM traffic of every kernel of one bench run from rocprofv3 PMC counters (separate passes, as the MI355X guide
# prescribes).  Writes gpurun_out/pmc_traffic.json: per kernel average FETCH_SIZE / WRITE_SIZE (KiB) per launch.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_$c.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{c}/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k][c + "_KiB_avg"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KiB_avg", 0) * kv[1]["launches"])[:12]:
    print(k[:70], v)
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
