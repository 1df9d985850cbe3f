# usage: bash tools/pmc_gemm.sh <mode> <M> <N> <K>   -- PMC counters of the GEMM microbench (separate passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MODE=${1:-bf16x6}; M=${2:-65536}; N=${3:-1024}; K=${4:-1024}
run() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -o p -- python tools/gemm_one.py $MODE $M $N $K > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_$name/*counter_collection.csv')
if not f: print('$name: no file'); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'gemm' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print('$name', k, '%.4g' % (sum(v)/len(v)), len(v))
PY
  rm -rf gpurun_out/pmc_$name
}
run a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run b TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
run c TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run d SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES
run e GRBM_GUI_ACTIVE
run f TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum TD_BUSY_avr
