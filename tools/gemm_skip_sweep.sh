#!/bin/bash
# Which component of the pipelined split-GEMM main loop costs wall time?  Builds one debug library per set of
# components compiled OUT of the loop (results are then wrong, only the time is of interest) and times each.
#   tools/gemm_skip_sweep.sh build      (here, no GPU)      tools/gemm_skip_sweep.sh run   (on the GPU box)
# mask bits: 1 every block fetches tile (0,0)   2 no global loads   4 no split + ds_write   8 no MFMA
#            16 no barrier   32 no fragment ds_reads   64 ds_write without the split VALU
cd "$(dirname "$0")/.."
masks=(${MASKS:-0 2 4 8 16 32 64 6 38 54 46})
if [ "$1" = build ]; then
  rm -f tools/liblinetr_skip*.so
  for n in "${masks[@]}"; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DLT_GEMM_DEBUG_SKIP=$n -Iinclude \
      -o tools/liblinetr_skip$n.so linetr_amd/csrc/linetr_core.hip linetr_amd/csrc/linetr_net.hip linetr_amd/csrc/linetr_match.hip 2>&1 | grep -E " error|undefined" &
  done; wait
else
  for shape in ${SHAPES:-8192x4096x4096}; do for m in ${MODES:-bf16x6 bf16x3}; do for n in "${masks[@]}"; do
    LABEL="$m $shape skip-mask $n" LINETR_LIB=$PWD/tools/liblinetr_skip$n.so timeout 100 python tools/gemm_time_one.py $m ${shape//x/ } 2>&1 | grep TF
  done; done; done
fi
