#!/bin/bash
# debug build of the library with in-kernel phase stamps (see tools/gemm_phase_timing.py)
cd "$(dirname "$0")/.." && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ${LT_DEBUG_DEFS:--DLT_GEMM_TIMING -DLT_GEMM_DEBUG_FETCH} \
  -Iinclude -o tools/liblinetr_timing.so linetr_amd/csrc/linetr_hip.hip 2>&1 | grep -E " error|undefined" ; true
