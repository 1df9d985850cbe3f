"""Build the gfx950 shared library in-tree:  python -m linetr_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU present; the resulting
linetr_amd/csrc/liblinetr_hip.so travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "liblinetr_hip.so")
OUT_X = os.path.join(CSRC, "liblinetr_hip_experiments.so")     # same sources, -DLINETR_EXPERIMENTS (tools/, experiment tests)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) +
                  [os.path.join(HERE, "..", "include", "linetr_hip.h")])


def up_to_date(out=OUT) -> bool:
    if not os.path.exists(out):
        return False
    t = os.path.getmtime(out)
    return all(os.path.getmtime(s) <= t for s in sources())


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    """experiments=False: the product library.  experiments=True: liblinetr_hip_experiments.so (tuning switches and the
    measured-and-rejected kernels compiled in)."""
    out = OUT_X if experiments else OUT
    if not force and up_to_date(out):
        return out
    cmd = [_hipcc(), *FLAGS, *(["-DLINETR_EXPERIMENTS"] if experiments else []), "-o", out, os.path.join(CSRC, "linetr_hip.hip")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--experiments" in sys.argv or "--all" in sys.argv:
        print(build(force="--force" in sys.argv, experiments=True))
