"""Build the gfx950 shared library in-tree:  python -m linetr_amd.build [--force] [--experiments | --all]

Default = the product library only.  --experiments additionally builds experiments/liblinetr_hip_experiments.so: the product sources
with -DLINETR_EXPERIMENTS plus the measured-and-rejected kernels of experiments/csrc/ (never loaded by the package or by `pytest -m gpu`).

hipcc cross-compiles for gfx950 without a GPU present; the resulting linetr_amd/csrc/liblinetr_hip.so travels to the
GPU box with the repo snapshot.  The library is a handful of translation units (csrc/linetr_*.hip, see lt_handle.h);
each is compiled to an object under csrc/build/ (in parallel, re-compiled only when one of the files it includes
changed: hipcc's -MD dependency files) and the objects are linked into the .so."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
OUT = os.path.join(CSRC, "liblinetr_hip.so")
XDIR = os.path.abspath(os.path.join(HERE, "..", "experiments"))
XSRC = os.path.join(XDIR, "csrc")                               # kernels that were measured and not shipped + their host unit
OUT_X = os.path.join(XDIR, "liblinetr_hip_experiments.so")      # product sources + XSRC, -DLINETR_EXPERIMENTS (tools/, `pytest -m experiments`)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-undefined-inline"]
LFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def units(experiments: bool = False):
    return sorted(glob.glob(os.path.join(CSRC, "linetr_*.hip")) + (glob.glob(os.path.join(XSRC, "linetr_*.hip")) if experiments else []))


def sources(experiments: bool = False):
    return sorted(units(experiments) + glob.glob(os.path.join(CSRC, "*.h")) + (glob.glob(os.path.join(XSRC, "*.h")) if experiments else [])
                  + [os.path.join(HERE, "..", "include", "linetr_hip.h")])


def _deps(dfile):
    """prerequisites listed in a make-style dependency file"""
    try:
        txt = open(dfile).read()
    except OSError:
        return None
    txt = txt.replace("\\\n", " ")
    return [t for t in txt.split(":", 1)[1].split() if t] if ":" in txt else None


def _object_stale(obj, dfile):
    if not os.path.exists(obj):
        return True
    deps = _deps(dfile)
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    for d in deps:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if not os.path.exists(p) or os.path.getmtime(p) > t:
            return True
    return False


def up_to_date(out=OUT) -> bool:
    if not os.path.exists(out):
        return False
    t = os.path.getmtime(out)
    return all(os.path.getmtime(s) <= t for s in sources(out == OUT_X))


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    """experiments=False: the product library.  experiments=True: liblinetr_hip_experiments.so (tuning switches and the
    measured-and-rejected kernels compiled in)."""
    out = OUT_X if experiments else OUT
    if not force and up_to_date(out):
        return out
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    tag = "x" if experiments else "p"
    defs = ["-DLINETR_EXPERIMENTS", "-I", XSRC, "-I", CSRC] if experiments else []
    jobs, objs = [], []
    for src in units(experiments):
        stem = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(OBJ, f"{stem}.{tag}.o")
        dfile = obj[:-2] + ".d"
        objs.append(obj)
        if force or _object_stale(obj, dfile):
            jobs.append([hipcc, *CFLAGS, *defs, "-MD", "-MF", dfile, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, *LFLAGS, "-o", out, *objs])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--experiments" in sys.argv or "--all" in sys.argv:
        print(build(force="--force" in sys.argv, experiments=True))
