"""Registers / LDS / residency of every kernel in the shipped library, from the code-object metadata (no GPU needed).
    python tools/resource_table.py [filter ...]
Compiles every translation unit linetr_amd/csrc/linetr_*.hip with -save-temps into a scratch directory and prints one line per kernel:
threads, VGPRs (incl. AGPRs), spilled VGPRs, static LDS bytes (dynamic LDS is set by the launchers) and the waves per SIMD
the register count allows (512 registers per lane and SIMD, allocated in blocks of 8).  Behind DESIGN.md section 4.2."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    filters = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        asm = ""
        for src in sorted(glob.glob(os.path.join(ROOT, "linetr_amd", "csrc", "linetr_*.hip"))):
            stem = os.path.splitext(os.path.basename(src))[0]
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                            "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-save-temps", "-o", stem + ".o", src],
                           cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            asm += open(os.path.join(tmp, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    items = re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", asm, flags=re.S)
    names, rows = [], []
    for it in items:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, it).group(1)
        names.append(g("name"))
        rows.append((int(g("max_flat_workgroup_size")), int(g("vgpr_count")), int(g("vgpr_spill_count")),
                     int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size"))))
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    print(f"{'kernel':100s} threads  vgpr spill  staticLDS scratch waves/SIMD(regs)")
    for d, (thr, vg, sp, lds, scr) in sorted(zip(dem, rows)):
        if filters and not any(f in d for f in filters):
            continue
        waves = 512 // max(8, (vg + 7) // 8 * 8)
        print(f"{d[:100]:100s} {thr:7d} {vg:5d} {sp:5d} {lds:10d} {scr:7d} {min(waves, 8):5d}")


if __name__ == "__main__":
    main()
