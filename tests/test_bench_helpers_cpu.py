"""CPU: the bookkeeping bench.py does around the measurements -- which kernel class is priced against which roofline, which
profiles/ file a counter may come from, and that every kernel class of the committed cfg3 profile resolves to a counter record
taken on that workload (a class added in csrc/ without its symbol in PMC_KERNEL_NAMES would silently report traffic = null)."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def _entry(name, ms=1.0, calls=3, flops=0.0, nbytes=3e9):
    return {"name": name, "calls": calls, "ms": ms, "flops": flops, "bytes": nbytes}


def test_split_precision_classes_are_priced_on_the_bf16_pipe(bench):
    # 3 launches x 1e12 fp32-equivalent flop in 3 ms = 1000 TF-eq; six bf16 products per fp32 product
    r = bench.roofline_of(_entry("gemm_bf16x6_128x256", ms=3.0, flops=3e12), 3, 10.0, "none")
    assert r["bound"] == "mfma" and r["peak"] == bench.BF16_MFMA_PEAK_TFLOPS and r["mfma_flops_per_algorithmic_flop"] == 6
    assert r["fp32_equivalent_tflops"] == pytest.approx(1000.0) and r["achieved"] == pytest.approx(6000.0)
    assert r["frac"] == pytest.approx(6000.0 / bench.BF16_MFMA_PEAK_TFLOPS, rel=1e-3)
    assert r["launches_per_step"] == 1 and r["share_of_gpu_time"] == pytest.approx(0.3)
    r3 = bench.roofline_of(_entry("gemm_f16x3_128x256", ms=3.0, flops=3e12), 3, 10.0, "none")
    assert r3["mfma_flops_per_algorithmic_flop"] == 3
    rf = bench.roofline_of(_entry("gemm_f32_128x128", ms=3.0, flops=3e11), 3, 10.0, "none")
    assert rf["peak"] == bench.FP32_MFMA_PEAK_TFLOPS and rf["achieved"] == pytest.approx(100.0)


def test_pooling_and_token_kernels_are_priced_against_hbm(bench):
    for name in bench.HBM_BOUND_CLASSES:
        r = bench.roofline_of(_entry(name, ms=3.0, flops=1e9, nbytes=12e9), 3, 10.0, "none")
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBS
        assert r["achieved"] == pytest.approx(4000.0)          # 12 GB of algorithmic bytes in 3 ms


def test_counters_are_only_attached_from_a_pass_of_the_same_workload(bench):
    assert bench.pmc_traffic("gemm_bf16x6_128x256", "no-such-workload") is None
    assert bench.pmc_mfma_busy("gemm_bf16x6_128x256", "no-such-workload") is None
    t = bench.pmc_traffic("gemm_bf16x6_128x256", "cfg3")
    assert t is not None and 5e7 < t < 5e8                      # ~116 MB per launch in the committed pass
    b = bench.pmc_mfma_busy("gemm_bf16x6_128x256", "cfg3")
    assert b and 0.0 < b["mfma_busy"] < 1.0 and b["mfma_busy_source"].endswith("_cfg3_gemm_pmc.json")


def test_every_kernel_class_of_the_committed_step_has_its_counters(bench):
    d = json.load(open(os.path.join(ROOT, "profiles", f"{bench.PROFILE_TAG}_bench.json")))
    no_counters_expected = {"tokenize", "line_fill"}            # latency-bound helpers, not in the table on purpose
    for name, rec in d["kernels"].items():
        if name in no_counters_expected:
            continue
        assert name in bench.PMC_KERNEL_NAMES, f"{name}: add its kernel symbol to PMC_KERNEL_NAMES"
        assert bench.pmc_traffic(name, "cfg3") is not None, f"{name}: no FETCH/WRITE record in profiles/ for cfg3"
    dom = d["roofline"]
    assert dom["kernel"] in d["kernels"] and dom["traffic"] and dom["mfma_busy"]
    assert all(o["traffic"] for o in dom["other_kernels"])


def test_socket_filling_cpu_baseline_adds_up_process_rates(bench):
    """cpu_baseline_socket: N independent oracle processes at once (here 2 x 1 thread for a second each); the reported rate is the sum
    of the processes' own rates and the record says how many processes and cores it used."""
    r = bench.cpu_baseline_socket("cfg2", 1, 2, seconds=1.0)
    assert r["processes"] == 2 and r["threads_per_process"] == 1 and r["cores"] == 2 and r["physical_cores"] == 2
    assert r["value"] > 100 and r["unit"] == "line-descriptors/s"
