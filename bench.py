#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native LineTR hot path.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json / SURVEY.md section 8d): line-descriptors/sec = sub-line descriptors written to
`line_desc` / wall time of tokenise + forward (host pre-filter, H2D of the line records, tokeniser,
descriptor network; for N>1 also the single RCCL all-gather of the descriptors).  The dense maps are
already resident in HBM in the layout the repo's own producer (linetr_superpoint_heads: score map +
channel-last descriptor map) emits, detected lines are resident on the host.  A "step" is one pass of
that path over one batch of P synthetic image pairs per GPU (weak scaling: every rank gets its own P
pairs).  Default workload = cfg3 of BASELINE.json (64 pairs of 640x480, 200 lines -> 199 sub-lines x
21 tokens per image), the configuration the roofline is defined on.  The same JSON line carries
`cfg2` (single pair: latency) and `cfg5` (1280x960, 600 lines x 41 tokens) sub-objects, pair-match ms,
the step fed with the reference's NCHW map, and the CPU baseline.

--workload cfg4 runs BASELINE.json's fourth configuration instead: 1024 homography-augmented pairs,
STRONG-sharded round-robin over the ranks (pair p -> rank p mod R), ONE all-gather of the descriptor
slabs, then global matching of every local query image against candidates taken from the gathered set;
a step is the whole job and the JSON carries compute_ms / gather_ms / global_match_ms.

One JSON line is printed by rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from linetr_amd import parallel
from workloads import synth  # noqa: E402
from linetr_amd.engine import DescribePipeline, Engine  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA
HBM_PEAK_GBS = 8000.0
PROFILE_TAG = "r06"             # profiles/<tag>_<workload>_pmc_traffic.json, profiles/<tag>_<workload>_gemm_pmc.json

WORKLOADS = {
    # name: (H, W, lines/image, len_lo, len_hi, max_tokens, default pairs per GPU)
    "cfg2": (480, 640, 200, 17.0, 167.0, 21, 1),
    "cfg3": (480, 640, 200, 17.0, 167.0, 21, 64),
    "cfg4": (480, 640, 200, 17.0, 167.0, 21, 64),     # pairs per describe call; the job is --pairs-total pairs
    "cfg5": (960, 1280, 600, 40.0, 327.0, 41, 8),
}
LINE_CFG = dict(min_length=16, token_distance=8, remove_borders=8, max_keylines=-1, nn_threshold=0.8)


def algorithmic_flops_per_image(N, T, K=None):
    """F_img = 2*[N*T*108640 + N*108704 + N*(2*S*65536 + 2*65536 + 2*S*256 + 524288)
    + 7*(N*655360 + 512*N^2) + N*65536]."""
    S = T + 1
    return 2.0 * (N * T * 108640 + N * 108704 + N * (2 * S * 65536 + 2 * 65536 + 2 * S * 256 + 524288)
                  + 7 * (N * 655360 + 512 * N * N) + N * 65536)


def make_inputs(workload, pairs, rank, device, eng):
    """Synthetic detector output + dense maps of `pairs` image pairs.  The dense maps are generated in the reference's
    NCHW layout and passed once (outside any timed region) through the repo's producer layout, so that the default
    step is fed what FusedHeadSuperPoint feeds it; the NCHW map is kept for the secondary measurement."""
    H, W, n_lines, lo, hi, T, _ = WORKLOADS[workload]
    lines, dds, dss = [], [], []
    for p in range(pairs):
        gp = rank * pairs + p
        for side in (0, 1):
            seed = 1000 + 2 * gp + side if workload != "cfg2" else 11 + side + 2 * gp
            lines.append(synth.synth_lines(seed, n_lines, H, W, lo, hi))
            dd, ds = synth.synth_dense_maps(seed, H, W)
            dds.append(dd)
            dss.append(ds)
    dd = torch.cat(dds).to(device)
    ds = torch.cat(dss).to(device)
    nhwc = dd.permute(0, 2, 3, 1).contiguous()     # layout change only: values identical to the NCHW map
    return lines, dd, nhwc, ds, (H, W), T


class Pipeline:
    def __init__(self, eng, lines, dd, ds, hw, T, world, pairs, layout="nhwc", pipelined=0, dpipe=None):
        self.eng, self.lines, self.dd, self.ds, self.hw, self.T = eng, lines, dd, ds, hw, T
        self.world, self.pairs = world, pairs
        # pipelined: consecutive steps as a two-stream software pipeline (DescribePipeline): step() submits batch i and hands back
        # batch i - 1; drain() -- part of the barrier that closes every timed region -- joins the batch still in flight
        # (dpipe: a pipeline shared by several batches of one job -- cfg4)
        self.dpipe = dpipe if dpipe is not None else (DescribePipeline(eng, int(pipelined)) if pipelined else None)
        self.last_joined = None
        self.layout = layout
        self.n_img_cap = 2 * pairs
        self.packed = [None, None]      # double-buffered: the all-gather of step i overlaps the compute of step i+1
        self.pending = [None, None]
        self.slot = 0
        # the detector output of the batch as one host array + row offsets (input format of the batched API)
        self.offsets = np.zeros(len(lines) + 1, dtype=np.int32)
        np.cumsum([len(l) for l in lines], out=self.offsets[1:])
        self.cat = np.ascontiguousarray(np.concatenate(lines), dtype=np.float64)
        # rows of the all-gather slab: the batch's sub-line count (a long line has several sub-lines), known from the
        # host pre-filter alone
        self.rows_cap = max(int(self.prefilter_only()[2][-1]), 1)

    def prefilter_only(self):
        e, c = self.eng, LINE_CFG
        return e.prefilter(self.cat, self.hw[0], self.hw[1], remove_borders=c["remove_borders"], min_length=c["min_length"],
                           max_keylines=c["max_keylines"], token_distance=c["token_distance"], max_tokens=self.T,
                           offsets=self.offsets)

    def describe(self, dd=None, layout=None):
        e, c = self.eng, LINE_CFG
        return e.describe_lines(self.cat, self.offsets, self.dd if dd is None else dd, self.ds,
                                remove_borders=c["remove_borders"], min_length=c["min_length"],
                                max_keylines=c["max_keylines"], token_distance=c["token_distance"], max_tokens=self.T,
                                dense_layout=layout or self.layout)

    def pack(self, tb, ld, out=None):
        return parallel.pack_descriptors(ld, tb.cu_n, self.n_img_cap, self.rows_cap, out, cu_k=tb.cu_k,
                                         sub2line=tb.sub2line, d_cu_n=tb.extra.get("d_cu_n"), d_cu_k=tb.extra.get("d_cu_k"))

    def submit(self):
        e, c = self.eng, LINE_CFG
        return self.dpipe.submit(self.cat, self.offsets, self.dd, self.ds, remove_borders=c["remove_borders"],
                                 min_length=c["min_length"], max_keylines=c["max_keylines"], token_distance=c["token_distance"],
                                 max_tokens=self.T, dense_layout=self.layout)

    def step(self):
        if self.dpipe is not None:
            done = self.submit()
            if done is None:            # first step after a drain: nothing to hand back yet (the batch just submitted is in flight)
                return self.last_joined
            tb, ld = done
        else:
            tb, ld = self.describe()
        return self.after_describe(tb, ld)

    def after_describe(self, tb, ld):
        gathered = None
        if self.world > 1:
            s = self.slot
            if self.pending[s] is not None:            # the buffer we are about to overwrite: its gather must be done
                self.pending[s][0].wait()
            self.packed[s] = self.pack(tb, ld, self.packed[s])
            work, gathered = parallel.allgather_descriptors(self.packed[s], async_op=True)
            self.pending[s] = (work, gathered)
            self.slot ^= 1
        self.last_joined = (tb, ld, gathered)
        return self.last_joined

    def drain(self):
        """join the batch still in flight in the describe pipeline (its all-gather included) and wait for every outstanding
        all-gather (called before the closing barrier of a timed region)."""
        if self.dpipe is not None:
            for done in self.dpipe.drain():
                self.after_describe(*done)
        for i, p in enumerate(self.pending):
            if p is not None:
                p[0].wait()
                self.pending[i] = None

    def match_args(self, tb, ld):
        """image 2p vs image 2p+1 for every local pair, addressed in place (no gather of rows)."""
        cu_n, cu_k = tb.cu_n.astype(np.int64), tb.cu_k.astype(np.int64)
        n, k = np.diff(cu_n), np.diff(cu_k)
        dims = np.stack([n[0::2], k[0::2], n[1::2], k[1::2]], axis=1).astype(np.int32)
        return (ld, tb.sub2line, dims, cu_n[0:-1:2], cu_n[0:-1:2], cu_n[1::2], cu_n[1::2])


def settle(step_fn, sync_fn, min_s, max_s=8.0, window=5, tol=0.03, agree=None):
    """Steady state before anything is timed: at least `min_s` seconds of load AND the last three `window`-step
    averages within `tol` of each other (a fresh lease ramps its clocks for more than a second; r01's 0.15 s of
    warm-up left the driver's 0.11 s timed window inside that ramp).  With several ranks the steps contain
    collectives, so the stop decision is taken together (`agree`: every rank must be done).
    Returns the window history (ms per step)."""
    hist = []
    t_start = time.perf_counter()
    while True:
        sync_fn()
        t0 = time.perf_counter()
        for _ in range(window):
            step_fn()
        sync_fn()
        hist.append((time.perf_counter() - t0) / window * 1e3)
        el = time.perf_counter() - t_start
        done = el >= max_s or (el >= min_s and len(hist) >= 3 and
                               (max(hist[-3:]) - min(hist[-3:])) <= tol * min(hist[-3:]))
        if agree is not None:
            done = agree(done)
        if done:
            break
    return hist


def make_agree(dist, world, device):
    """all ranks stop settling in the same iteration: done only if every rank says so (one tiny all-reduce per window)."""
    if world <= 1:
        return None

    def agree(done):
        t = torch.tensor([1.0 if done else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)
    return agree


def producer_section(eng, H, W, n_img):
    """linetr_superpoint_heads (the dense-map producer) on raw head outputs of this batch's shape,
    HIP-event timed, beside the same maths in stock PyTorch on the GPU."""
    Hc, Wc = H // 8, W // 8
    g = torch.Generator(device=eng.device).manual_seed(3)
    sl = torch.randn(n_img, 65, Hc, Wc, device=eng.device, generator=g) * 2
    dr = torch.randn(n_img, 256, Hc, Wc, device=eng.device, generator=g)

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ms = timed(lambda: eng.superpoint_heads(sl, dr, nhwc=True, nchw=False))

    def torch_heads():
        p = torch.softmax(sl, 1)[:, :-1]
        p = p.permute(0, 2, 3, 1).reshape(n_img, Hc, Wc, 8, 8).permute(0, 1, 3, 2, 4).reshape(n_img, Hc * 8, Wc * 8)
        return p, torch.nn.functional.normalize(dr, p=2, dim=1)
    ms_torch = timed(torch_heads)
    nbytes = n_img * Hc * Wc * (256 * 2 + 65 + 64) * 4
    return {"kernel": "sp_desc_head + sp_score_head (linetr_superpoint_heads)", "images": n_img, "ms": round(ms, 4),
            "algorithmic_bytes": nbytes, "achieved_GBps": round(nbytes / ms / 1e6, 1), "hbm_peak_GBps": HBM_PEAK_GBS,
            "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4), "torch_same_ops_ms": round(ms_torch, 4)}


def timed_steps(step_fn, barrier_fn, steps, device):
    """EXACTLY `steps` steps between two barriers (+ device synchronisation) -- the contract's timed region -- with a
    HIP event behind every step (per-step device times without any host synchronisation inside the region) and the
    host time spent inside each step call."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = np.zeros(steps)
    barrier_fn()
    evs[0].record()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        h0 = time.perf_counter()
        last = step_fn()
        host[i] = time.perf_counter() - h0
        evs[i + 1].record()
    barrier_fn()
    elapsed = time.perf_counter() - t0
    per_step = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
    return elapsed, per_step, host * 1e3, last


def cpu_baseline(workload, budget_s=14.0):
    """The CPU oracle (a faithful port of the reference's as-executed PyTorch-CPU/NumPy path, incl. the per-line Python
    tokeniser loop and the full 22-token descriptive layer) timed on this box over a sweep of torch thread counts
    (1 thread and the physical cores; default-128-thread runs oversubscribe and were 3-4x slower).
    `value` is the best setting's descriptors / (tokenise + forward) seconds."""
    from oracle import linetr_oracle as O
    H, W, n_lines, lo, hi, T, _ = WORKLOADS[workload]
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    cfg = dict(LINE_CFG, max_tokens=T)
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = max(1, (os.cpu_count() or 2) // 2)
    default_threads = torch.get_num_threads()
    sweep = sorted({t for t in (1, 8, 16, 32, phys) if t <= max(phys, 1)})
    per_setting_budget = budget_s / len(sweep)
    rows, inputs = [], []
    for side in (0, 1):
        seed = 5000 + side
        rows_ = synth.synth_lines(seed, n_lines, H, W, lo, hi)
        dd, ds = synth.synth_dense_maps(seed, H, W)
        inputs.append((synth.array_to_keylines(rows_), dd, ds))
    with torch.no_grad():
        for nt in sweep:
            torch.set_num_threads(nt)
            n_desc, t_tok, t_fwd, t_match, pairs = 0, 0.0, 0.0, 0.0, 0
            t_start = time.perf_counter()
            while pairs < 1 or (time.perf_counter() - t_start < per_setting_budget and pairs < 6):
                outs = []
                for kl, dd, ds in inputs:
                    t0 = time.perf_counter()
                    out = O.preprocess(kl, (1, 1, H, W), dd, ds, cfg)
                    t1 = time.perf_counter()
                    out = O.forward(sd, out, (H, W))
                    t2 = time.perf_counter()
                    t_tok += t1 - t0
                    t_fwd += t2 - t1
                    n_desc += out["line_desc"].shape[2]
                    outs.append(out)
                t0 = time.perf_counter()
                O.match_lines(outs[0]["line_desc"], outs[1]["line_desc"], outs[0]["mat_klines2sublines"][0],
                              outs[1]["mat_klines2sublines"][0], 0.8)
                t_match += time.perf_counter() - t0
                pairs += 1
            rows.append({"threads": nt, "pairs": pairs, "desc_per_s": round(n_desc / (t_tok + t_fwd), 1),
                         "forward_only_desc_per_s": round(n_desc / t_fwd, 1),
                         "tokenise_ms_per_pair": round(t_tok / pairs * 1e3, 1), "forward_ms_per_pair": round(t_fwd / pairs * 1e3, 1),
                         "match_ms_per_pair": round(t_match / pairs * 1e3, 2)})
    torch.set_num_threads(default_threads)
    best = max(rows, key=lambda r: r["desc_per_s"])
    one = next(r for r in rows if r["threads"] == 1)
    return {
        "value": best["desc_per_s"], "unit": "line-descriptors/s", "cores": best["threads"], "kind": "port",
        "sample": f"thread sweep {sweep} x >=1 {workload}-shaped pair each (2 x {n_lines} lines), oracle tokenise (Python loop, "
                  f"1 thread) + forward; best = {best['threads']} threads; torch {torch.__version__} CPU, "
                  f"{phys} physical / {os.cpu_count()} logical cpus",
        "one_thread_value": one["desc_per_s"], "physical_cores": phys, "sweep": rows,
        "pair_match_ms": best["match_ms_per_pair"],
    }


def cpu_worker(workload, threads, seconds):
    """One process of the socket-filling CPU run (cpu_baseline_socket): the oracle's tokenise + forward on one workload-shaped pair, over
    and over for `seconds`, on `threads` torch threads.  Prints one JSON line."""
    from oracle import linetr_oracle as O
    torch.set_num_threads(threads)
    H, W, n_lines, lo, hi, T, _ = WORKLOADS[workload]
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    cfg = dict(LINE_CFG, max_tokens=T)
    inputs = []
    for side in (0, 1):
        rows_ = synth.synth_lines(5000 + side, n_lines, H, W, lo, hi)
        dd, ds = synth.synth_dense_maps(5000 + side, H, W)
        inputs.append((synth.array_to_keylines(rows_), dd, ds))
    n_desc, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while time.perf_counter() - t0 < seconds or n_desc == 0:
            for kl, dd, ds in inputs:
                out = O.forward(sd, O.preprocess(kl, (1, 1, H, W), dd, ds, cfg), (H, W))
                n_desc += out["line_desc"].shape[2]
    print(json.dumps({"n_desc": n_desc, "seconds": time.perf_counter() - t0, "threads": threads}), flush=True)


def cpu_baseline_socket(workload, threads, phys, seconds=8.0, max_procs=32):
    """The CPU column at its strongest on this host: N independent processes x the best thread count of the single-process sweep, N =
    physical cores / threads (capped), all running at once; the rates add up.  (One process cannot use the socket: the oracle's per-image
    GEMMs are small, its tokeniser is a Python loop.)  kind stays "port": it is the oracle that is timed."""
    import subprocess
    n = max(1, min(max_procs, phys // max(threads, 1)))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{workload},{threads},{seconds}"]
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(n)]
    rate, done = 0.0, 0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=seconds * 6 + 120)
            rec = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            rate += rec["n_desc"] / rec["seconds"]
            done += 1
        except Exception:
            pr.kill()
    return {"value": round(rate, 1), "unit": "line-descriptors/s", "processes": done, "threads_per_process": threads,
            "cores": done * threads, "physical_cores": phys, "seconds_per_process": seconds,
            "note": f"{done} independent processes x {threads} threads, each the oracle's tokenise + forward on one {workload}-shaped pair in a "
                    "loop; the sum of their rates"}


PMC_KERNEL_NAMES = {   # profile class -> kernel symbol prefix in profiles/<tag>_pmc_traffic.json (rocprofv3 --pmc run)
    "gemm_bf16x6_128x256": "void lt::gemm_split_kernel<128, 256, 2, 4, 3, true, 0",
    "gemm_bf16x3_128x256": "void lt::gemm_split_kernel<128, 256, 2, 4, 2, true, 0",
    "gemm_f16x3_128x256": "void lt::gemm_split_kernel<128, 256, 2, 4, 2, true, 1",
    "gemm_bf16x6_256x128": "void lt::gemm_split_kernel<256, 128, 4, 2, 3, true, 0",
    "gemm_bf16x3_256x128": "void lt::gemm_split_kernel<256, 128, 4, 2, 2, true, 0",
    "gemm_f16x3_256x128": "void lt::gemm_split_kernel<256, 128, 4, 2, 2, true, 1",
    "gemm_bf16x6_128x128s": "void lt::gemm_split_kernel<128, 128, 2, 4, 3, false, 0",
    "gemm_bf16x6_ws64x256": "lt::gemm_ws_kernel",
    "tok_mlp_bf16x6": "void lt::tok_mlp_kernel<true>",
    "line_mlp_bf16x6": "void lt::tok_mlp_kernel<false>",
    "pos_mlp_dual_bf16x6": "lt::tok_mlp_",          # tok_mlp_seq_kernel (large batch) / tok_mlp_dual_kernel (single pair)
    "gemm_bf16x6_64x64": "void lt::gemm_split_kernel<64, 64, 2, 2, 3, true, 0",
    "gemm_bf16x6_64x128": "void lt::gemm_split_kernel<64, 128, 2, 2, 3, true, 0",
    "gemm_bf16x6_32x32k4": "void lt::gemm_split_small_kernel<3, 0",
    "gemm_f32_128x128": "void lt::gemm_kernel<128, 128, 2, 2>",
    "sig_attn_bf16x6": "void lt::sig_attn_split_kernel<8>",
    "sig_qkv_attn_bf16x6": "lt::sig_qkv_attn_kernel",
    "cls_pool_online": "void lt::cls_pool_online_kernel<1>",
    "gemm_bf16x6_128x64": "void lt::gemm_split_kernel<128, 64, 4, 1, 3, true, 0",
    "nchw_to_nhwc": "lt::nchw_to_nhwc_kernel",
    "row_norm": "lt::row_norm_kernel",
    "sig_attn_bf16x6_small": "lt::sig_attn_small_kernel",
}


def _sym(kernel_name):
    """kernel symbol as rocprofv3 prints it, without the return type a template instantiation carries ("void lt::k<0>(...)" / "lt::k(...)")"""
    return kernel_name[5:] if kernel_name.startswith("void ") else kernel_name


def _profile_json(name, workload):
    """committed counter pass of THIS workload and THIS round's binary only (profiles/<tag>_<workload>_<name>.json);
    a pass of another workload or an older binary is not evidence for this line."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{workload}_{name}.json")
    if os.path.exists(path):
        return json.load(open(path)), os.path.relpath(path, ROOT)
    return None, None


def pmc_traffic(kernel_class, workload):
    """HBM-side bytes per launch of `kernel_class` from the committed rocprofv3 PMC pass (FETCH_SIZE/WRITE_SIZE in
    KiB, separate --pmc runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads
    on gfx950).  None if no PMC data is committed for this kernel and workload."""
    data, _ = _profile_json("pmc_traffic", workload)
    name = PMC_KERNEL_NAMES.get(kernel_class)
    if not name or not data:
        return None
    rec = next((v for k, v in data.items() if _sym(k).startswith(_sym(name))), None)
    if not rec or "FETCH_SIZE_KiB_avg" not in rec or "WRITE_SIZE_KiB_avg" not in rec:
        return None
    return (2.0 * rec["FETCH_SIZE_KiB_avg"] + rec["WRITE_SIZE_KiB_avg"]) * 1024.0


def pmc_mfma_busy(kernel_class, workload):
    """SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE) of the dominant kernel from the committed PMC
    pass of the same workload (profiles/<tag>_<workload>_gemm_pmc.json, written by tools/pmc_kernels.sh), or None."""
    data, src = _profile_json("gemm_pmc", workload)
    name = PMC_KERNEL_NAMES.get(kernel_class)
    if not name or not data:
        return None
    rec = next((v for k, v in data.get("kernels", {}).items() if _sym(k).startswith(_sym(name))), None)
    if not rec or "mfma_busy" not in rec:
        return None
    return {"mfma_busy": rec["mfma_busy"], "mfma_busy_source": src}


def mfma_terms(kernel_class):
    """bf16 / fp16 MFMA products executed per fp32 product of this kernel class (None: exact-fp32 MFMA or no MFMA)."""
    return 6 if "bf16x6" in kernel_class else 3 if ("bf16x3" in kernel_class or "f16x3" in kernel_class) else None


HBM_BOUND_CLASSES = ("cls_pool_online", "cls_pool_fused", "cls_pool", "tokenize", "nchw_to_nhwc", "sample_desc")   # their flops are VALU work


def roofline_of(dom, prof_steps, tot_ms, workload):
    """roofline object of the dominant kernel (largest summed HIP-event time over the profiled steps).
    For a split-precision MFMA kernel `frac` is the fraction of the pipe it actually executes on: the bf16 MFMA products it
    issues (6 or 3 per fp32 product) against the 2.5 PFLOP/s dense bf16 peak; the ratio to the fp32 matrix peak (what an
    exact-fp32 kernel could reach at most) is kept as `frac_of_fp32_matrix_peak`."""
    common = {"kernel": dom["name"], "avg_launch_us": round(dom["ms"] / dom["calls"] * 1e3, 2),
              "launches_per_step": dom["calls"] // prof_steps, "share_of_gpu_time": round(dom["ms"] / tot_ms, 3)}
    traffic = pmc_traffic(dom["name"], workload)
    if dom["flops"] > 0 and dom["name"] not in HBM_BOUND_CLASSES:
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12      # executed fp32-equivalent flops (2*M*N*K of each launch) / time
        terms = mfma_terms(dom["name"])
        if terms:
            r = {"bound": "mfma", "achieved": round(ach * terms, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(ach * terms / BF16_MFMA_PEAK_TFLOPS, 4),
                 "achieved_note": f"bf16 MFMA flops issued: {terms} products per fp32 product x 2MNK / time, against the dense bf16 peak",
                 "fp32_equivalent_tflops": round(ach, 2), "frac_of_fp32_matrix_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                 "mfma_flops_per_algorithmic_flop": terms}
        else:
            r = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "achieved_note": "exact-fp32 MFMA: 2MNK / time against the fp32 matrix peak"}
        r.update({"traffic": traffic, "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["calls"])})
        busy = pmc_mfma_busy(dom["name"], workload)
        if busy:
            r.update(busy)
    else:
        ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic}
    return {**common, **r}


def whole_step_executed(prof, prof_steps, ms_per_step):
    """flops the kernels of one step actually execute (sum of every launch's own 2MNK etc.) and the fraction of the pipe
    the whole step keeps busy; counts every MFMA class with its own product count."""
    gf, pipe = 0.0, 0.0
    for e in prof:
        f = e["flops"] / prof_steps
        gf += f
        t = mfma_terms(e["name"])
        pipe += f * t if t else 0.0
    sec = ms_per_step * 1e-3
    return {"executed_gflop_per_step": round(gf / 1e9, 1),
            "executed_tflops_fp32_equivalent": round(gf / sec / 1e12, 1),
            "bf16_mfma_tflops_issued": round(pipe / sec / 1e12, 1),
            "frac_of_bf16_mfma_peak": round(pipe / sec / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
            "note": "sum over the step's launches of the flops each kernel executes (2MNK per GEMM, 4 n^2 d per attention), "
                    "split-precision classes counted with their 6 (or 3) bf16 products, divided by the timed step"}


def profile_steps(eng, fn, prof_steps=3):
    """per-kernel-class HIP-event profile of `prof_steps` calls of fn; returns (entries sorted by time, total ms)."""
    eng.set_profiling(True)
    for _ in range(prof_steps):
        fn()
    torch.cuda.synchronize()
    prof = eng.get_profile()
    eng.set_profiling(False)
    prof.sort(key=lambda e: -e["ms"])
    return prof, sum(e["ms"] for e in prof)


def breakdown_of(prof, prof_steps):
    return {e["name"]: {"calls": e["calls"] // prof_steps, "ms": round(e["ms"] / prof_steps, 4),
                        "tflops": round(e["flops"] / max(e["ms"], 1e-9) / 1e9, 1) if e["flops"] else None} for e in prof}


def sub_workload(eng, name, device, settle_s, repeat=1, brief=False, pipelined=3):
    """cfg2 / cfg5 sub-object of the N = 1 line: the same step on another BASELINE.json configuration (own inputs,
    own short settle), with its dominant kernel.  repeat > 1: the workload's default batch repeated that many times
    (BASELINE.json names no batch size for cfg5; the default of 8 pairs leaves two thirds of the CUs without a GEMM tile).
    cfg5: `value` is the pipelined step like the headline's, the one-stream step sits beside it (`serial`); cfg2 (a single pair:
    latency is the metric) keeps the one-stream step as `value` and reports pairs described back to back through the pipeline as
    `pipelined`."""
    H, W, n_lines, lo, hi, T, pairs = WORKLOADS[name]
    lines, dd_nchw, nhwc, ds, hw, T = make_inputs(name, pairs, 0, device, eng)
    if pairs != 1:
        del dd_nchw
    if repeat > 1:
        lines, nhwc, ds, pairs = lines * repeat, nhwc.repeat(repeat, 1, 1, 1), ds.repeat(repeat, 1, 1), pairs * repeat
    sync = torch.cuda.synchronize
    steps = 20

    def timed(depth):
        pp = Pipeline(eng, lines, nhwc, ds, hw, T, 1, pairs, pipelined=depth)      # sub-workloads are fed the producer's channel-last map (stated in "workload")

        def bar():
            pp.drain()
            sync()
        settle(pp.step, bar, min_s=settle_s, max_s=max(settle_s * 3, 1.0))
        elapsed, per_step, host_ms, last = timed_steps(pp.step, bar, steps, device)
        if depth:
            last = pp.last_joined
            per_step = per_step[depth:] if len(per_step) > depth + 1 else per_step
        return pp, elapsed, per_step, host_ms, last

    main_depth = 0 if (pairs == 1 or not pipelined) else pipelined
    pipe, elapsed, per_step, host_ms, (tb, ld, _g) = timed(main_depth)
    other = None
    if pipelined:
        _pp, el2, ps2, _h2, _l2 = timed(pipelined if main_depth == 0 else 0)
        other = {"value": round(tb.N * steps / el2, 1), "ms_per_step": round(el2 / steps * 1e3, 4), "ms_per_step_median": round(float(np.median(ps2)), 4)}
        del _pp, _l2
    prof, tot = profile_steps(eng, pipe.describe)
    out = {"workload": f"{name}: {pairs} pair(s) of {W}x{H}, {n_lines} lines/image -> {int(tb.N / (2 * pairs))} sub-lines x {T} tokens, fed the channel-last map",
           "descriptors_per_step": int(tb.N), "value": round(tb.N * steps / elapsed, 1), "unit": "line-descriptors/s",
           "schedule": f"pipelined, {main_depth} batches in flight" if main_depth else "one stream, step after step",
           "ms_per_step": round(elapsed / steps * 1e3, 4), "ms_per_step_median": round(float(np.median(per_step)), 4),
           "host_ms_per_step": round(float(np.median(host_ms)), 4), "gpu_ms_per_step_profiled": round(tot / 3, 4),
           "launches_per_step": int(sum(e["calls"] for e in prof) // 3),
           "roofline": roofline_of(prof[0], 3, tot, name if repeat == 1 else f"{name}x{repeat}"),
           "whole_step": whole_step_executed(prof, 3, elapsed / steps * 1e3), "kernels": breakdown_of(prof, 3)}
    if other is not None:
        out["serial" if main_depth else "pipelined"] = other
    if brief:
        for k in ("kernels", "host_ms_per_step", "gpu_ms_per_step_profiled"):
            out.pop(k, None)
        out["inputs"] = f"the {pairs // repeat}-pair set repeated {repeat}x"
    if not brief:        # argmin margins of this workload's own matches
        margs = pipe.match_args(tb, ld)
        dk, off_dk, m01, off_k0 = eng.match_offsets(*margs, LINE_CFG["nn_threshold"], True)
        out["argmin"] = argmin_margins(dk, off_dk, margs[2])
        out["matches_per_step"] = int((m01 >= 0).sum().item())
        if repeat == 1:     # the workload's own descriptors and matches against the CPU oracle (cfg2: 1 pair, cfg5: 8 pairs)
            out["oracle_check"] = oracle_check(lines, lambda i: nhwc[i].permute(2, 0, 1)[None].contiguous().cpu(), ds, hw,
                                               tuple(eng.cfg["image_shape"][-2:]), T, ld, tb,
                                               m01, dk, off_dk, off_k0, pairs)
            out["argmin"] = argmin_margins(dk, off_dk, margs[2], out["oracle_check"]["max_abs_dk_err_vs_oracle"])
    if pairs == 1:      # single pair: strict latency (submit, wait, repeat) of describe and of describe + match
        margs = pipe.match_args(tb, ld)
        lat, lat_m = [], []
        for _ in range(5):
            pipe.describe(); eng.match_offsets(*margs, LINE_CFG["nn_threshold"], True)
        sync()
        for _ in range(20):
            t1 = time.perf_counter()
            tb1, ld1 = pipe.describe()
            sync()
            t2 = time.perf_counter()
            eng.match_offsets(*pipe.match_args(tb1, ld1), LINE_CFG["nn_threshold"], True)
            sync()
            t3 = time.perf_counter()
            lat.append(t2 - t1)
            lat_m.append(t3 - t2)
        out["pair_latency_sync_ms"] = round(float(np.median(lat)) * 1e3, 4)
        out["pair_match_latency_ms"] = round(float(np.median(lat_m)) * 1e3, 4)
        # the same two waits as a caller that cares about latency would do them: an event after the call, polled (hipEventQuery)
        # instead of a blocking synchronize (whose interrupt-driven wake-up costs tens of microseconds on top of the GPU time)
        ev = torch.cuda.Event()
        lat, lat_m = [], []
        for _ in range(20):
            t1 = time.perf_counter()
            tb1, ld1 = pipe.describe()
            ev.record()
            while not ev.query():
                pass
            t2 = time.perf_counter()
            eng.match_offsets(*pipe.match_args(tb1, ld1), LINE_CFG["nn_threshold"], True)
            ev.record()
            while not ev.query():
                pass
            t3 = time.perf_counter()
            lat.append(t2 - t1)
            lat_m.append(t3 - t2)
        out["pair_latency_polled_ms"] = round(float(np.median(lat)) * 1e3, 4)
        out["pair_match_latency_polled_ms"] = round(float(np.median(lat_m)) * 1e3, 4)
        out["dropin"] = dropin_pair_section(device, lines, dd_nchw, nhwc, ds, hw, T)
    del pipe, lines, nhwc, ds
    return out


ARGMIN_CONTRACT = ("an argmin whose best-vs-second-best margin exceeds 4 x max|Dk - oracle| is identical to the oracle's by arithmetic; "
                   "below that, the reference's own fp32 result depends on summation order (two BLAS builds can flip it) and EITHER index "
                   "is a correct answer -- margins_at_risk counts those rows and columns (tests/test_gpu_properties.py::"
                   "test_near_tie_contract builds such a tie on purpose)")


def argmin_margins(dk, off_dk, dims, dk_err=None):
    """Best-vs-second-best gap of every row and every column of each pair's key-line distance matrix Dk (SURVEY.md section 7,
    "hard parts": an argmin is only as stable as its margin).  Returns the smallest gap over the batch, how many gaps are
    below 1e-5 (the descriptors agree with the reference to ~4e-7, i.e. distances to ~1e-6) and -- given the measured
    max|Dk - oracle| of the same run -- how many are below that error and below four times it (ARGMIN_CONTRACT)."""
    gaps = []
    for p in range(len(dims)):
        k0, k1 = int(dims[p][1]), int(dims[p][3])
        if k0 < 1 or k1 < 1:
            continue
        d = dk[int(off_dk[p]):int(off_dk[p + 1])].view(k0, k1).clamp(min=0)
        for m in (d, d.t()):
            if m.shape[1] < 2:
                continue
            two = torch.topk(m, 2, dim=1, largest=False).values
            gaps.append(two[:, 1] - two[:, 0])
    if not gaps:
        return {"min_argmin_margin": None, "margins_below_1e-5": 0, "argmins_checked": 0}
    g = torch.cat(gaps)
    out = {"min_argmin_margin": float(g.min().item()), "margins_below_1e-5": int((g < 1e-5).sum().item()), "argmins_checked": int(g.numel())}
    if dk_err is not None:
        out["margins_below_dk_err"] = int((g < dk_err).sum().item())
        out["margins_at_risk"] = int((g < 4 * dk_err).sum().item())
        out["dk_err_used"] = dk_err
        out["contract"] = ARGMIN_CONTRACT
    return out


def oracle_check(lines, dense_nchw_of, ds, hw, norm_hw, T, ld, tb, m01, dk, off_dk, off_k0, pairs):
    """Outside every timed region (cfg2: the pair; cfg5: its 8 pairs, where the argmin margins are smallest): descriptors and
    line matches of the workload's own inputs against the CPU oracle (the oracle is the checker here, never the thing
    measured).  dense_nchw_of(i) -> image i's [1,256,H/8,W/8] map on the CPU.  norm_hw: the shape the ENGINE normalises key-line
    coordinates with -- the constructor-time config['image_shape'], not the image's (models/line_transformer.py:206, :238)."""
    from oracle import linetr_oracle as O
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    cfg = dict(LINE_CFG, max_tokens=T)
    ld_c, cu, m_all, dk_all = ld.cpu().numpy(), tb.cu_n, m01.cpu().numpy(), dk.cpu().numpy()
    err = err_dk = 0.0
    differing = n_match = 0
    nt = torch.get_num_threads()
    torch.set_num_threads(min(8, nt))          # the oracle's small GEMMs run slower on 128 threads than on 8
    try:
        with torch.no_grad():
            for p in range(pairs):
                outs = []
                for i in (2 * p, 2 * p + 1):
                    o = O.preprocess(synth.array_to_keylines(lines[i]), (1, 1, *hw), dense_nchw_of(i), ds[i:i + 1].cpu(), cfg)
                    outs.append(O.forward(sd, o, norm_hw))
                    err = max(err, float(np.abs(ld_c[cu[i]:cu[i + 1]].T - outs[-1]["line_desc"][0].numpy()).max()))
                M, Dk = O.match_lines(outs[0]["line_desc"], outs[1]["line_desc"], outs[0]["mat_klines2sublines"][0],
                                      outs[1]["mat_klines2sublines"][0], LINE_CFG["nn_threshold"])
                got = np.zeros_like(M[0])
                m = m_all[int(off_k0[p]):int(off_k0[p + 1])]
                got[np.nonzero(m >= 0)[0], m[m >= 0]] = 1
                differing += int(not np.array_equal(got, M[0]))
                n_match += int(got.sum())
                err_dk = max(err_dk, float(np.abs(dk_all[int(off_dk[p]):int(off_dk[p + 1])].reshape(Dk[0].shape) - Dk[0]).max()))
    finally:
        torch.set_num_threads(nt)
    return {"pairs_checked": pairs, "matches_identical_to_oracle": differing == 0, "pairs_with_a_differing_match_matrix": differing,
            "matches": n_match, "max_abs_desc_err_vs_oracle": err, "max_abs_dk_err_vs_oracle": err_dk}


class _StubSuperPoint(torch.nn.Module):
    """Stands where SuperPoint stands in Matching (models/matching.py:20-22, out of the metric): hands back dense maps and
    key points that already sit in HBM, so that what is timed is the line branch of the reference call surface."""

    def __init__(self, outs):
        super().__init__()
        self.outs, self.i, self.config = outs, 0, {"nn_threshold": 0.7}

    def forward(self, data):
        o = self.outs[self.i % len(self.outs)]
        self.i += 1
        return dict(o)


class _StubLSD:
    def __init__(self, klines):
        self.klines, self.i = klines, 0

    def detect_torch(self, image):
        k = self.klines[self.i % len(self.klines)]
        self.i += 1
        return k


def dropin_pair_section(device, lines, dd_nchw, dd_nhwc, ds, hw, T, n_kp=512):
    """What the reference's own scripts pay per pair (match_line_pairs.py:90, demo_LineTR.py:207): the calls of the
    `models.*` import path -- LineTransformer.preprocess + forward for both images and Matching.match_lines -- with the
    detector's KeyLines and SuperPoint's maps injected (both out of the metric).  Wall time, one synchronisation at the end
    (match_lines returns NumPy arrays, like the reference)."""
    from models.matching import Matching         # the shim package: the import path of the reference's scripts
    H, W = hw
    g = torch.Generator(device=device).manual_seed(5)
    klines = [synth.array_to_keylines(l) for l in lines[:2]]

    def sp_out(i, nhwc):
        kp = torch.rand(n_kp, 2, device=device, generator=g) * torch.tensor([W - 1.0, H - 1.0], device=device)
        de = torch.nn.functional.normalize(torch.randn(256, n_kp, device=device, generator=g), dim=0)
        o = {"keypoints": [kp], "scores": (torch.rand(n_kp, device=device, generator=g),), "descriptors": [de],
             "dense_descriptor": dd_nchw[i:i + 1], "dense_score": ds[i:i + 1]}
        if nhwc:
            o["dense_descriptor_nhwc"] = dd_nhwc[i:i + 1]
        return o

    res = {}
    img = torch.zeros(1, 1, H, W, device=device)
    for tag, nhwc in (("", False), ("_fed_nhwc", True)):
        m = Matching({"auto_min_length": False, "linetransformer": {"mode": "train", "max_tokens": T, "image_shape": [H, W],
                                                                    **{k: LINE_CFG[k] for k in ("min_length", "token_distance", "remove_borders", "max_keylines", "nn_threshold")}}},
                     superpoint=_StubSuperPoint([sp_out(0, nhwc), sp_out(1, nhwc)]), lsd=_StubLSD(klines))
        m.linetransformer.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict()), strict=True)
        m = m.to(device).eval()
        lt = m.linetransformer
        sp = [sp_out(0, nhwc), sp_out(1, nhwc)]
        shape = (1, 1, H, W)

        def line_branch():
            o = [lt(lt.preprocess(klines[i], shape, sp[i], img)) for i in range(2)]
            return m.match_lines(o[0]["line_desc"], o[0]["mat_klines2sublines"], o[1]["line_desc"], o[1]["mat_klines2sublines"],
                                 lt.config["nn_threshold"])

        def full_forward():
            return m({"image0": img, "image1": img})

        with torch.no_grad():
            for fn, key in ((line_branch, "dropin_pair" + tag + "_ms"), (full_forward, "dropin_matching_forward" + tag + "_ms")):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(20):
                    t0 = time.perf_counter()
                    r = fn()
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                res[key] = round(float(np.median(ts)) * 1e3, 4)
            if not nhwc:
                ml = line_branch()[0]
                res["dropin_matches"] = int(ml.sum())
                # where the wall time goes: host-only part of preprocess (NumPy glue + record packing), and the GPU's share
                t0 = time.perf_counter()
                for _ in range(10):
                    from models.line_process import change_cv2_T_np, filter_by_length, remove_borders
                    kl = filter_by_length(remove_borders(change_cv2_T_np(klines[0]), LINE_CFG["remove_borders"], H, W, None),
                                          LINE_CFG["min_length"], LINE_CFG["max_keylines"])
                res["dropin_host_numpy_glue_ms_per_image"] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    line_branch()
                e1.record()
                torch.cuda.synchronize()
                res["dropin_pair_hip_event_ms"] = round(e0.elapsed_time(e1) / 10, 4)
    res["dropin_note"] = ("LineTransformer.preprocess + forward x2 + Matching.match_lines through `from models.matching import Matching` "
                          "(KeyLines and SuperPoint maps injected, both outside the metric); *_fed_nhwc: the producer also hands out the "
                          f"channel-last map (FusedHeadSuperPoint); dropin_matching_forward adds the point matcher on {n_kp} key points per image")
    return res


# =====================================================================================================================
# cfg4: 1024 homography-augmented pairs, strong-sharded, one all-gather, global matching
# =====================================================================================================================

class Cfg4Job:
    """BASELINE.json cfg4 on this rank: the pairs p = rank (mod world) of a P-pair job.

    step() = describe all local pairs in batches -> one slab (descriptors + counts + key-line maps) -> ONE all-gather
    -> every local query image (side 0 of a local pair p) is matched against the side-1 images of pairs
    p, p+1, .. p+S-1 (mod P) taken from the GATHERED set (for world > 1 all but one in world of them live on other
    ranks; s = 0 is the query's own partner, whose matches can be checked against the known homography)."""

    def __init__(self, eng, device, rank, world, pairs_total, batch_pairs, candidates, strength, dist=None, pipelined=3):
        self.eng, self.device, self.rank, self.world, self.dist = eng, device, rank, world, dist
        # the job's describe calls as ONE software pipeline over its batches (drained before the slab is packed)
        self.dpipe = DescribePipeline(eng, int(pipelined)) if pipelined else None
        self.P, self.S = pairs_total, candidates
        H, W, n_lines, lo, hi, T, _ = WORKLOADS["cfg4"]
        self.hw, self.T = (H, W), T
        self.mine = parallel.shard_pairs(pairs_total, rank, world)
        self.batch_pairs = batch_pairs
        lines, nhwc, ds, self.gt = [], [], [], {}
        g = torch.Generator(device=device)
        for p in self.mine:
            l0, l1, m, gt = synth.homography_pair(40000 + p, n_lines, H, W, lo, hi, strength=strength)
            g.manual_seed(40000 + p)
            dd0 = torch.nn.functional.normalize(torch.randn(1, 256, H // 8, W // 8, device=device, generator=g), p=2, dim=1)
            ds0 = torch.rand(1, H, W, device=device, generator=g)
            dd1, ds1 = synth.warp_dense_maps(dd0, ds0, m, seed=p)
            lines += [l0, l1]
            nhwc += [dd0.permute(0, 2, 3, 1).contiguous(), dd1.permute(0, 2, 3, 1).contiguous()]
            ds += [ds0, ds1]
            self.gt[p] = (l0, l1, m, gt)
        self.batches = []
        for b0 in range(0, len(self.mine), batch_pairs):
            i0, i1 = 2 * b0, 2 * min(b0 + batch_pairs, len(self.mine))
            pipe = Pipeline(eng, lines[i0:i1], torch.cat(nhwc[i0:i1]), torch.cat(ds[i0:i1]), self.hw, T, 1, (i1 - i0) // 2, dpipe=self.dpipe)
            self.batches.append(pipe)
        self.lines = lines
        per_rank = (pairs_total + world - 1) // world
        self.n_img_cap = 2 * per_rank
        self.rows_cap = max(sum(p.rows_cap for p in self.batches), 1)
        if world > 1:     # every rank's slab must have the same height
            t = torch.tensor([self.rows_cap], dtype=torch.int64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            self.rows_cap = int(t.item())
        self.slab = torch.zeros((parallel.slab_rows(self.n_img_cap, self.rows_cap), 256), dtype=torch.float32, device=device)
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        self.last = None

    def step(self):
        e0, e1, e2, e3 = self.ev
        e0.record()
        # ---- compute: describe every local batch, results appended into the slab's regions -----------------------------------
        lds, cu_n, cu_k, s2l = [], [0], [0], []
        if self.dpipe is not None:
            outs = [done for done in (pipe.submit() for pipe in self.batches) if done is not None] + self.dpipe.drain()
        else:
            outs = [pipe.describe() for pipe in self.batches]
        for tb, ld in outs:
            lds.append(ld); s2l.append(tb.sub2line)
            cu_n += list(np.asarray(tb.cu_n[1:], dtype=np.int64) + cu_n[-1])
            cu_k += list(np.asarray(tb.cu_k[1:], dtype=np.int64) + cu_k[-1])
        cu_n, cu_k = np.asarray(cu_n, dtype=np.int32), np.asarray(cu_k, dtype=np.int32)
        ld = torch.cat(lds) if len(lds) > 1 else lds[0]
        s2l = torch.cat(s2l) if len(s2l) > 1 else s2l[0]
        parallel.pack_descriptors(ld, cu_n, self.n_img_cap, self.rows_cap, self.slab, cu_k=cu_k, sub2line=s2l)
        e1.record()
        # ---- the single collective ---------------------------------------------------------------------------------------------
        if self.world > 1:
            gathered = parallel.allgather_descriptors(self.slab)
        else:
            gathered = self.slab[None]
        e2.record()
        # ---- global matching on the gathered set --------------------------------------------------------------------------------
        gs = parallel.GatheredSet(gathered, self.n_img_cap, self.rows_cap)
        queries, cands = [], []
        for p in self.mine:
            for s in range(self.S):
                c = (p + s) % self.P
                queries.append((self.rank, 2 * (p // self.world)))
                rc, lc = parallel.owner_of(c, self.world)
                cands.append((rc, 2 * lc + 1))
        res = []
        CH = 2048          # pair-matches per linetr_match call (bounds the distance-matrix workspace)
        for i in range(0, len(queries), CH):
            res.append(parallel.global_match(self.eng, gs, queries[i:i + CH], cands[i:i + CH], LINE_CFG["nn_threshold"], True))
        e3.record()
        self.last = (gs, ld, cu_n, cu_k, res, queries, cands)
        return int(cu_n[-1])

    def phase_ms(self):
        torch.cuda.synchronize()
        e0, e1, e2, e3 = self.ev
        return e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)

    def recall(self, max_pairs=16):
        """matches of (query p, candidate p) -- a pair's own two views -- against the known homography: a key-line of view
        0 whose warped end points coincide with a key-line of view 1 (0.5 px) is a ground-truth match."""
        gs, ld, cu_n, cu_k, res, queries, cands = self.last
        dk, off_dk, m01, off_k0 = res[0]
        m01 = m01.cpu().numpy()
        hit = tot = matched = 0
        for qi in range(0, min(len(queries), 2048), self.S):
            p = self.mine[qi // self.S]
            if qi // self.S >= max_pairs:
                break
            loc = p // self.world
            pipe = self.batches[loc // self.batch_pairs]
            j = 2 * (loc % self.batch_pairs)
            tb, _ = pipe.describe()
            k0 = tb.klines[tb.cu_k[j]:tb.cu_k[j + 1]].cpu().numpy().astype(np.float64)
            k1 = tb.klines[tb.cu_k[j + 1]:tb.cu_k[j + 2]].cpu().numpy().astype(np.float64)
            m = self.gt[p][2]
            w = synth.warp_points(m, k0.reshape(-1, 2)).reshape(-1, 2, 2)
            mm = m01[off_k0[qi]:off_k0[qi + 1]]
            for i in range(len(k0)):
                d = np.minimum(np.abs(k1 - w[i][None]).reshape(len(k1), -1).max(1),
                               np.abs(k1 - w[i][::-1][None]).reshape(len(k1), -1).max(1))
                jj = int(d.argmin()) if len(d) else -1
                if jj >= 0 and d[jj] < 0.5:
                    tot += 1
                    hit += int(mm[i] == jj)
            matched += int((mm >= 0).sum())
        return {"gt_pairs": tot, "recovered": hit, "recall": round(hit / max(tot, 1), 4), "matches": matched}


def run_cfg4(args, eng, device, rank, world, dist):
    job = Cfg4Job(eng, device, rank, world, args.pairs_total, args.pairs or WORKLOADS["cfg4"][6], args.candidates,
                  args.homography_strength, dist, pipelined=args.pipeline)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    settle(job.step, torch.cuda.synchronize, min_s=args.settle_s, max_s=max(3 * args.settle_s, 1.0), window=2,
           agree=make_agree(dist, world, device))
    for _ in range(args.warmup):
        job.step()
    elapsed, per_step, host_ms, n_local = timed_steps(job.step, barrier, args.steps, device)
    phases = np.array(job.phase_ms())
    if world > 1:
        t = torch.tensor([elapsed, float(n_local), *phases], dtype=torch.float64, device=device)
        mx = t.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = t.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, n_total, phases = float(mx[0]), float(sm[1]), mx[2:].cpu().numpy()
    else:
        n_total = float(n_local)
    rec = job.recall()
    H, W, n_lines, _, _, T, _ = WORKLOADS["cfg4"]
    out = {
        "metric": "line_descriptors_per_sec", "value": round(n_total * args.steps / elapsed, 1), "unit": "line-descriptors/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32 in/out; GEMMs as 6 bf16-split MFMA products, fp32 accumulate (fp32-faithful)", "data": "synthetic",
        "config": {"workload": f"cfg4: {args.pairs_total} homography-augmented pairs of {W}x{H} ({n_lines} lines/image, recipe "
                               f"dataloaders/confs/homography.yaml, strength {args.homography_strength}), pair p -> rank p mod {world}, "
                               f"one all-gather, {args.candidates} gathered candidates per query",
                   "pairs_total": args.pairs_total, "pairs_per_describe_call": job.batch_pairs,
                   "descriptors_per_step": int(n_total), "collective": "all_gather(slab)" if world > 1 else "none (one rank)",
                   "pair_matches_per_step": args.pairs_total * args.candidates},
        "compute_ms": round(float(phases[0]), 3), "gather_ms": round(float(phases[1]), 3),
        "global_match_ms": round(float(phases[2]), 3),
        "ms_per_step_median": round(float(np.median(per_step)), 3), "host_ms_per_step": round(float(np.median(host_ms)), 3),
        "recall_vs_homography_rank0": rec,
        "recall_note": "seeded, untrained weights are not viewpoint-invariant: recall is only meaningful for mild views "
                       "(--homography-strength 0.05 gives > 0.8; tests/test_gpu_cfg4.py)",
    }
    return out


def cfg4_section(eng, device, args, steps=3, n_check=8):
    """cfg4 sub-object of the N = 1 line (BASELINE.json's fourth configuration: the whole 1024-pair job on this one GPU -- every pair on
    rank 0, no collective): job time with its compute / pack / match split, descriptors/s, the first pairs' descriptors and line
    matches against the CPU oracle, the job's own global matches against the pairwise call, and recall against the known
    homographies (on a mild-view job too: seeded weights are not viewpoint-invariant).  Recipe: dataloaders/utils/homographies.py:12-141,
    dataloaders/confs/homography.yaml:31-46."""
    H, W, n_lines, _lo, _hi, T, batch_pairs = WORKLOADS["cfg4"]
    thr = LINE_CFG["nn_threshold"]
    sync = torch.cuda.synchronize
    t_setup = time.perf_counter()
    job = Cfg4Job(eng, device, 0, 1, args.pairs_total, batch_pairs, args.candidates, 1.0, None, pipelined=args.pipeline)
    t_setup = time.perf_counter() - t_setup
    for _ in range(2):
        job.step()
    sync()
    elapsed, per_step, host_ms, n_desc = timed_steps(job.step, sync, steps, device)
    phases = job.phase_ms()
    out = {"workload": f"cfg4: {args.pairs_total} homography-augmented pairs of {W}x{H} ({n_lines} lines/image, recipe dataloaders/confs/"
                       f"homography.yaml at full strength) on ONE GPU: {len(job.batches)} describe calls of <= {batch_pairs} pairs "
                       f"({'pipelined, ' + str(args.pipeline) + ' batches in flight' if args.pipeline else 'one stream'}), one slab, "
                       f"{args.candidates} candidates per query out of the packed set; the collective only exists for N > 1",
           "pairs_total": args.pairs_total, "descriptors_per_job": int(n_desc), "pair_matches_per_job": args.pairs_total * args.candidates,
           "value": round(n_desc * steps / elapsed, 1), "unit": "line-descriptors/s", "ms_per_job": round(elapsed / steps * 1e3, 3),
           "compute_ms": round(float(phases[0]), 3), "gather_ms": round(float(phases[1]), 3), "global_match_ms": round(float(phases[2]), 3),
           "host_ms_per_job": round(float(np.median(host_ms)), 3), "setup_s": round(t_setup, 1),
           "recall_vs_homography": job.recall(),
           "recall_note": "full-strength views: seeded, untrained weights are not viewpoint-invariant, so this recall is a property of the "
                          "weights, not of the kernels (parity is what oracle_check asserts); recall_mild_views is the meaningful figure"}
    # the job's own inputs against the CPU oracle: the first n_check pairs of the first describe call, matched pair by pair
    pipe0 = job.batches[0]
    n_check = min(n_check, pipe0.pairs)
    tb, ld = pipe0.describe()
    margs = pipe0.match_args(tb, ld)
    m8 = (margs[0], margs[1], margs[2][:n_check], *(m[:n_check] for m in margs[3:]))
    dk, off_dk, m01, off_k0 = eng.match_offsets(*m8, thr, True)
    out["oracle_check"] = oracle_check(job.lines, lambda i: pipe0.dd[i].permute(2, 0, 1)[None].contiguous().cpu(), pipe0.ds, (H, W),
                                       tuple(eng.cfg["image_shape"][-2:]), T, ld, tb, m01, dk, off_dk, off_k0, n_check)
    out["argmin"] = argmin_margins(dk, off_dk, m8[2], out["oracle_check"]["max_abs_dk_err_vs_oracle"])
    # ... and the job's global matcher (query p against candidate p, read out of the packed slab) against that pairwise call
    _gs, _ld, _cu_n, _cu_k, res, queries, _cands = job.last
    gm01, goff = res[0][2].cpu().numpy(), res[0][3]
    pm01 = m01.cpu().numpy()
    same = all(np.array_equal(gm01[int(goff[p * job.S]):int(goff[p * job.S + 1])], pm01[int(off_k0[p]):int(off_k0[p + 1])]) for p in range(n_check))
    out["global_match_equals_pairwise_call"] = bool(same)
    del job, pipe0, tb, ld
    torch.cuda.empty_cache()
    mild = Cfg4Job(eng, device, 0, 1, min(128, args.pairs_total), batch_pairs, args.candidates, 0.05, None, pipelined=args.pipeline)
    mild.step()
    sync()
    out["recall_mild_views"] = {"homography_strength": 0.05, "pairs": min(128, args.pairs_total), **mild.recall()}
    del mild
    torch.cuda.empty_cache()
    return out


# =====================================================================================================================

def fail_line(msg, world, rank, code=3):
    """A multi-GPU run that cannot be trusted must not look like a slow one: ONE JSON line with "error" and a null value from rank 0,
    then a non-zero exit code on every rank."""
    if rank == 0:
        print(json.dumps({"metric": "line_descriptors_per_sec", "value": None, "unit": "line-descriptors/s", "n_gpus": world,
                          "error": msg}), flush=True)
    sys.stdout.flush()
    os._exit(code)


def preflight(dist, world, rank, device, backend):
    """Before anything is timed on N > 1 ranks: the process group really has N ranks that see each other (an all-reduce of ones and an
    all-gather of the rank ids through the SAME backend the benchmark uses), and -- unless the one-device test hook is on -- every rank
    sits on its own GPU.  Returns what was seen; raises SystemExit through fail_line otherwise."""
    ones = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    ids = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int64, device=device))
    torch.cuda.synchronize()
    seen = sorted(int(v) for v in ids.cpu())
    if int(ones.item()) != world or seen != list(range(world)):
        fail_line(f"pre-flight: the collective saw {int(ones.item())} ranks {seen}, expected {world}", world, rank)
    # one node (the contract: --nnodes=1): a GPU is identified by the index this rank opened within the set of devices visible to it
    # (uuid / PCI ids are not relied on: they are not populated by every ROCm / torch build, and a false alarm here would fail a good run)
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES") or ""
    tag = f"{vis}#{device.index}"
    box = [None] * world
    dist.all_gather_object(box, (rank, device.index, tag))
    distinct = len({t for _r, _i, t in box})
    if distinct != world and not os.environ.get("LINETR_BENCH_ONE_DEVICE"):
        fail_line(f"pre-flight: {world} ranks on {distinct} distinct GPUs ({box})", world, rank)
    ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" and hasattr(torch.cuda, "nccl") else None
    return {"ranks_seen": world, "distinct_gpus": distinct, "backend": backend, "rccl_version": ver,
            "devices": [f"rank {r}: cuda:{i}" for r, i, _t in sorted(box)]}


def self_launch(n):
    """Re-run this command under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free port)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("LINETR_BENCH_ONE_DEVICE"):
        print(f"bench.py: --gpus {n} but only {have} HIP device(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print("# self-launch: " + " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="image pairs per GPU per step (default: workload's)")
    ap.add_argument("--precision", default="bf16x6", choices=["f32", "bf16x6", "bf16x3", "f16x3"],
                    help="MFMA path of the dense contractions (bf16x6 = fp32-faithful split, the default)")
    ap.add_argument("--dense-layout", default="nchw", choices=["nhwc", "nchw"],
                    help="layout of the resident dense descriptor map: nchw = the reference's 'dense_descriptor' (models/superpoint.py:193; "
                         "the metric's input, SURVEY 8d), nhwc = what the repo's own producer emits (reported beside it as value_fed_nhwc)")
    ap.add_argument("--pipeline", type=int, default=3, choices=[0, 2, 3, 4],
                    help="batches in flight: 2-4 = consecutive steps as the software pipeline (linetr_describe_submit / _join: the front "
                         "of step i + 1 under the signature network of step i, full batch in every GEMM); 0: one step after the other on one stream")
    ap.add_argument("--settle-s", type=float, default=2.0, help="minimum seconds of load before anything is timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--alt-precisions", action="store_true",
                    help="also time the step in the other MFMA modes (their fast kernels differ from bf16x6's: not a like-for-like "
                         "cost of the six products, so no longer part of the default line)")
    ap.add_argument("--no-sub-workloads", action="store_true", help="skip the cfg2 / cfg5 / cfg4 sub-objects")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the cfg4 sub-object (the 1024-pair job on this GPU) of the default line")
    ap.add_argument("--cpu-budget", type=float, default=14.0)
    ap.add_argument("--pairs-total", type=int, default=1024, help="cfg4: pairs of the whole job")
    ap.add_argument("--candidates", type=int, default=4, help="cfg4: gathered candidate images matched per query image")
    ap.add_argument("--homography-strength", type=float, default=1.0, help="cfg4: 1 = the yaml's recipe, <1 milder views")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)      # internal: one process of cpu_baseline_socket
    args = ap.parse_args()

    if args.cpu_worker:
        wl, th, sec = args.cpu_worker.split(",")
        cpu_worker(wl, int(th), float(sec))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same command
        # line the driver's torch.distributed.run invocation uses) and hand back rank 0's JSON line through stdout
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the LineTR hot path has no CPU fallback")
    # test hooks (single-GPU box): LINETR_BENCH_ONE_DEVICE=1 maps every rank to cuda:0, LINETR_BENCH_BACKEND=gloo
    # replaces RCCL, so the N>1 code path can be exercised without N GPUs.  Never set by the driver.
    if os.environ.get("LINETR_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LINETR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    pre = preflight(dist, world, rank, device, os.environ.get("LINETR_BENCH_BACKEND", "nccl")) if world > 1 else None

    H, W, n_lines, lo, hi, T, def_pairs = WORKLOADS[args.workload]
    pairs = args.pairs or def_pairs
    eng = Engine(synth.calibrated_state_dict(), device, image_shape=[H, W])
    eng.set_precision(args.precision)

    if args.workload == "cfg4":
        out = run_cfg4(args, eng, device, rank, world, dist)
        out["preflight"] = pre
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    lines, dd_nchw, dd_nhwc, ds, hw, T = make_inputs(args.workload, pairs, rank, device, eng)
    feed = dd_nhwc if args.dense_layout == "nhwc" else dd_nchw
    pipe = Pipeline(eng, lines, feed, ds, hw, T, world, pairs, args.dense_layout, pipelined=args.pipeline)
    if world > 1:     # every rank's slab must have the same height: the largest sub-line count of any rank
        t = torch.tensor([pipe.rows_cap], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pipe.rows_cap = int(t.item())

    def barrier():
        pipe.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Python's cyclic collector scans the whole start-up heap (torch, the engine, the inputs) on a full collection: an 8-10 ms pause of the
    # HOST every few hundred steps (tools/pipeline_soak.py).  A one-stream step has a deep launch queue to absorb it; the describe pipeline
    # keeps the host at most `depth` batches ahead and turns it into a bubble.  Everything alive now is set-up: park it in the permanent
    # generation (the collector keeps running on what the steps allocate).
    import gc
    gc.collect()
    gc.freeze()
    # ---- steady state, then the contract: W untimed warm-up steps, EXACTLY K timed steps between barriers ----------
    settle_hist = settle(pipe.step, barrier, min_s=args.settle_s, agree=make_agree(dist, world, device))
    for _ in range(args.warmup):
        pipe.step()
    elapsed, per_step, host_ms, (tb, ld, _g) = timed_steps(pipe.step, barrier, args.steps, device)
    if args.pipeline:
        # the closing barrier joined the last batch; the first step of the region only submits (nothing in flight behind the opening
        # barrier), so its event interval is empty, the first join's interval is the pipeline filling, and the last batches' completions
        # fall behind the last event: the per-step statistics are taken over the K - depth completion-to-completion intervals in between
        tb, ld, _g = pipe.last_joined
        per_step = per_step[args.pipeline:] if len(per_step) > args.pipeline + 1 else (per_step[per_step > 0.05] if (per_step > 0.05).any() else per_step)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([float(tb.N)], dtype=torch.float64, device=device)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        n_desc_step = float(cnt.item())
    else:
        n_desc_step = float(tb.N)
    ms_per_step = elapsed / args.steps * 1e3
    value = n_desc_step * args.steps / elapsed
    # host side of a step alone (C++ pre-filter on the worker pool; the rest of host_ms_per_step is Python + launches)
    t0 = time.perf_counter()
    for _ in range(10):
        pipe.prefilter_only()
    host_prefilter_ms = (time.perf_counter() - t0) / 10 * 1e3

    gathered_ok, global_match, gather_ms, gather_ms_per_rank, native_collective = None, None, None, None, None
    if world > 1 and _g is not None:
        # the collective alone (not part of `value`'s clock, which overlaps it with the next step's compute)
        slab = pipe.pack(tb, ld)
        for _ in range(2):
            parallel.allgather_descriptors(slab, async_op=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            parallel.allgather_descriptors(slab, async_op=False)
        torch.cuda.synchronize()
        gather_ms = round((time.perf_counter() - t0) / 5 * 1e3, 4)
        box = [None] * world
        dist.all_gather_object(box, gather_ms)
        gather_ms_per_rank = [float(v) for v in box]
        if os.environ.get("LINETR_BENCH_COLLECTIVE") == "native":
            # the C ABI's own collective (linetr_allgather_desc over an RCCL communicator of the same ranks) beside torch's: same
            # bytes out, timed the same way
            nag = parallel.NativeAllGather(device)
            ref = parallel.allgather_descriptors(slab, async_op=False)
            got = nag(slab)
            torch.cuda.synchronize()
            same = bool(torch.equal(got, ref))
            t0 = time.perf_counter()
            for _ in range(5):
                nag(slab)
            torch.cuda.synchronize()
            native_collective = {"entry_point": "linetr_allgather_desc", "equal_to_torch_all_gather": same,
                                 "gather_ms": round((time.perf_counter() - t0) / 5 * 1e3, 4)}
            nag.close()
            if not same:
                fail_line("linetr_allgather_desc returned other bytes than torch.distributed.all_gather_into_tensor", world, rank)
        # every rank's slab of the last all-gather carries that rank's descriptors; then GLOBAL matching on the gathered set:
        # this rank's side-0 images against the side-1 images of the NEXT rank's pairs (descriptors this rank never computed)
        gs = parallel.GatheredSet(_g, pipe.n_img_cap, pipe.rows_cap)
        gathered_ok = True
        for r in range(world):
            d_r, cu_r = parallel.unpack_descriptors(_g[r], pipe.n_img_cap, pipe.rows_cap)
            gathered_ok &= bool(len(cu_r) == 2 * pairs + 1 and d_r.shape[0] == cu_r[-1])
            if r == rank:
                gathered_ok &= bool(torch.equal(d_r, ld))
            gathered_ok &= bool(((d_r.norm(dim=1) - 1).abs() < 1e-4).all().item())
        nxt = (rank + 1) % world
        q = [(rank, 2 * p) for p in range(pairs)]
        c = [(nxt, 2 * p + 1) for p in range(pairs)]
        for _ in range(2):
            parallel.global_match(eng, gs, q, c, LINE_CFG["nn_threshold"], True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            _dk, _odk, m01g, _ok0 = parallel.global_match(eng, gs, q, c, LINE_CFG["nn_threshold"], True)
        torch.cuda.synchronize()
        global_match = {"ms_per_batch": round((time.perf_counter() - t0) / 5 * 1e3, 4), "pair_matches": pairs,
                        "against": f"side-1 images of rank {nxt} (gathered)", "matches": int((m01g >= 0).sum().item())}
        if os.environ.get("LINETR_BENCH_INJECT") == "bad_gather" and rank == world - 1:
            gathered_ok = False          # test hook (tests/test_gpu_bench_launch.py): the failure path below must be loud
        # every rank must have seen every rank's rows intact: one rank's failure fails the run
        ok_all = torch.tensor([int(gathered_ok)], dtype=torch.int64, device=device)
        dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
        gathered_ok = bool(ok_all.item())
        if not gathered_ok:
            fail_line("gathered_rows_checked is false: a rank received descriptor rows that differ from what their owner packed", world, rank)

    # ---- pair-match ms (a19-a21) on the descriptors just produced --------------------------------------------
    margs = pipe.match_args(tb, ld)
    for _ in range(2):
        eng.match_offsets(*margs, LINE_CFG["nn_threshold"], True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        dk, off_dk, m01, _ok0 = eng.match_offsets(*margs, LINE_CFG["nn_threshold"], True)
    torch.cuda.synchronize()
    pair_match_ms = (time.perf_counter() - t0) / reps / pairs * 1e3
    n_matches = int((m01 >= 0).sum().item())
    argmin = argmin_margins(dk, off_dk, margs[2])
    # the step's own descriptors and line matches against the CPU oracle on the same inputs (outside every timed region; N = 1 only,
    # skipped together with the CPU baseline): "line matches bit-exact by index" on the benchmark's input, not only on fixtures
    oracle_chk = None
    if world == 1 and not args.no_cpu_baseline:
        oracle_chk = oracle_check(lines, lambda i: dd_nchw[i:i + 1].cpu(), ds, hw, (H, W), T, ld, tb, m01, dk, off_dk, _ok0, pairs)
        argmin = argmin_margins(dk, off_dk, margs[2], oracle_chk["max_abs_dk_err_vs_oracle"])
    # ... and for ONE pair at a time (latency of get_dist_matrix + subline2keyline + nn_matcher_distmat)
    one_args = (margs[0], margs[1], margs[2][:1], margs[3][:1], margs[4][:1], margs[5][:1], margs[6][:1])
    for _ in range(3):
        eng.match_offsets(*one_args, LINE_CFG["nn_threshold"], True)
    torch.cuda.synchronize()
    lat = []
    for _ in range(20):
        t1 = time.perf_counter()
        eng.match_offsets(*one_args, LINE_CFG["nn_threshold"], True)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    pair_match_latency_ms = float(np.median(lat)) * 1e3

    # ---- per-kernel HIP-event profile of the same step (roofline of the dominant kernel) ----------------------
    pipe.drain()
    for _ in range(20):
        pipe.describe()                   # one-stream steps: the clocks settle where a one-stream step runs (after the light matcher section
    torch.cuda.synchronize()              # they are high, after pipelined steps ~8 % low: the package cap); nothing of the pipeline still runs
    prof_steps = 3
    prof, tot_ms = profile_steps(eng, pipe.describe, prof_steps)
    roofline = roofline_of(prof[0], prof_steps, tot_ms, args.workload)
    # the other kernels that matter (>= 3 % of the GPU time), same definitions, so that the line shows where each one stands
    roofline["other_kernels"] = [
        {k: v for k, v in roofline_of(e, prof_steps, tot_ms, args.workload).items()
         if k in ("kernel", "avg_launch_us", "launches_per_step", "share_of_gpu_time", "bound", "achieved", "peak", "unit", "frac",
                  "fp32_equivalent_tflops", "traffic", "mfma_busy")}
        for e in prof[1:] if e["ms"] / tot_ms >= 0.03]
    n_img = 2 * pairs
    alg_flops_step = sum(algorithmic_flops_per_image(int(n), T) for n in np.diff(tb.cu_n))

    out = {
        "metric": "line_descriptors_per_sec", "value": round(value, 1), "unit": "line-descriptors/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32 (v_mfma_f32_32x32x2_f32)", "bf16x6": "f32 in/out; GEMMs as 6 bf16-split MFMA products, fp32 accumulate (fp32-faithful)",
                  "bf16x3": "f32 in/out; GEMMs as 3 bf16-split MFMA products, fp32 accumulate (~1e-5)",
                  "f16x3": "f32 in/out; GEMMs as 3 fp16-split MFMA products, fp32 accumulate (~1e-6)"}[args.precision],
        "precision": args.precision, "data": "synthetic",
        "config": {"workload": f"{args.workload}: {pairs} pairs/GPU of {W}x{H}, {n_lines} lines/image -> "
                               f"{int(tb.N / n_img)} sub-lines x {T} tokens, d_model=256, seeded weights",
                   "pairs_per_gpu": pairs, "descriptors_per_step": int(n_desc_step),
                   "dense_layout": args.dense_layout + (" (the reference's 'dense_descriptor' [B,256,H/8,W/8]: the step includes the layout pass; "
                                                        "fed with the channel-last map the repo's producer linetr_superpoint_heads emits it is "
                                                        "ms_per_step_fed_nhwc / value_fed_nhwc)" if args.dense_layout == "nchw" else
                                                        " (channel-last map of the repo's own producer; the reference's layout: value_fed_nchw)"),
                   "collective": "all_gather(line_desc + counts + key-line maps)" if world > 1 else "none"},
        "ms_per_step_median": round(float(np.median(per_step)), 4), "ms_per_step_p10": round(float(np.percentile(per_step, 10)), 4),
        "ms_per_step_p90": round(float(np.percentile(per_step, 90)), 4),
        "ms_per_step_each": [round(float(v), 3) for v in per_step],
        "host_ms_each": [round(float(v), 3) for v in host_ms],
        # median: once the host is a launch queue ahead of the GPU the runtime blocks it for a whole GPU step (host_ms_each)
        "host_ms_per_step": round(float(np.median(host_ms)), 4), "host_prefilter_ms": round(host_prefilter_ms, 4),
        "settle": {"seconds_min": args.settle_s, "windows": len(settle_hist), "first_ms": round(settle_hist[0], 4),
                   "last3_ms": [round(v, 4) for v in settle_hist[-3:]]},
        "gathered_rows_checked": gathered_ok, "global_match": global_match, "gather_ms": gather_ms,
        "gather_ms_per_rank": gather_ms_per_rank, "preflight": pre, "native_collective": native_collective,
        "collective_backend": (None if world == 1 else os.environ.get("LINETR_BENCH_BACKEND", "nccl")),
        "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if world > 1 and hasattr(torch.cuda, "nccl") else None),
        "pair_match_ms": round(pair_match_ms, 4), "pair_match_latency_ms": round(pair_match_latency_ms, 4),
        "matches_per_step": n_matches, "argmin": argmin, "oracle_check": oracle_chk,
        "whole_step": {**whole_step_executed(prof, prof_steps, ms_per_step),
                       "survey_algorithmic_gflop_per_step": round(alg_flops_step / 1e9, 1),
                       "survey_note": "SURVEY 8(d)'s formula counts the reference's graph; exact algebra (K-projection fold, "
                                      "V / W5 after pooling, padding key with multiplicity) executes less -- not a roofline input"},
        "gpu_ms_per_step_profiled": round(tot_ms / prof_steps, 3),
        "roofline": roofline, "kernels": breakdown_of(prof, prof_steps),
    }
    if world == 1 and args.pipeline:
        # the same step on ONE stream, step after step (what r01-r05 measured), in the same process on the same inputs
        pipe.drain()
        for _ in range(8):
            pipe.describe()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pipe.describe()
        torch.cuda.synchronize()
        ser = (time.perf_counter() - t0) / 10 * 1e3
        out["pipeline"] = {"batches_in_flight": args.pipeline,
                           "stages": "linetr_describe_submit: stage k of a batch on stream k (cuts behind signature layer 0 and layer 3 for "
                                     "3-4 batches in flight, behind layer 2 for 2); every GEMM sees the full batch; results bit-identical "
                                     "to linetr_describe (tests/test_gpu_pipeline.py)",
                           "ms_per_step_one_stream": round(ser, 4), "value_one_stream": round(n_desc_step / ser * 1e3, 1),
                           "speedup_vs_one_stream": round(ser / ms_per_step, 4),
                           "p90_over_p10": round(float(np.percentile(per_step, 90) / np.percentile(per_step, 10)), 4),
                           "trace": "profiles/r06_cfg3_pipeline_overlap.txt (rocprofv3 --kernel-trace of the steady state: which kernels "
                                    "execute concurrently)"}
    if world == 1:
        # the same step fed with the reference's NCHW 'dense_descriptor' (one extra layout pass inside linetr_describe)
        other = "nchw" if args.dense_layout == "nhwc" else "nhwc"
        odd = dd_nchw if other == "nchw" else dd_nhwc
        opipe = Pipeline(eng, lines, odd, ds, hw, T, 1, pairs, other, pipelined=args.pipeline)   # same schedule as the headline step

        def obar():
            opipe.drain()
            torch.cuda.synchronize()
        for _ in range(10):
            opipe.step()
        obar()
        t0 = time.perf_counter()
        for _ in range(20):
            opipe.step()
        obar()
        out[f"ms_per_step_fed_{other}"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        del opipe
        out[f"value_fed_{other}"] = round(n_desc_step / out[f"ms_per_step_fed_{other}"] * 1e3, 1)
        out["producer"] = producer_section(eng, H, W, n_img)
    if world == 1 and args.alt_precisions:   # the same step in the other MFMA modes (few steps each); opt-in
        alt = {}
        for mode in ("bf16x3", "f16x3", "bf16x6", "f32"):
            if mode == args.precision:
                continue
            eng.set_precision(mode)
            for _ in range(10):
                pipe.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                tb_a, _ld, _gg = pipe.step()
            barrier()
            alt[mode] = round(tb_a.N * 10 / (time.perf_counter() - t0), 1)
        eng.set_precision(args.precision)
        out["alt_precisions_desc_per_s"] = alt
    cfg4_obj = None
    if world == 1 and not args.no_sub_workloads:
        del pipe, dd_nchw, dd_nhwc, ds, feed
        torch.cuda.empty_cache()
        for name in ("cfg2", "cfg5"):
            if name != args.workload:
                out[name] = sub_workload(eng, name, device, min(args.settle_s, 0.6), pipelined=args.pipeline)
        if "cfg5" in out:      # chip-filling batches of the long-line workload beside the 8-pair point
            out["cfg5"]["larger_batches"] = [sub_workload(eng, "cfg5", device, 0.4, repeat=r, brief=True, pipelined=args.pipeline) for r in (4, 8)]
        if args.workload != "cfg4" and not args.no_cfg4:
            cfg4_obj = cfg4_section(eng, device, args)      # (added to the line LAST, so that its oracle check ends the line)
        if "cfg2" in out:      # the single-pair figures of the metric, also at top level
            out["pair_latency_ms"] = out["cfg2"]["ms_per_step"]
            out["pair_latency_sync_ms"] = out["cfg2"]["pair_latency_sync_ms"]
            out["pair_latency_polled_ms"] = out["cfg2"].get("pair_latency_polled_ms")
            out["pair_match_latency_polled_ms"] = out["cfg2"].get("pair_match_latency_polled_ms")
            out["pair_match_latency_ms"] = out["cfg2"]["pair_match_latency_ms"]
    if world == 1 and not args.no_cpu_baseline:   # reported at N = 1 only (the contract), so scaling runs stay short
        cb = cpu_baseline(args.workload, args.cpu_budget)
        # ... and the same oracle filling the socket: N processes x the best thread count, all at once
        cb["socket"] = cpu_baseline_socket(args.workload, cb["cores"], cb["physical_cores"])
        out["cpu_baseline"] = cb
        best_cpu = max(cb["value"], cb["socket"]["value"])
        out["speedup_vs_cpu"] = round(value / best_cpu, 1)          # against the LARGER of the two CPU figures
        out["speedup_vs_cpu_single_process"] = round(value / cb["value"], 1)
    if cfg4_obj is not None:
        # oracle_check and the consistency verdict as the last keys of the last object of the line
        tail_keys = ("recall_mild_views", "global_match_equals_pairwise_call", "oracle_check")
        out["cfg4"] = {**{k: v for k, v in cfg4_obj.items() if k not in tail_keys}, **{k: cfg4_obj[k] for k in tail_keys if k in cfg4_obj}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
