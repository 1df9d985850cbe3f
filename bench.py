#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native LineTR hot path.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json / SURVEY.md section 8d): line-descriptors/sec = sub-line descriptors written to
`line_desc` / wall time of tokenise + forward (host pre-filter, H2D of the line records, tokeniser,
descriptor network; for N>1 also the single RCCL all-gather of the descriptors).  SuperPoint dense
maps are already resident in HBM, detected lines are resident on the host.  A "step" is one pass of
that path over one batch of P synthetic image pairs per GPU (weak scaling: every rank gets its own P
pairs).  Default workload = cfg3 of BASELINE.json (64 pairs of 640x480, 200 lines -> 199 sub-lines x
21 tokens per image), the configuration the roofline is defined on; --workload cfg2 / cfg5 select the
single-pair and the long-line configurations.  pair-match ms and the single-pair latency are reported
in the same JSON line.  One JSON line is printed by rank 0.
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

from linetr_amd import parallel, synth  # noqa: E402
from linetr_amd.engine import Engine  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
HBM_PEAK_GBS = 8000.0
SETTLE_STEPS = 25      # untimed steps run once before the W warm-up steps (see main)

WORKLOADS = {
    # name: (H, W, lines/image, len_lo, len_hi, max_tokens, default pairs per GPU)
    "cfg2": (480, 640, 200, 17.0, 167.0, 21, 1),
    "cfg3": (480, 640, 200, 17.0, 167.0, 21, 64),
    "cfg5": (960, 1280, 600, 40.0, 327.0, 41, 8),
}
LINE_CFG = dict(min_length=16, token_distance=8, remove_borders=8, max_keylines=-1, nn_threshold=0.8)


def algorithmic_flops_per_image(N, T, K=None):
    """F_img = 2*[N*T*108640 + N*108704 + N*(2*S*65536 + 2*65536 + 2*S*256 + 524288)
    + 7*(N*655360 + 512*N^2) + N*65536]."""
    S = T + 1
    return 2.0 * (N * T * 108640 + N * 108704 + N * (2 * S * 65536 + 2 * 65536 + 2 * S * 256 + 524288)
                  + 7 * (N * 655360 + 512 * N * N) + N * 65536)


def make_inputs(workload, pairs, rank, device):
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
    return lines, dd, ds, (H, W), T


class Pipeline:
    def __init__(self, eng, lines, dd, ds, hw, T, world, pairs, n_streams=1):
        self.eng, self.lines, self.dd, self.ds, self.hw, self.T = eng, lines, dd, ds, hw, T
        self.world, self.pairs = world, pairs
        self.n_streams = n_streams
        self.n_img_cap = 2 * pairs
        self.rows_cap = sum(len(l) for l in lines)
        self.packed = [None, None]      # double-buffered: the all-gather of step i overlaps the compute of step i+1
        self.pending = [None, None]
        self.slot = 0
        # the detector output of the batch as one host array + row offsets (input format of the batched API)
        self.offsets = np.zeros(len(lines) + 1, dtype=np.int32)
        np.cumsum([len(l) for l in lines], out=self.offsets[1:])
        self.cat = np.ascontiguousarray(np.concatenate(lines), dtype=np.float64)

    def describe(self):
        e, c = self.eng, LINE_CFG
        return e.describe_lines(self.cat, self.offsets, self.dd, self.ds, remove_borders=c["remove_borders"],
                                min_length=c["min_length"], max_keylines=c["max_keylines"],
                                token_distance=c["token_distance"], max_tokens=self.T, n_streams=self.n_streams)

    def step(self):
        tb, ld = self.describe()
        gathered = None
        if self.world > 1:
            s = self.slot
            if self.pending[s] is not None:            # the buffer we are about to overwrite: its gather must be done
                self.pending[s][0].wait()
            self.packed[s] = parallel.pack_descriptors(ld, tb.cu_n, self.n_img_cap, self.rows_cap, self.packed[s])
            work, gathered = parallel.allgather_descriptors(self.packed[s], async_op=True)
            self.pending[s] = (work, gathered)
            self.slot ^= 1
        return tb, ld, gathered

    def drain(self):
        """wait for every outstanding all-gather (called before the closing barrier of a timed region)."""
        for i, p in enumerate(self.pending):
            if p is not None:
                p[0].wait()
                self.pending[i] = None

    def match(self, tb, ld):
        """image 2p vs image 2p+1 for every local pair."""
        cu_n, cu_k = tb.cu_n, tb.cu_k
        # de-interleave the two sides: side 0 = even images, side 1 = odd images (row ranges are contiguous per image)
        ev, od = slice(0, None, 2), slice(1, None, 2)
        n = np.diff(cu_n); k = np.diff(cu_k)
        idx0 = torch.cat([torch.arange(cu_n[i], cu_n[i + 1]) for i in range(0, len(n), 2)]).to(ld.device)
        idx1 = torch.cat([torch.arange(cu_n[i], cu_n[i + 1]) for i in range(1, len(n), 2)]).to(ld.device)
        d0, d1 = ld[idx0], ld[idx1]
        s0, s1 = tb.sub2line[idx0], tb.sub2line[idx1]
        c = lambda v: np.concatenate([[0], np.cumsum(v)]).astype(np.int32)
        args = (d0, c(n[ev]), s0, c(k[ev]), d1, c(n[od]), s1, c(k[od]))
        return args


def cpu_baseline(workload, budget_s=12.0, max_pairs=16):
    """The CPU oracle (a faithful port of the reference's as-executed PyTorch-CPU/NumPy path, incl. the
    per-line Python tokeniser loop and the full 22-token descriptive layer) timed on this box."""
    from oracle import linetr_oracle as O
    H, W, n_lines, lo, hi, T, _ = WORKLOADS[workload]
    sd = synth.to_torch_state_dict(synth.calibrated_state_dict())
    cfg = dict(LINE_CFG, max_tokens=T)
    n_desc, t_tok, t_fwd, t_match, pairs = 0, 0.0, 0.0, 0.0, 0
    t_start = time.perf_counter()
    with torch.no_grad():
        while pairs < max_pairs and (pairs < 2 or time.perf_counter() - t_start < budget_s):
            outs = []
            for side in (0, 1):
                seed = 5000 + 2 * pairs + side
                rows = synth.synth_lines(seed, n_lines, H, W, lo, hi)
                dd, ds = synth.synth_dense_maps(seed, H, W)
                kl = synth.array_to_keylines(rows)
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
    return {
        "value": n_desc / (t_tok + t_fwd), "unit": "line-descriptors/s", "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{pairs} {workload}-shaped pairs ({n_desc} descriptors), oracle tokenise {t_tok / pairs * 1e3:.1f} ms + "
                  f"forward {t_fwd / pairs * 1e3:.1f} ms + match {t_match / pairs * 1e3:.2f} ms per pair, "
                  f"torch {torch.__version__} CPU, {os.cpu_count()} logical cpus",
        "pair_match_ms": t_match / pairs * 1e3,
    }


PMC_KERNEL_NAMES = {   # profile class -> kernel symbol in profiles/r01_pmc_traffic.json (rocprofv3 --pmc run)
    # matched as a prefix of the demangled name (trailing template arguments may grow)
    "gemm_bf16x6_128x256": "void lt::gemm_split_kernel<128, 256, 2, 4, 3, true, 0",
    "gemm_bf16x3_128x256": "void lt::gemm_split_kernel<128, 256, 2, 4, 2, true, 0",
    "gemm_f16x3_128x256": "void lt::gemm_split_kernel<128, 256, 2, 4, 2, true, 1",
    "gemm_bf16x6_256x128": "void lt::gemm_split_kernel<256, 128, 4, 2, 3, true, 0",
    "gemm_bf16x3_256x128": "void lt::gemm_split_kernel<256, 128, 4, 2, 2, true, 0",
    "gemm_f16x3_256x128": "void lt::gemm_split_kernel<256, 128, 4, 2, 2, true, 1",
    "gemm_f32_128x128": "void lt::gemm_kernel<128, 128, 2, 2>",
}


def producer_section(eng, pipe, H, W, n_img, ms_per_step):
    """linetr_superpoint_heads on raw head outputs of this batch's shape (synthetic logits / descriptors), HIP-event
    timed; the same maths in stock PyTorch on the GPU for scale; and the descriptor step fed with the producer's
    NHWC map (no transposition pass inside linetr_describe)."""
    Hc, Wc = H // 8, W // 8
    g = torch.Generator(device=eng.device).manual_seed(3)
    sl = torch.randn(n_img, 65, Hc, Wc, device=eng.device, generator=g) * 2
    dr = torch.randn(n_img, 256, Hc, Wc, device=eng.device, generator=g)

    def timed(fn, reps=10):
        for _ in range(3):
            r = fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, r

    ms, (score, nhwc, _) = timed(lambda: eng.superpoint_heads(sl, dr, nhwc=True, nchw=False))

    def torch_heads():
        p = torch.softmax(sl, 1)[:, :-1]
        p = p.permute(0, 2, 3, 1).reshape(n_img, Hc, Wc, 8, 8).permute(0, 1, 3, 2, 4).reshape(n_img, Hc * 8, Wc * 8)
        return p, torch.nn.functional.normalize(dr, p=2, dim=1)
    ms_torch, _ = timed(torch_heads)
    nbytes = n_img * Hc * Wc * (256 * 2 + 65 + 64) * 4
    c = LINE_CFG

    def step_nhwc():
        return eng.describe_lines(pipe.cat, pipe.offsets, nhwc, score, remove_borders=c["remove_borders"],
                                  min_length=c["min_length"], max_keylines=c["max_keylines"],
                                  token_distance=c["token_distance"], max_tokens=pipe.T, dense_layout="nhwc")
    for _ in range(5):
        step_nhwc()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step_nhwc()
    torch.cuda.synchronize()
    ms_nhwc = (time.perf_counter() - t0) / 10 * 1e3
    return {"kernel": "sp_desc_head + sp_score_head (linetr_superpoint_heads)", "images": n_img, "ms": round(ms, 4),
            "algorithmic_bytes": nbytes, "achieved_GBps": round(nbytes / ms / 1e6, 1), "hbm_peak_GBps": HBM_PEAK_GBS,
            "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4), "torch_same_ops_ms": round(ms_torch, 4),
            "ms_per_step_fed_nhwc": round(ms_nhwc, 4), "ms_per_step_fed_nchw": round(ms_per_step, 4)}


def pmc_traffic(kernel_class):
    """HBM-side bytes per launch of `kernel_class` from the committed rocprofv3 PMC pass (FETCH_SIZE/WRITE_SIZE in
    KiB, separate --pmc runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads
    on gfx950).  None if no PMC data is committed for this kernel."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    name = PMC_KERNEL_NAMES.get(kernel_class)
    if not name or not os.path.exists(path):
        return None
    rec = next((v for k, v in json.load(open(path)).items() if k.startswith(name)), None)
    if not rec or "FETCH_SIZE_KiB_avg" not in rec or "WRITE_SIZE_KiB_avg" not in rec:
        return None
    return (2.0 * rec["FETCH_SIZE_KiB_avg"] + rec["WRITE_SIZE_KiB_avg"]) * 1024.0


def roofline_of(dom, prof_steps, tot_ms, precision):
    """roofline object of the dominant kernel (largest summed HIP-event time over the profiled steps)."""
    common = {"kernel": dom["name"], "avg_launch_us": round(dom["ms"] / dom["calls"] * 1e3, 2),
              "launches_per_step": dom["calls"] // prof_steps, "share_of_gpu_time": round(dom["ms"] / tot_ms, 3)}
    traffic = pmc_traffic(dom["name"])
    if dom["flops"] > 0:
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12      # algorithmic fp32 flops (2*M*N*K of each launch) / time
        r = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
             "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["calls"]),
             "peak_note": "157.3 TF = dense fp32 matrix peak (the contract of the kernel is fp32 in / fp32 out)"}
        if "bf16x" in dom["name"] or "f16x3" in dom["name"]:
            terms = 6 if "bf16x6" in dom["name"] else 3
            r.update({"mfma_flops_per_algorithmic_flop": terms,
                      "bf16_pipe_frac": round(ach * terms / 2500.0, 4),
                      "pipe_note": f"each fp32 product = {terms} bf16 MFMA products; {terms}*achieved / 2.5 PF dense bf16 peak"})
    else:
        ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic}
    return {**common, **r}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="image pairs per GPU per step (default: workload's)")
    ap.add_argument("--precision", default="bf16x6", choices=["f32", "bf16x6", "bf16x3", "f16x3"],
                    help="MFMA path of the dense contractions (bf16x6 = fp32-faithful split, the default)")
    ap.add_argument("--streams", type=int, default=1, help="independent sub-batches run on this many HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-precisions", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

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

    H, W, n_lines, lo, hi, T, def_pairs = WORKLOADS[args.workload]
    pairs = args.pairs or def_pairs
    eng = Engine(synth.calibrated_state_dict(), device, image_shape=[H, W])
    eng.set_precision(args.precision)
    lines, dd, ds, hw, T = make_inputs(args.workload, pairs, rank, device)
    pipe = Pipeline(eng, lines, dd, ds, hw, T, world, pairs, args.streams)

    def barrier():
        pipe.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one-time settling before the contract's W warm-up steps: first-touch of workspaces / pinned staging, and ~0.1 s of
    # load so that the power manager has left its idle state (a cold start was measured up to 15 % slower per step)
    for _ in range(SETTLE_STEPS):
        pipe.step()
    barrier()
    for _ in range(args.warmup):
        tb, ld, _g = pipe.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tb, ld, _g = pipe.step()
    barrier()
    elapsed = time.perf_counter() - t0
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
    gathered_ok = None
    if world > 1 and _g is not None:     # every rank's slab of the last all-gather carries that rank's descriptors
        gathered_ok = True
        for r in range(world):
            d_r, cu_r = parallel.unpack_descriptors(_g[r], pipe.n_img_cap)
            gathered_ok &= bool(len(cu_r) == 2 * pairs + 1 and d_r.shape[0] == cu_r[-1])
            if r == rank:
                gathered_ok &= bool(torch.equal(d_r, ld))
            gathered_ok &= bool(((d_r.norm(dim=1) - 1).abs() < 1e-4).all().item())

    # ---- pair-match ms (a19-a21) on the descriptors just produced --------------------------------------------
    margs = pipe.match(tb, ld)
    for _ in range(2):
        eng.match(*margs, LINE_CFG["nn_threshold"], True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        dk, off_dk, m01 = eng.match(*margs, LINE_CFG["nn_threshold"], True)
    torch.cuda.synchronize()
    pair_match_ms = (time.perf_counter() - t0) / reps / pairs * 1e3
    n_matches = int((m01 >= 0).sum().item())
    # ... and for ONE pair at a time (latency of get_dist_matrix + subline2keyline + nn_matcher_distmat)
    n0, n1 = int(tb.cu_n[1]), int(tb.cu_n[2] - tb.cu_n[1])
    k0, k1 = int(tb.cu_k[1]), int(tb.cu_k[2] - tb.cu_k[1])
    one_args = (ld[:n0], np.array([0, n0]), tb.sub2line[:n0], np.array([0, k0]), ld[n0:n0 + n1], np.array([0, n1]),
                tb.sub2line[n0:n0 + n1], np.array([0, k1]))
    for _ in range(3):
        eng.match(*one_args, LINE_CFG["nn_threshold"], True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.match(*one_args, LINE_CFG["nn_threshold"], True)
    torch.cuda.synchronize()
    pair_match_latency_ms = (time.perf_counter() - t0) / 20 * 1e3

    # ---- single-pair latency (cfg2 shape, tokenise + forward + match) -----------------------------------------
    one = Pipeline(eng, lines[:2], dd[:2], ds[:2], hw, T, 1, 1)
    for _ in range(3):
        pipe.step()                       # bring the clocks back up after the light matcher section
    for _ in range(20):
        tb1, ld1 = one.describe()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        tb1, ld1 = one.describe()
    torch.cuda.synchronize()
    pair_latency_ms = (time.perf_counter() - t0) / 30 * 1e3   # back-to-back single pairs (throughput-latency)
    lat = []
    for _ in range(10):                   # and strictly one at a time: submit, wait, repeat
        t1 = time.perf_counter()
        one.describe()
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    pair_latency_sync_ms = float(np.median(lat)) * 1e3

    # ---- per-kernel HIP-event profile of the same step (roofline of the dominant kernel) ----------------------
    prof_steps = 3
    eng.set_profiling(True)
    for _ in range(prof_steps):
        pipe.describe()
    torch.cuda.synchronize()
    prof = eng.get_profile()
    eng.set_profiling(False)
    prof.sort(key=lambda e: -e["ms"])
    dom = prof[0]
    tot_ms = sum(e["ms"] for e in prof)
    roofline = roofline_of(dom, prof_steps, tot_ms, args.precision)
    n_img = 2 * pairs
    alg_flops_step = sum(algorithmic_flops_per_image(int(n), T) for n in np.diff(tb.cu_n))
    breakdown = {e["name"]: {"calls": e["calls"] // prof_steps, "ms": round(e["ms"] / prof_steps, 4),
                             "tflops": round(e["flops"] / max(e["ms"], 1e-9) / 1e9, 1) if e["flops"] else None}
                 for e in prof}

    out = {
        "metric": "line_descriptors_per_sec", "value": round(value, 1), "unit": "line-descriptors/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": SETTLE_STEPS,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32 (v_mfma_f32_32x32x2_f32)", "bf16x6": "f32 in/out; GEMMs as 6 bf16-split MFMA products, fp32 accumulate (fp32-faithful)",
                  "bf16x3": "f32 in/out; GEMMs as 3 bf16-split MFMA products, fp32 accumulate (~1e-5)",
                  "f16x3": "f32 in/out; GEMMs as 3 fp16-split MFMA products, fp32 accumulate (~1e-6)"}[args.precision],
        "precision": args.precision, "data": "synthetic",
        "gathered_rows_checked": gathered_ok,
        "config": {"workload": f"{args.workload}: {pairs} pairs/GPU of {W}x{H}, {n_lines} lines/image -> "
                               f"{int(tb.N / n_img)} sub-lines x {T} tokens, d_model=256, seeded weights",
                   "pairs_per_gpu": pairs, "descriptors_per_step": int(n_desc_step),
                   "collective": "all_gather(line_desc)" if world > 1 else "none"},
        "pair_match_ms": round(pair_match_ms, 4), "pair_match_latency_ms": round(pair_match_latency_ms, 4),
        "pair_latency_ms": round(pair_latency_ms, 3), "pair_latency_sync_ms": round(pair_latency_sync_ms, 3),
        "matches_per_step": n_matches,
        "whole_step_algorithmic_tflops": round(alg_flops_step / (ms_per_step * 1e-3) / 1e12 * (world if world > 1 else 1) / max(world, 1), 2),
        "gpu_ms_per_step_profiled": round(tot_ms / prof_steps, 3),
        "roofline": roofline, "kernels": breakdown,
    }
    if world == 1:  # the dense-map producer feeding this batch (reported beside the metric, never in it)
        out["producer"] = producer_section(eng, pipe, H, W, n_img, ms_per_step)
    if world == 1 and not args.no_alt_precisions:   # the same step in the other MFMA modes (few steps each), for reference
        alt = {}
        for mode in ("bf16x3", "f16x3", "bf16x6", "f32"):
            if mode == args.precision:
                continue
            eng.set_precision(mode)
            for _ in range(8):     # re-warm: clocks drop during the light single-pair / profiling sections above
                pipe.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(8):
                tb_a, _ld, _g = pipe.step()
            barrier()
            alt[mode] = round(tb_a.N * 8 / (time.perf_counter() - t0) * world, 1)
        eng.set_precision(args.precision)
        out["alt_precisions_desc_per_s"] = alt
    if world == 1 and not args.no_cpu_baseline:   # reported at N = 1 only (the contract), so scaling runs stay short
        cb = cpu_baseline(args.workload, args.cpu_budget)
        out["cpu_baseline"] = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in cb.items()}
        out["speedup_vs_cpu"] = round(value / cb["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
