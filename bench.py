#!/usr/bin/env python3
"""Benchmark of the correlation hot path on MI355X (contract: see the task description).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one frame pair at BASELINE.json's configs[1]: Sintel 436x1024 (padded 440x1024,
1/8 resolution 55x128, N = 7040 queries), batch 1 per GPU, D = 256, fp32: one all-pairs build
with the fused 4-level pyramid followed by 32 canonical lookups (r = 4), all through the
public ``CorrBlock(fmap1, fmap2, num_levels, radius)(coords)`` API.  Inputs (features and the 32
coordinate fields) are synthetic and resident in HBM before the timed region starts.

Multi-GPU: frame pairs are independent, so ranks run replicas with no
data-path collective ("weak" scaling); the only communication is the barrier and the
max-over-ranks of the elapsed time.

The JSON line carries, besides the contract fields:
  roofline      dominant kernel's achieved rate vs the MI355X peak (HIP events, live)
  kernels       the same figures for both kernels of the step (build: MFMA, lookup: HBM)
  extras        (N = 1 only, outside the timed region) the other kernels of the path at the same
                shape -- bf16-operand build, on-the-fly lookup -- and the box's achievable HBM
                rates (fill / copy), so fractions can be read against nominal and achievable
  cpu_baseline  the CPU oracle (stock PyTorch CPU ops in the reference's order) timed on this
                host on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from raft_amd import AlternateCorrBlock, CorrBlock, _lib       # noqa: E402
from raft_amd.sharding import max_over_ranks   # noqa: E402

# workload: BASELINE.json configs[1]
IMG_H, IMG_W = 436, 1024
H, W = 55, 128                 # InputPadder -> 440 x 1024, /8
D, LEVELS, RADIUS, ITERS = 256, 4, 4, 32
B = 1

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32)


def algorithmic_work():
    """bytes/flops that the algorithm must move/execute per launch."""
    N = H * W
    rd, T = 2 * RADIUS + 1, 2 * RADIUS + 2
    pyr_cells = sum((H >> l) * (W >> l) for l in range(LEVELS))
    build_flops = 2.0 * B * N * N * D
    build_bytes = 2.0 * B * N * D * 4 + B * N * pyr_cells * 4
    lookup_bytes = B * N * (LEVELS * T * T * 4 + LEVELS * rd * rd * 4 + 8)
    return build_flops, build_bytes, float(lookup_bytes)


def make_inputs(dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    f1 = torch.randn(B, D, H, W, generator=g).to(dev)
    f2 = torch.randn(B, D, H, W, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing="ij")
    base = torch.stack([xx, yy], 0)[None].repeat(B, 1, 1, 1)
    coords, drift = [], torch.zeros(B, 2, H, W)
    for _ in range(ITERS):
        coords.append((base + drift).to(dev).contiguous())
        drift = drift + 3.0 * torch.randn(B, 2, H, W, generator=g)
    return f1, f2, coords


def one_pair(f1, f2, coords):
    cb = CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS)
    out = None
    for c in coords:
        out = cb(c)
    return out


def ev_time_us(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def extras(f1, f2, coords, dev):
    """Untimed side measurements at the workload's shape."""
    N, T = H * W, 2 * RADIUS + 2
    bf, bb, _ = algorithmic_work()
    out = {}
    with torch.no_grad():
        for prec, terms, note in (("fp16x2", 3, "fp32-grade: maps range-scaled, 2 fp16 pieces per operand, 3 products (parity <= 1e-5)"),
                                  ("bf16x3", 6, "fp32-grade: 3 bf16 pieces per operand, 6 products (parity <= 1e-5)"),
                                  ("bf16", 1, "operands rounded to bf16 (parity <= 1e-2)")):
            t = ev_time_us(lambda: CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, precision=prec), reps=100, warm=30)
            out["build_" + prec] = {"kernel": "corr_build_bf16_kernel (+ pack" + (" + max|x| pass)" if terms == 3 else ")"),
                                    "us_per_launch": t, "bound": "hbm", "achieved": bb / t / 1e3,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bb / t / 1e3 / HBM_PEAK_GBS,
                                    "algorithmic_bytes": bb, "algorithmic_TFLOPs": bf / t / 1e6,
                                    "executed_16bit_TFLOPs": terms * bf / t / 1e6, "note": note}
        alt = AlternateCorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS)
        alt_flops = 2.0 * B * N * LEVELS * T * T * D
        alt_bytes = 4.0 * (B * N * D + sum(B * (H >> l) * (W >> l) * D for l in range(LEVELS)) + 2 * B * N +
                           B * N * LEVELS * (2 * RADIUS + 1) ** 2)
        for tag, c in (("alt_fwd_iter0", coords[0]), ("alt_fwd_iter8", coords[8])):
            t = ev_time_us(lambda: alt(c))
            out[tag] = {"kernel": "alt_tile_fwd_kernel", "us_per_launch": t,
                        "algorithmic_TFLOPs": alt_flops / t / 1e6, "compulsory_GBs": alt_bytes / t / 1e3,
                        "note": "identity flow" if tag.endswith("0") else "flow = 8 steps of N(0, 3^2) drift per pixel"}
        # the whole step again with the fp32-grade split build (CorrBlock(..., precision="fp16x2")): same
        # workload and parity bar as the headline number, which stays on the plain fp32 MFMA build
        def split_pair():
            cb = CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, precision="fp16x2")
            for c in coords:
                cb(c)
        n = 100
        t = ev_time_us(split_pair, reps=n, warm=10)
        out["step_fp16x2"] = {"frame_pairs_per_sec": 1e6 / t, "ms_per_step": t / 1e3, "steps": n,
                              "note": "build with 2 fp16 pieces x 3 products (parity <= 1e-5, measured 2e-7), "
                                      f"then the same {ITERS} lookups"}
        big = torch.empty(1 << 28, dtype=torch.float32, device=dev)        # 1 GiB
        big2 = torch.empty_like(big)
        out["hbm_achievable_GBs"] = {"fill": big.numel() * 4 / ev_time_us(lambda: big.zero_(), 5) / 1e3,
                                     "copy": 2 * big.numel() * 4 / ev_time_us(lambda: big2.copy_(big), 5) / 1e3}
    return out


def cpu_baseline(seed, budget_s=15.0, cores=None, max_pairs=150):
    """Oracle (= the reference's operator sequence on stock PyTorch CPU ops) on the same
    workload, bounded sample.  Checker code, timed beside the GPU -- never the product path."""
    from oracle import corr_oracle as O
    cores = cores or int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    f1, f2, coords = make_inputs("cpu", seed)
    with torch.no_grad():
        O.CorrBlockOracle(f1, f2, LEVELS, RADIUS)(coords[0])          # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            cb = O.CorrBlockOracle(f1, f2, LEVELS, RADIUS)
            for c in coords:
                cb(c)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= max_pairs:
                break
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), model)
    except OSError:
        pass
    return {"value": n / el, "unit": "frame pairs/sec", "cores": cores, "kind": "port", "cpu_model": model,
            "sample": f"{n} pair(s) of the same workload (build + {ITERS} lookups, 55x128, D=256, fp32) "
                      f"in {el:.1f} s with torch.set_num_threads({cores})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    _lib.load()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    f1, f2, coords = make_inputs(dev, 1234 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # One step = one frame pair: build, then ITERS lookups.  The pair's CorrBlock is a local of the
    # step, released before the next pair's is built -- as in RAFT.forward (core/raft.py:103-129),
    # where corr_fn lives for one forward pass -- so the caching allocator hands every build the same
    # 261 MB block.  HIP events (build / lookup split for the roofline figures) are recorded on every
    # EV_EVERY-th step of the timed region: each record costs ~3 us of GPU time.
    EV_EVERY = 4

    def step(e):
        if e is not None:
            e[0].record()
        cb = CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS)
        if e is not None:
            e[1].record()
        out = None
        for c in coords:
            out = cb(c)
        if e is not None:
            e[2].record()
        return out

    with torch.no_grad():
        for _ in range(args.warmup):
            step(None)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)]
              for _ in range((args.steps + EV_EVERY - 1) // EV_EVERY)]
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            step(ev[s // EV_EVERY] if s % EV_EVERY == 0 else None)
        barrier()
        elapsed = time.perf_counter() - t0

    elapsed = max_over_ranks(elapsed, dev)      # slowest rank defines the job time

    build_ms = float(np.median([e[0].elapsed_time(e[1]) for e in ev]))
    lookup_ms = float(np.median([e[1].elapsed_time(e[2]) for e in ev])) / ITERS
    bf, bb, lb = algorithmic_work()
    k_build = {"kernel": "corr_build_f32_kernel", "bound": "mfma", "us_per_launch": build_ms * 1e3,
               "achieved": bf / (build_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
               "algorithmic_flops": bf, "algorithmic_bytes": bb,
               "hbm_GBs": bb / (build_ms * 1e-3) / 1e9}
    k_build["frac"] = k_build["achieved"] / k_build["peak"]
    k_lookup = {"kernel": "corr_lookup_fwd_kernel", "bound": "hbm", "us_per_launch": lookup_ms * 1e3,
                "achieved": lb / (lookup_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "algorithmic_bytes": lb}
    k_lookup["frac"] = k_lookup["achieved"] / k_lookup["peak"]
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # per-launch HBM bytes from rocprofv3 PMC
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))
    k_build["traffic"] = traffic.get("corr_build_f32_kernel")
    k_lookup["traffic"] = traffic.get("corr_lookup_fwd_kernel")
    dom = k_build if build_ms >= lookup_ms * ITERS else k_lookup
    roofline = {k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")}

    if rank == 0:
        line = {
            "metric": "frame pairs/sec (node) + µs per corr build/lookup at Sintel 436×1024",
            "value": args.gpus * args.steps * B / elapsed, "unit": "frame pairs/sec",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Sintel {IMG_H}x{IMG_W} -> {H}x{W} (N={H*W}), batch {B}/GPU, D={D}, fp32 "
                                   f"all-pairs build + {LEVELS}-level pyramid + {ITERS}-iter canonical lookup "
                                   f"fwd (r={RADIUS}) [BASELINE.json configs[1]]",
                       "parallelism": f"{args.gpus} independent replicas (batch-sharded pairs, no collective)"},
            "us_per_build": build_ms * 1e3, "us_per_lookup": lookup_ms * 1e3,
            "roofline": roofline, "kernels": {"build": k_build, "lookup": k_lookup},
        }
        if args.gpus == 1:
            line["extras"] = extras(f1, f2, coords, dev)
            fill = line["extras"]["hbm_achievable_GBs"]["fill"]          # measured on this box, same process
            k_lookup["frac_of_measured_fill_rate"] = k_lookup["achieved"] / fill
        line["build_plus_12_lookups_us"] = build_ms * 1e3 + 12 * lookup_ms * 1e3     # the north star's composite
        if args.gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline_single_core"] = cpu_baseline(1234, budget_s=6.0, cores=1, max_pairs=8)
            line["cpu_baseline"] = cpu_baseline(1234)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
