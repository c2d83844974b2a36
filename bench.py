#!/usr/bin/env python3
"""Benchmark of the correlation hot path on MI355X (contract: see the task description).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config sintel|kitti_alt|things_train|1080p]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default (`--config sintel`) is BASELINE.json's configs[1], the configuration the metric is quoted on:
a *step* is one frame pair at Sintel 436x1024 (padded 440x1024, 1/8 resolution 55x128, N = 7040
queries), batch 1 per GPU, D = 256, fp32: one all-pairs build with the fused 4-level pyramid followed
by 32 canonical lookups (r = 4), all through the public ``CorrBlock(fmap1, fmap2, num_levels,
radius)(coords)`` API.  Inputs are synthetic and resident in HBM before the timed region starts.
The other configs of BASELINE.json run through the same harness:
  kitti_alt     configs[2]: KITTI 47x156, AlternateCorrBlock (on-the-fly), 12 lookups fwd + bwd
  things_train  configs[3]: FlyingThings crops 50x90, batch 6 per GPU, build + 12 lookups fwd + bwd;
                with N > 1 ranks each step ends with the all-reduce of a RAFT-sized gradient bucket
                (5,257,536 fp32, the DDP traffic of core/train.py's model) over RCCL
  1080p         configs[4]: 136x240, bf16 MFMA all-pairs build into a bf16 volume + 32 lookups; the float32-volume
                all-pairs builds and the on-the-fly variant are timed beside it (extras)

Multi-GPU: frame pairs are independent, so ranks run replicas with no data-path
collective ("weak" scaling).  `--gpus N` with N > 1 started WITHOUT torchrun re-launches itself as
N ranks in child processes (before anything touches the GPU) and relays rank 0's line; a rank count
that differs from --gpus is an error.  `value` and `n_gpus` come from the ranks that actually ran
(`ranks_seen`), never from the flag.

The JSON line carries, besides the contract fields:
  roofline      dominant kernel's achieved rate vs the MI355X peak (HIP events, live)
  kernels       the same figures for every kernel of the step
  extras        (N = 1 only, outside the timed region) the other kernels of the path
  cpu_baseline  the CPU oracle (stock PyTorch CPU ops in the reference's order) timed on this
                host on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32)
MFMA_16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 matrix
GRAD_BUCKET = 5_257_536        # parameters of RAFT-basic: the DDP all-reduce of config 3

CONFIGS = {
    #               image        H    W   B  iters  BASELINE.json index
    "sintel":       ((436, 1024), 55, 128, 1, 32, 1),
    "kitti_alt":    ((375, 1242), 47, 156, 1, 12, 2),
    "things_train": ((400, 720), 50, 90, 6, 12, 3),
    "1080p":        ((1088, 1920), 136, 240, 1, 32, 4),
}
D, LEVELS, RADIUS = 256, 4, 4
METRIC = "frame pairs/sec (node) + µs per corr build/lookup at Sintel 436×1024"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="sintel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--min-warm-seconds", type=float, default=0.5,
                    help="warm up until this much wall time has passed as well (clocks settle)")
    ap.add_argument("--device", choices=["cuda", "cpu"], default="cuda",
                    help="cpu = dry run of the launcher / rank plumbing only (no kernels, value is null)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None)
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without torchrun: start N ranks as fresh child processes (this process has not
    touched the GPU: nothing but argparse has run), relay their output and exit with their status."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd)
    raise SystemExit(proc.returncode)


# ------------------------------------------------------------------------------------------------
# workload definitions
# ------------------------------------------------------------------------------------------------

def work_of(cfg):
    """bytes / flops that the algorithm must move / execute per launch."""
    _, H, W, B, iters, _ = CONFIGS[cfg]
    N = H * W
    rd, T = 2 * RADIUS + 1, 2 * RADIUS + 2
    cells = sum((H >> l) * (W >> l) for l in range(LEVELS))
    w = {"N": N, "B": B, "build_flops": 2.0 * B * N * N * D,
         "build_bytes": 2.0 * B * N * D * 4 + 4.0 * B * N * cells,
         "build_bytes_16": 2.0 * B * N * D * 4 + 2.0 * B * N * cells,       # 16-bit volume
         "lookup_bytes": float(B * N * (LEVELS * T * T * 4 + LEVELS * rd * rd * 4 + 8)),
         "lookup_bytes_16": float(B * N * (LEVELS * T * T * 2 + LEVELS * rd * rd * 4 + 8)),
         "alt_flops": 2.0 * B * N * LEVELS * T * T * D,
         "alt_bytes": 4.0 * (B * N * D + B * cells * D + 2 * B * N + B * N * LEVELS * rd * rd),
         # lookup backward: read d(out), read-modify-write the (2r+2)^2 taps of every (query, level)
         "lookup_bwd_bytes": float(B * N * (LEVELS * rd * rd * 4 + 2 * LEVELS * T * T * 4 + 8)),
         # fold: read the whole gradient pyramid, write level 0
         "fold_bytes": 4.0 * B * N * (cells + H * W),
         "gemm_flops": 2.0 * B * N * N * D}
    w["alt_bwd_flops"] = 2.0 * w["alt_flops"]
    w["alt_bwd_bytes"] = w["alt_bytes"] + 4.0 * (B * N * D + B * cells * D)
    return w


def frac_entry(kernel, us, bound, amount, peak, unit, **kw):
    rate = amount / (us * 1e-6) / (1e12 if unit == "TFLOP/s" else 1e9)
    e = {"kernel": kernel, "bound": bound, "us_per_launch": us, "achieved": rate, "peak": peak, "unit": unit,
         "frac": rate / peak}
    e.update(kw)
    return e


def mfma16_entry(kernel, us, algorithmic_flops, terms, **kw):
    """A kernel whose contraction runs on the 16-bit matrix cores as `terms` products per algorithmic product (fp16x2: 3,
    bf16x3: 6): `frac` is EXECUTED flops against the dense 16-bit MFMA peak (the unit the kernel runs on, never > 1);
    the algorithmic rate is given beside it with its ratio to the fp32-matrix peak SURVEY 8(d) names (a ratio, not a
    roofline fraction: it exceeds 1 when the split arithmetic beats the fp32 unit)."""
    ex = terms * algorithmic_flops
    e = frac_entry(kernel, us, "mfma", ex, MFMA_16_PEAK_TFLOPS, "TFLOP/s", executed_flops=ex, mfma_terms=terms,
                   algorithmic_flops=algorithmic_flops, algorithmic_TFLOPs=algorithmic_flops / us / 1e6,
                   algorithmic_vs_fp32_mfma_peak=algorithmic_flops / us / 1e6 / MFMA_F32_PEAK_TFLOPS)
    e.update(kw)
    return e


def lookup_cap(w, vol_bytes=4):
    """Where the lookup's fraction ON ALGORITHMIC BYTES ends for this access pattern: a (2r+1)^2 window is
    fetched as the even-aligned (2r+4)^2 tile of 2x2 blocks; in lines of 4x8 cells (levels 0/1; 2x16 at levels 2/3, fp32)
    that tile spans 3.5 x 2.25 resp. 6 x 1.6 lines = 31.6 lines of 128 B per query over 4 levels, and a bare gather of
    such lines runs at 5.7 TB/s at best on this part (tools/gather_bw.hip): cap = algorithmic bytes / (line bytes +
    output bytes) x 5.7 / 8."""
    n = w["N"] * w.get("B", 1)
    lines = 31.6 * 128.0 * n * (vol_bytes / 4.0)
    out = n * LEVELS * (2 * RADIUS + 1) ** 2 * 4.0
    alg = w["lookup_bytes"] if vol_bytes == 4 else w["lookup_bytes_16"]
    return {"cap": alg / (lines + out) * 5.7 / 8.0,
            "cap_derivation": "algorithmic bytes / (31.6 lines x 128 B per query + output) x 5.7 TB/s line-gather rate / 8 TB/s peak; "
                              "the north star's 0.70 is not reachable with per-query slabs at fp32 (DESIGN 4.2)"}


def flow_coords(torch, B, H, W, iters, g, sigma=3.0):
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing="ij")
    base = torch.stack([xx, yy], 0)[None].repeat(B, 1, 1, 1)
    coords, drift = [], torch.zeros(B, 2, H, W)
    for _ in range(iters):
        coords.append((base + drift).contiguous())
        drift = drift + sigma * torch.randn(B, 2, H, W, generator=g)
    return coords


def make_inputs(torch, cfg, dev, seed):
    _, H, W, B, iters, _ = CONFIGS[cfg]
    g = torch.Generator(device="cpu").manual_seed(seed)
    f1 = torch.randn(B, D, H, W, generator=g).to(dev)
    f2 = torch.randn(B, D, H, W, generator=g).to(dev)
    coords = [c.to(dev) for c in flow_coords(torch, B, H, W, iters, g)]
    gouts = None
    if cfg in ("kitti_alt", "things_train"):
        gouts = [torch.randn(B, LEVELS * (2 * RADIUS + 1) ** 2, H, W, generator=g).to(dev) for _ in range(iters)]
    return f1, f2, coords, gouts


def ev_time_us(torch, fn, reps=20, warm=3):
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


def load_profile_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return {}


# ------------------------------------------------------------------------------------------------
# steps (one frame pair / one batch of pairs each)
# ------------------------------------------------------------------------------------------------

class Workload:
    """step(ev) runs one step; ev (or None) is a list of HIP events recorded at the step's phase
    boundaries on torch's current stream (= the stream the library launches on)."""
    n_events = 3
    needs_grad = False

    def __init__(self, torch, cfg, dev, rank, world, dist):
        self.torch, self.cfg, self.dev, self.world, self.dist = torch, cfg, dev, world, dist
        self.f1, self.f2, self.coords, self.gouts = make_inputs(torch, cfg, dev, 1234 + rank)


class SintelFwd(Workload):
    precision = "fp32"
    volume = "fp32"

    def step(self, ev):
        from raft_amd import CorrBlock
        if ev:
            ev[0].record()
        cb = CorrBlock(self.f1, self.f2, num_levels=LEVELS, radius=RADIUS, **self.kw())
        if ev:
            ev[1].record()
        out = None
        for c in self.coords:
            out = cb(c)
        if ev:
            ev[2].record()
        return out

    def kw(self):
        return {}

    def kernels(self, phases_us):
        w = work_of(self.cfg)
        iters = CONFIGS[self.cfg][4]
        build_us, lookup_us = phases_us[0], phases_us[1] / iters
        tsrc = "r03_pmc_traffic_sintel.json" if load_profile_json("r03_pmc_traffic_sintel.json") else "r02_pmc_traffic_sintel.json"
        traffic = load_profile_json(tsrc)
        kb = frac_entry("corr_build_f32_kernel (+ corr_pack_kernel)", build_us, "mfma", w["build_flops"],
                        MFMA_F32_PEAK_TFLOPS, "TFLOP/s", algorithmic_flops=w["build_flops"],
                        algorithmic_bytes=w["build_bytes"], hbm_GBs=w["build_bytes"] / build_us / 1e3,
                        traffic=traffic.get("corr_build_f32_kernel"), traffic_source="profiles/" + tsrc + " (PMC passes, not live)",
                        note="us_per_launch = HIP events around the constructor: pack + build kernels + host gap")
        kl = frac_entry("corr_lookup_fwd_kernel", lookup_us, "hbm", w["lookup_bytes"], HBM_PEAK_GBS, "GB/s",
                        algorithmic_bytes=w["lookup_bytes"], traffic=traffic.get("corr_lookup_fwd_kernel"),
                        traffic_source="profiles/" + tsrc + " (PMC passes, not live)", **lookup_cap(w))
        return {"build": kb, "lookup": kl}, ("build" if build_us >= phases_us[1] else "lookup")


class Bf16AllPairs(SintelFwd):
    def kw(self):
        return {"precision": "bf16", "volume_dtype": "bf16"}

    def kernels(self, phases_us):
        w = work_of(self.cfg)
        iters = CONFIGS[self.cfg][4]
        build_us, lookup_us = phases_us[0], phases_us[1] / iters
        kb = frac_entry("corr_build_bf16_kernel<1, bf16 volume> (+ corr_pack_bf16_kernel)", build_us, "hbm", w["build_bytes_16"],
                        HBM_PEAK_GBS, "GB/s", algorithmic_bytes=w["build_bytes_16"], algorithmic_flops=w["build_flops"],
                        mfma_TFLOPs=w["build_flops"] / build_us / 1e6, traffic=None,
                        note="bf16 operands, bf16 pyramid: bound by the pyramid's store traffic")
        kl = frac_entry("corr_lookup_fwd_kernel<4, bf16 volume>", lookup_us, "hbm", w["lookup_bytes_16"], HBM_PEAK_GBS, "GB/s",
                        algorithmic_bytes=w["lookup_bytes_16"], traffic=None, **lookup_cap(w, 2))
        return {"build": kb, "lookup": kl}, ("build" if build_us >= phases_us[1] else "lookup")


class KittiAlt(Workload):
    needs_grad = True
    tiles = "window"            # the config's coordinates are per-pixel random walks: query tiles by window position

    def __init__(self, *a):
        super().__init__(*a)
        self.f1.requires_grad_(True)
        self.f2.requires_grad_(True)

    def step(self, ev):
        from raft_amd import AlternateCorrBlock
        self.f1.grad = None
        self.f2.grad = None
        if ev:
            ev[0].record()
        blk = AlternateCorrBlock(self.f1, self.f2, num_levels=LEVELS, radius=RADIUS, tiles=self.tiles)
        outs = [blk(c) for c in self.coords]
        if ev:
            ev[1].record()
        self.torch.autograd.backward(outs, self.gouts)
        if ev:
            ev[2].record()
        return self.f1.grad

    def kernels(self, phases_us):
        w = work_of(self.cfg)
        iters = CONFIGS[self.cfg][4]
        fwd_us, bwd_us = phases_us[0] / iters, phases_us[1] / iters
        note = ("phase time / lookups: includes the feature pyramid and packs (fwd) and autograd glue (bwd); the product runs "
                "on the 16-bit matrix cores (3 fp16 terms per product): frac = executed flops (useful taps x 3) / 2.5 PF; the tile "
                "products cover whole boxes, i.e. the matrix cores execute several times that")
        tsrc = "r03_pmc_traffic_kitti.json" if load_profile_json("r03_pmc_traffic_kitti.json") else "r02_pmc_traffic_kitti.json"
        tr = load_profile_json(tsrc)
        k = {"alt_fwd": mfma16_entry("window groups (alt_wgroups_small_kernel) + alt_fwd16_kernel", fwd_us, w["alt_flops"], 3,
                                     compulsory_bytes=w["alt_bytes"], hbm_GBs=w["alt_bytes"] / fwd_us / 1e3,
                                     hbm_frac=w["alt_bytes"] / fwd_us / 1e3 / HBM_PEAK_GBS, traffic=tr.get("alt_fwd16_kernel"),
                                     traffic_source="profiles/" + tsrc, note=note),
             "alt_bwd": mfma16_entry("raftcorr_alt_lookup_bwd_many, one call for the pair's lookups: alt_bwd_pre + alt_bwd_kernel + "
                                     "alt_bwd_reduce + alt_bwd_ksum (per-lookup share; groups from the forward pass)", bwd_us,
                                     w["alt_bwd_flops"], 3,
                                     compulsory_bytes=w["alt_bwd_bytes"], hbm_GBs=w["alt_bwd_bytes"] / bwd_us / 1e3,
                                     hbm_frac=w["alt_bwd_bytes"] / bwd_us / 1e3 / HBM_PEAK_GBS,
                                     traffic=(tr.get("alt_bwd_kernel") or 0) // tr.get("alt_bwd_lookups_per_launch", 1) or None,
                                     traffic_source="profiles/" + tsrc, note=note)}
        return k, ("alt_bwd" if bwd_us >= fwd_us else "alt_fwd")


class ThingsTrain(Workload):
    needs_grad = True

    def __init__(self, *a):
        super().__init__(*a)
        self.f1.requires_grad_(True)
        self.f2.requires_grad_(True)
        from raft_amd.sharding import GradBucket
        # the DDP gradient bucket of the model around the path (config 3): a dummy of RAFT-basic's size, no-op at N = 1
        self.bucket = GradBucket(GRAD_BUCKET, self.dev)

    def step(self, ev):
        from raft_amd import CorrBlock
        self.f1.grad = None
        self.f2.grad = None
        if ev:
            ev[0].record()
        cb = CorrBlock(self.f1, self.f2, num_levels=LEVELS, radius=RADIUS)
        outs = [cb(c) for c in self.coords]
        if ev:
            ev[1].record()
        self.bucket.start()                     # async: overlaps the backward kernels, as DDP's buckets do
        self.torch.autograd.backward(outs, self.gouts)
        self.bucket.finish()
        if ev:
            ev[2].record()
        return self.f1.grad

    def kernels(self, phases_us):
        w = work_of(self.cfg)
        k = {"fwd": {"kernel": "corr_pack + corr_build_f32 + 12 x corr_lookup_fwd", "us": phases_us[0]},
             "bwd": {"kernel": "corr_record (12 lookups, one launch) + corr_splat_fold + 2 GEMMs (+ all-reduce when N > 1, overlapped)", "us": phases_us[1]}}
        k["fwd"].update(frac_entry(k["fwd"]["kernel"], phases_us[0], "mfma", w["build_flops"], MFMA_F32_PEAK_TFLOPS,
                                   "TFLOP/s"))
        k["bwd"].update(mfma16_entry(k["bwd"]["kernel"], phases_us[1], 2 * w["gemm_flops"], 3,
                                     note="the two feature-gradient GEMMs run as 3 fp16 products on the 16-bit matrix cores; the phase "
                                          "also holds the records and the fold-in-LDS kernel (HBM / LDS bound)"))
        return k, "bwd"


WORKLOADS = {"sintel": SintelFwd, "kitti_alt": KittiAlt, "things_train": ThingsTrain, "1080p": Bf16AllPairs}
DEFAULT_STEPS = {"sintel": (200, 20), "kitti_alt": (50, 5), "things_train": (30, 5), "1080p": (20, 3)}


# ------------------------------------------------------------------------------------------------
# extras: untimed side measurements (N = 1)
# ------------------------------------------------------------------------------------------------

def lookup_fracs(torch, dev):
    """corr_lookup_fwd_kernel at the shapes where the launch latency is amortised:
    Sintel B=1, Things B=6, 1080p; float32 and bf16 volume, canonical coordinates."""
    from raft_amd import CorrBlock
    out = {}
    g = torch.Generator().manual_seed(7)
    for cfg in ("sintel", "things_train", "1080p"):
        _, H, W, B, _, _ = CONFIGS[cfg]
        w = work_of(cfg)
        f1 = torch.randn(B, D, H, W, generator=g).to(dev)
        f2 = torch.randn(B, D, H, W, generator=g).to(dev)
        co = flow_coords(torch, B, H, W, 3, g)[2].to(dev)
        with torch.no_grad():
            for vol in ("fp32", "bf16"):
                cb = CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, precision="fp16x2", volume_dtype=vol)
                t = ev_time_us(torch, lambda: cb(co), reps=50, warm=5)
                nbytes = w["lookup_bytes"] if vol == "fp32" else w["lookup_bytes_16"]
                out[cfg + "_" + vol] = frac_entry("corr_lookup_fwd_kernel", t, "hbm", nbytes, HBM_PEAK_GBS, "GB/s",
                                                  algorithmic_bytes=nbytes, shape=f"B={B} {H}x{W}", volume=vol)
                del cb
        del f1, f2
        torch.cuda.empty_cache()
    return out


def extras_sintel(torch, wl, dev):
    from raft_amd import AlternateCorrBlock, CorrBlock
    cfg = wl.cfg
    w = work_of(cfg)
    f1, f2, coords = wl.f1, wl.f2, wl.coords
    iters = CONFIGS[cfg][4]
    out = {}
    with torch.no_grad():
        for prec, terms, note in (("fp16x2", 3, "fp32-grade: maps range-scaled, 2 fp16 pieces per operand, 3 products (parity <= 1e-5)"),
                                  ("bf16x3", 6, "fp32-grade: 3 bf16 pieces per operand, 6 products (parity <= 1e-5)"),
                                  ("bf16", 1, "operands rounded to bf16 (parity <= 1e-2)")):
            t = ev_time_us(torch, lambda: CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, precision=prec),
                           reps=100, warm=30)
            out["build_" + prec] = frac_entry(
                "corr_build_bf16_kernel (+ pack" + (" + max|x| pass)" if terms == 3 else ")"), t, "hbm",
                w["build_bytes"], HBM_PEAK_GBS, "GB/s", algorithmic_bytes=w["build_bytes"],
                algorithmic_TFLOPs=w["build_flops"] / t / 1e6, executed_16bit_TFLOPs=terms * w["build_flops"] / t / 1e6,
                mfma16_frac=terms * w["build_flops"] / t / 1e6 / MFMA_16_PEAK_TFLOPS, note=note)
        for prec, vol in (("bf16", "bf16"), ("fp16x2", "fp16")):
            t = ev_time_us(torch, lambda: CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, precision=prec, volume_dtype=vol),
                           reps=100, warm=30)
            out[f"build_{prec}_{vol}vol"] = frac_entry(
                "corr_build_bf16_kernel, 16-bit volume (+ pack)", t, "hbm", w["build_bytes_16"], HBM_PEAK_GBS, "GB/s",
                algorithmic_bytes=w["build_bytes_16"], algorithmic_TFLOPs=w["build_flops"] / t / 1e6,
                note="pyramid stored in 16 bits: parity <= 1e-2 (bf16) / 1e-3 (fp16)")

        def pair16():
            cb = CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, precision="bf16", volume_dtype="bf16")
            for c in coords:
                cb(c)
        t = ev_time_us(torch, pair16, reps=100, warm=10)
        out["step_bf16_bf16vol"] = {"frame_pairs_per_sec": 1e6 / t, "ms_per_step": t / 1e3, "steps": 100,
                                    "note": f"bf16 operands, bf16 volume (parity <= 1e-2), then the same {iters} lookups"}
        alt = AlternateCorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS)
        for tag, c in (("alt_fwd_iter0", coords[0]), ("alt_fwd_iter8", coords[8])):
            t = ev_time_us(torch, lambda: alt(c))
            out[tag] = mfma16_entry("alt_fwd16_kernel", t, w["alt_flops"], 3, compulsory_bytes=w["alt_bytes"],
                                    hbm_GBs=w["alt_bytes"] / t / 1e3, hbm_frac=w["alt_bytes"] / t / 1e3 / HBM_PEAK_GBS,
                                    note="identity flow" if tag.endswith("0") else "flow = 8 steps of N(0, 3^2) drift per pixel")

        def split_pair():
            cb = CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, precision="fp16x2")
            for c in coords:
                cb(c)
        n = 100
        t = ev_time_us(torch, split_pair, reps=n, warm=10)
        out["step_fp16x2"] = {"frame_pairs_per_sec": 1e6 / t, "ms_per_step": t / 1e3, "steps": n,
                              "note": "build with 2 fp16 pieces x 3 products (parity <= 1e-5, measured 2e-7), "
                                      f"then the same {iters} lookups"}
        big = torch.empty(1 << 28, dtype=torch.float32, device=dev)        # 1 GiB
        big2 = torch.empty_like(big)
        out["hbm_achievable_GBs"] = {"fill": big.numel() * 4 / ev_time_us(torch, lambda: big.zero_(), 5) / 1e3,
                                     "copy": 2 * big.numel() * 4 / ev_time_us(torch, lambda: big2.copy_(big), 5) / 1e3}
        del big, big2
    out["lookup_fwd_by_shape"] = lookup_fracs(torch, dev)
    return out


def extras_kitti(torch, wl, dev):
    """Per-launch times of the two on-the-fly kernels through the C ABI (no autograd glue), for a
    smooth and for a noisy flow field, against both rooflines of SURVEY 8(d)."""
    from raft_amd import AlternateCorrBlock, _lib
    lib = _lib.load()
    w = work_of(wl.cfg)
    _, H, W, B, _, _ = CONFIGS[wl.cfg]
    out = {}
    with torch.no_grad():
        blk = AlternateCorrBlock(wl.f1.detach(), wl.f2.detach(), num_levels=LEVELS, radius=RADIUS)
        f1g = wl.f1.detach().clone().requires_grad_(True)               # a block that packed the backward's copies
        with torch.enable_grad():
            gblk = AlternateCorrBlock(f1g, wl.f2.detach(), num_levels=LEVELS, radius=RADIUS)
        gout = wl.gouts[0]
        d1 = torch.empty(B, H, W, D, device=dev)
        d2 = [torch.empty(B, H >> l, W >> l, D, device=dev) for l in range(LEVELS)]
        dlv = _lib.ptr_array(d2)
        wb = lib.raftcorr_alt_lookup_bwd_workspace_bytes(B, H, W, D, LEVELS, RADIUS)
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        o = torch.empty(B, LEVELS * 81, H, W, device=dev)
        swb = lib.raftcorr_alt_lookup_fwd_sorted_workspace_bytes(B, H, W, LEVELS)
        swork = torch.empty(swb, dtype=torch.uint8, device=dev)
        wwb = lib.raftcorr_alt_lookup_bwd_sorted_workspace_bytes(B, H, W, D, LEVELS, RADIUS)
        wwork = torch.empty(wwb, dtype=torch.uint8, device=dev)
        for tag, c in (("identity", wl.coords[0]), ("sigma3", wl.coords[1]), ("sigma6", wl.coords[4]), ("sigma10", wl.coords[11])):
            tf = ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_fwd(blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(),
                                                                      B, H, W, D, LEVELS, RADIUS, c.data_ptr(), o.data_ptr(),
                                                                      stream))
            tb = ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_bwd(gblk._t1.data_ptr(), gblk._t2ptrs, gblk._scales.data_ptr(),
                                                                      B, H, W, D, LEVELS, RADIUS, c.data_ptr(), gout.data_ptr(),
                                                                      d1.data_ptr(), dlv, 0, work.data_ptr(), wb, stream))
            ts = ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_fwd_sorted(
                blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(), B, H, W, D, LEVELS, RADIUS, c.data_ptr(), o.data_ptr(),
                swork.data_ptr(), swb, stream))
            tw = ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_bwd_sorted(gblk._p1.data_ptr(), gblk._t2ptrs, gblk._scales.data_ptr(),
                                                                             B, H, W, D, LEVELS, RADIUS, c.data_ptr(), gout.data_ptr(),
                                                                             d1.data_ptr(), dlv, 0, None, wwork.data_ptr(), wwb, stream))
            tg = ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_bwd_sorted(gblk._p1.data_ptr(), gblk._t2ptrs, gblk._scales.data_ptr(),
                                                                             B, H, W, D, LEVELS, RADIUS, c.data_ptr(), gout.data_ptr(),
                                                                             d1.data_ptr(), dlv, 0, swork.data_ptr(), wwork.data_ptr(), wwb,
                                                                             stream))       # groups: what the forward call above left
            hb = dict(compulsory_bytes=w["alt_bytes"])
            out["alt_fwd_window_tiles_" + tag] = mfma16_entry("window groups (alt_wgroups_small_kernel; alt_wbucket x2 + alt_wscan beyond 8192 queries) + alt_fwd16_kernel", ts, w["alt_flops"], 3, **hb)
            out["alt_fwd_" + tag] = mfma16_entry("alt_fwd16_kernel", tf, w["alt_flops"], 3, hbm_GBs=w["alt_bytes"] / tf / 1e3,
                                                 hbm_frac=w["alt_bytes"] / tf / 1e3 / HBM_PEAK_GBS, **hb)
            hb = dict(compulsory_bytes=w["alt_bwd_bytes"])
            out["alt_bwd_" + tag] = mfma16_entry("alt_bwd_pre + alt_bwd + alt_bwd_reduce kernels", tb, w["alt_bwd_flops"], 3,
                                                 hbm_GBs=w["alt_bwd_bytes"] / tb / 1e3,
                                                 hbm_frac=w["alt_bwd_bytes"] / tb / 1e3 / HBM_PEAK_GBS, **hb)
            out["alt_bwd_window_groups_" + tag] = mfma16_entry(
                "window groups + alt_bwd_pre (windows, d(out) tiles, range pass, fmap1 gather) + alt_bwd + alt_bwd_reduce kernels", tw,
                w["alt_bwd_flops"], 3, hbm_GBs=w["alt_bwd_bytes"] / tw / 1e3,
                hbm_frac=w["alt_bwd_bytes"] / tw / 1e3 / HBM_PEAK_GBS, **hb)
            out["alt_bwd_window_groups_of_the_forward_" + tag] = mfma16_entry(
                "alt_bwd_pre (windows, d(out) tiles, range pass, fmap1 gather) + alt_bwd + alt_bwd_reduce kernels (groups from the lookup's forward call)", tg,
                w["alt_bwd_flops"], 3, hbm_GBs=w["alt_bwd_bytes"] / tg / 1e3,
                hbm_frac=w["alt_bwd_bytes"] / tg / 1e3 / HBM_PEAK_GBS, **hb)
        # the deferred backward pass: all lookups of the pair in ONE library call (what the timed step runs), groups of the
        # forward calls, the step's own random walks
        nl = len(wl.coords)
        if nl <= _lib.ALT_MAX_LOOKUPS:
            gws = []
            for c in wl.coords:
                g = torch.empty(swb, dtype=torch.uint8, device=dev)
                lib.raftcorr_alt_lookup_fwd_sorted(blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(), B, H, W, D, LEVELS, RADIUS,
                                                   c.data_ptr(), o.data_ptr(), g.data_ptr(), swb, stream)
                gws.append(g)
            for tag, window in (("window_groups", 1), ("pixel_tiles", 0)):
                mb = lib.raftcorr_alt_lookup_bwd_many_workspace_bytes(B, H, W, D, LEVELS, RADIUS, nl, window)
                mwork = torch.empty(mb, dtype=torch.uint8, device=dev)
                cp, dp, gp = _lib.ptr_array(wl.coords), _lib.ptr_array(wl.gouts), _lib.ptr_array(gws)
                f1 = gblk._p1 if window else gblk._t1
                tm = ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_bwd_many(
                    f1.data_ptr(), gblk._t2ptrs, gblk._scales.data_ptr(), B, H, W, D, LEVELS, RADIUS, nl, cp, dp, gp if window else None,
                    window, d1.data_ptr(), dlv, 0, mwork.data_ptr(), mb, stream), reps=10, warm=2)
                out["alt_bwd_many_%s_per_lookup" % tag] = mfma16_entry(
                    "raftcorr_alt_lookup_bwd_many over the step's %d lookups (alt_bwd_pre + alt_bwd + alt_bwd_reduce + alt_bwd_ksum), "
                    "time / lookups" % nl, tm / nl, w["alt_bwd_flops"], 3, compulsory_bytes=w["alt_bwd_bytes"],
                    hbm_GBs=w["alt_bwd_bytes"] / (tm / nl) / 1e3, hbm_frac=w["alt_bwd_bytes"] / (tm / nl) / 1e3 / HBM_PEAK_GBS)
                del mwork
    # the whole step with 8x8 pixel tiles in the forward pass (the default of AlternateCorrBlock: flow fields are smooth)
    wl.tiles = "pixel"
    try:
        for _ in range(3):
            wl.step(None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            wl.step(None)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n
        out["step_pixel_tiles"] = {"frame_pairs_per_sec": B / ms * 1e3, "ms_per_step": ms, "steps": n}
    finally:
        wl.tiles = type(wl).tiles
    return out


def extras_things(torch, wl, dev):
    """Per-launch times of the backward kernels through the C ABI at B = 6."""
    from raft_amd import CorrBlock, _lib
    lib = _lib.load()
    w = work_of(wl.cfg)
    _, H, W, B, _, _ = CONFIGS[wl.cfg]
    out = {}
    with torch.no_grad():
        cb = CorrBlock(wl.f1.detach(), wl.f2.detach(), num_levels=LEVELS, radius=RADIUS)
        pyr = cb._pyr
        pyr.grad_buffer()
        stream = torch.cuda.current_stream(dev).cuda_stream
        co, go = wl.coords[3], wl.gouts[3]
        t = ev_time_us(torch, lambda: cb(co), reps=50, warm=5)
        out["lookup_fwd"] = frac_entry("corr_lookup_fwd_kernel", t, "hbm", w["lookup_bytes"], HBM_PEAK_GBS, "GB/s",
                                       algorithmic_bytes=w["lookup_bytes"])
        t = ev_time_us(torch, lambda: lib.raftcorr_lookup_bwd(pyr.ptrs, pyr.dptrs, B, H, W, LEVELS, RADIUS,
                                                               co.data_ptr(), _lib.MODE_CANONICAL, go.data_ptr(), None,
                                                               stream), reps=50, warm=5)
        out["lookup_bwd"] = frac_entry("corr_lookup_bwd_kernel", t, "hbm", w["lookup_bwd_bytes"], HBM_PEAK_GBS, "GB/s",
                                       algorithmic_bytes=w["lookup_bwd_bytes"])
        wb = lib.raftcorr_build_bwd_workspace_bytes(B, D, H, W)
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        d1, d2 = torch.empty_like(wl.f1), torch.empty_like(wl.f2)
        t = ev_time_us(torch, lambda: lib.raftcorr_build_bwd(wl.f1.data_ptr(), wl.f2.data_ptr(), B, D, H, W, LEVELS,
                                                              pyr.dptrs, d1.data_ptr(), d2.data_ptr(), work.data_ptr(),
                                                              wb, stream), reps=10, warm=2)
        out["build_bwd"] = mfma16_entry("corr_fold_levels + cols_gather + 2 x gemm (fp16x2) + reduce + cols_scatter", t,
                                        2 * w["gemm_flops"], 3, fold_bytes=w["fold_bytes"],
                                        note="the accumulate route (raftcorr_lookup_bwd x K into a dense gradient pyramid, then this)")
        del work
        # the deferred route (what CorrBlock's backward runs at this shape): K records + one fold-in-LDS kernel + the GEMMs
        iters = CONFIGS[wl.cfg][4]
        C = LEVELS * (2 * RADIUS + 1) ** 2
        recs = [torch.empty(B, H * W, C, device=dev) for _ in range(iters)]
        t = ev_time_us(torch, lambda: lib.raftcorr_lookup_bwd_record(go.data_ptr(), B, H, W, C, recs[0].data_ptr(), stream),
                       reps=50, warm=5)
        out["lookup_bwd_record"] = frac_entry("corr_record_kernel", t, "hbm", 2.0 * go.numel() * 4, HBM_PEAK_GBS, "GB/s",
                                              algorithmic_bytes=2.0 * go.numel() * 4,
                                              note="d(out) (B, C, H, W) -> (B, N, C), once per lookup of the backward pass")
        for k in range(iters):
            _lib.check(lib.raftcorr_lookup_bwd_record(wl.gouts[k].data_ptr(), B, H, W, C, recs[k].data_ptr(), stream), "record")
        dwb = lib.raftcorr_build_bwd_deferred_workspace_bytes(B, D, H, W)
        dwork = torch.empty(dwb, dtype=torch.uint8, device=dev)
        cp, rp = _lib.ptr_array(wl.coords), _lib.ptr_array(recs)
        t = ev_time_us(torch, lambda: lib.raftcorr_build_bwd_deferred(wl.f1.data_ptr(), wl.f2.data_ptr(), B, D, H, W, LEVELS, RADIUS,
                                                                       _lib.MODE_CANONICAL, iters, cp, rp, None, d1.data_ptr(),
                                                                       d2.data_ptr(), dwork.data_ptr(), dwb, stream), reps=10, warm=2)
        d0_bytes = 4.0 * B * H * W * pyr.layout[0][2]
        out["build_bwd_deferred"] = mfma16_entry(
            "corr_splat_fold (all %d lookups, all levels) + cols_gather + 2 x gemm (fp16x2) + reduce + cols_scatter" % iters, t,
            2 * w["gemm_flops"], 3, splat_fold_bytes=iters * go.numel() * 4.0 + d0_bytes,
            note="replaces K x raftcorr_lookup_bwd + zero fill + fold: the folded level-0 gradient is made in LDS and written once")
    # the same step with the fp32-grade split-precision build (2 fp16 pieces x 3 products, parity <= 1e-5)
    from raft_amd import CorrBlock as _CB

    def split_step():
        wl.f1.grad = None
        wl.f2.grad = None
        cb2 = _CB(wl.f1, wl.f2, num_levels=LEVELS, radius=RADIUS, precision="fp16x2")
        torch.autograd.backward([cb2(c) for c in wl.coords], wl.gouts)
    for _ in range(3):
        split_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        split_step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    out["step_fp16x2_build"] = {"frame_pairs_per_sec": B / ms * 1e3, "ms_per_step": ms, "steps": n,
                                "note": "not the headline: the build runs as 3 fp16 products on the 16-bit matrix cores "
                                        "(fp32-grade, measured 2e-7 of max|C|); the headline keeps the plain fp32 MFMA build"}
    # the same step when d(out) arrives channels-last (a model run in torch.channels_last): the lookups' backward keeps
    # views instead of launching the 12 transposition passes
    saved = wl.gouts
    wl.gouts = [g.contiguous(memory_format=torch.channels_last) for g in saved]
    try:
        for _ in range(3):
            wl.step(None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            wl.step(None)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n
        out["step_channels_last_dout"] = {"frame_pairs_per_sec": B / ms * 1e3, "ms_per_step": ms, "steps": n,
                                          "note": "not the headline: BASELINE configs[3] hands d(out) over as contiguous NCHW"}
    finally:
        wl.gouts = saved
    prof = load_profile_json("r03_things_kernels.json") or load_profile_json("r02_things_kernels.json")
    psrc = "profiles/r03_things_kernels.json" if load_profile_json("r03_things_kernels.json") else "profiles/r02_things_kernels.json"
    for name, us in prof.items():                                # per-kernel durations from the committed trace
        if "splat_fold" in name:
            out["profile:" + name] = frac_entry(name, us, "hbm", iters * 4.0 * B * H * W * C + d0_bytes,
                                                HBM_PEAK_GBS, "GB/s", source=psrc,
                                                note="LDS / issue bound, not HBM bound (DESIGN 4.3): bytes = records read + dC0 written")
        elif "fold" in name:
            out["profile:" + name] = frac_entry(name, us, "hbm", w["fold_bytes"], HBM_PEAK_GBS, "GB/s", source=psrc)
        elif "gemm" in name and "reduce" not in name:
            out["profile:" + name] = mfma16_entry(name, us, w["gemm_flops"], 3, source=psrc)
    return out


def extras_1080p(torch, wl, dev):
    from raft_amd import AlternateCorrBlock, CorrBlock
    w = work_of(wl.cfg)
    iters = CONFIGS[wl.cfg][4]
    f1, f2, coords = wl.f1, wl.f2, wl.coords
    out = {}
    with torch.no_grad():
        for prec, vol in (("fp32", "fp32"), ("fp16x2", "fp32"), ("bf16", "fp32"), ("fp16x2", "fp16")):
            def pair():
                cb = CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, precision=prec, volume_dtype=vol)
                for c in coords:
                    cb(c)
            t = ev_time_us(torch, pair, reps=5, warm=1)
            tb = ev_time_us(torch, lambda: CorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, precision=prec, volume_dtype=vol),
                            reps=5, warm=1)
            lb = w["lookup_bytes"] if vol == "fp32" else w["lookup_bytes_16"]
            bb = w["build_bytes"] if vol == "fp32" else w["build_bytes_16"]
            terms = {"fp32": 0, "fp16x2": 3, "bf16": 1}[prec]
            out[f"all_pairs_{prec}_{vol}vol"] = {"frame_pairs_per_sec": 1e6 / t, "ms_per_pair": t / 1e3, "build_us": tb,
                                                 "build_TFLOPs": w["build_flops"] / tb / 1e6,
                                                 "build_mfma_frac": (w["build_flops"] / tb / 1e6 / MFMA_F32_PEAK_TFLOPS if terms == 0 else
                                                                     terms * w["build_flops"] / tb / 1e6 / MFMA_16_PEAK_TFLOPS),
                                                 "build_mfma_unit": "fp32 matrix (157.3 TF)" if terms == 0 else f"16-bit matrix (2500 TF), {terms} term(s) executed",
                                                 "build_hbm_frac": bb / tb / 1e3 / HBM_PEAK_GBS,
                                                 "lookup_us": (t - tb) / iters,
                                                 "lookup_hbm_frac": lb / ((t - tb) / iters) / 1e3 / HBM_PEAK_GBS}
            torch.cuda.empty_cache()
        for tiles in ("pixel", "window"):
            alt = AlternateCorrBlock(f1, f2, num_levels=LEVELS, radius=RADIUS, tiles=tiles)

            def alt_pair():
                for c in coords:
                    alt(c)
            t = ev_time_us(torch, alt_pair, reps=3, warm=1)
            out["on_the_fly" + ("" if tiles == "pixel" else "_window_tiles")] = {
                "frame_pairs_per_sec": 1e6 / t, "ms_per_pair": t / 1e3, "alt_fwd_us": t / iters,
                "alt_TFLOPs": w["alt_flops"] / (t / iters) / 1e6,
                "mfma16_frac": 3 * w["alt_flops"] / (t / iters) / 1e6 / MFMA_16_PEAK_TFLOPS,
                "algorithmic_vs_fp32_mfma_peak": w["alt_flops"] / (t / iters) / 1e6 / MFMA_F32_PEAK_TFLOPS,
                "hbm_frac": w["alt_bytes"] / (t / iters) / 1e3 / HBM_PEAK_GBS,
                "tiles": tiles + (" (8x8 pixel tiles; the coordinates are per-pixel random walks, the worst case for them)"
                                  if tiles == "pixel" else " (queries grouped by window position)"),
                "memory": "no volume (features only)"}
            del alt
    return out


EXTRAS = {"sintel": extras_sintel, "kitti_alt": extras_kitti, "things_train": extras_things, "1080p": extras_1080p}


# ------------------------------------------------------------------------------------------------
# CPU baseline (checker code timed beside the GPU -- never the product path)
# ------------------------------------------------------------------------------------------------

def host_cpu_info():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    pid = ln.split(":", 1)[1].strip()
                elif ln.startswith("core id"):
                    cid = ln.split(":", 1)[1].strip()
                elif not ln.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except OSError:
        pass
    return {"cpu_model": model, "host_physical_cores": len(phys) or None, "host_logical_cpus": os.cpu_count(),
            "cpu_share_of_this_job": int(os.environ.get("OMP_NUM_THREADS", "0")) or None}


def cpu_baseline(torch, cfg, seed, budget_s=15.0, cores=None, max_pairs=150):
    """Oracle (= the reference's operator sequence on stock PyTorch CPU ops) on the same workload,
    bounded sample.  cores: threads used = the job's CPU share on the GPU box (OMP_NUM_THREADS there;
    the host has more cores than one GPU's job may use)."""
    from oracle import corr_oracle as O
    cores = cores or int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    f1, f2, coords, gouts = make_inputs(torch, cfg, "cpu", seed)
    grad = gouts is not None
    iters = CONFIGS[cfg][4]

    def pair():
        if grad:
            a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
            cb = O.CorrBlockOracle(a, b, LEVELS, RADIUS)
            torch.autograd.backward([cb(c) for c in coords], gouts)
        else:
            with torch.no_grad():
                cb = O.CorrBlockOracle(f1, f2, LEVELS, RADIUS)
                for c in coords:
                    cb(c)

    with torch.no_grad():
        O.CorrBlockOracle(f1, f2, LEVELS, RADIUS)(coords[0])          # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        pair()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_pairs:
            break
    B = CONFIGS[cfg][3]
    info = host_cpu_info()
    info.update({"value": n * B / el, "unit": "frame pairs/sec", "cores": cores, "kind": "port",
                 "sample": f"{n} step(s) of the same workload ({cfg}: B={B}, build + {iters} lookups"
                           f"{' fwd+bwd' if grad else ''}, D={D}, fp32) in {el:.1f} s with torch.set_num_threads({cores})"})
    return info


def cpu_baseline_config0(torch, budget_s=8.0):
    """BASELINE.json configs[0]: the CPU CorrBlock (4 levels, r = 4), build + 12 lookups on the
    reference's demo-frames pair.  Features come from this repository's encoder with seed-0 random
    weights (raft-things.pth is not available offline); the frames are the fixture
    tests/golden/demo_frames.npz."""
    import numpy as np
    from oracle import corr_oracle as O
    from harness.raft_model import RAFT, InputPadder
    p = os.path.join(ROOT, "tests", "golden", "demo_frames.npz")
    if not os.path.exists(p):
        return None
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    with np.load(p, allow_pickle=False) as z:
        frames = [torch.from_numpy(z[k]).permute(2, 0, 1)[None].float() for k in sorted(z.files)[:2]]
    torch.manual_seed(0)
    model = RAFT(lambda a, b: None).eval()
    padder = InputPadder(frames[0].shape)
    i1, i2 = padder.pad(*frames)
    with torch.no_grad():
        f1, f2 = model.fnet([2 * (i1 / 255.0) - 1.0, 2 * (i2 / 255.0) - 1.0])
        f1, f2 = f1.float(), f2.float()
        B, _, H, W = f1.shape
        g = torch.Generator().manual_seed(0)
        coords = flow_coords(torch, B, H, W, 12, g)
        O.CorrBlockOracle(f1, f2, LEVELS, RADIUS)(coords[0])
        n, t0 = 0, time.perf_counter()
        while True:
            cb = O.CorrBlockOracle(f1, f2, LEVELS, RADIUS)
            for c in coords:
                cb(c)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s:
                break
    return {"value": n / el, "unit": "frame pairs/sec", "cores": cores, "kind": "port",
            "sample": f"{n} pair(s): demo frames 436x1024 -> {H}x{W} encoder features (seed-0 random weights), "
                      f"build + 12 lookups in {el:.1f} s"}


# ------------------------------------------------------------------------------------------------

def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    have_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_rank:
        self_launch(args)                       # never returns
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if have_rank else 1
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) were launched (WORLD_SIZE={world}); "
                         "launch with torch.distributed.run --nproc-per-node equal to --gpus, or without torchrun")
    steps = args.steps if args.steps is not None else DEFAULT_STEPS[args.config][0]
    warmup = args.warmup if args.warmup is not None else DEFAULT_STEPS[args.config][1]
    dry = args.device == "cpu"
    backend = args.backend or ("gloo" if dry else "nccl")

    import numpy as np
    import torch
    dist = None
    if dry:
        dev = torch.device("cpu")
    else:
        from raft_amd import _lib
        _lib.load()                             # fail loudly when the HIP extension is missing
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        from raft_amd.sharding import init_ranks
        dist = init_ranks(backend, dev)
    ranks_seen = dist.get_world_size() if dist is not None else 1

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    from raft_amd.sharding import max_over_ranks

    if dry:
        # launcher / rank plumbing only: no kernel runs (the product has no CPU path), no value is reported
        class Dry:
            n_events = 3

            def __init__(self):
                from raft_amd.sharding import GradBucket
                # config 3's collective runs in the dry mode too (gloo, CPU tensors): the same GradBucket calls as
                # ThingsTrain.step, so the N > 1 plumbing is executable without a GPU
                self.bucket = GradBucket(GRAD_BUCKET, dev) if args.config == "things_train" else None

            def step(self, ev):
                if self.bucket is not None:
                    self.bucket.start()
                time.sleep(0.001)
                if self.bucket is not None:
                    self.bucket.finish()
        wl = Dry()
    else:
        wl = WORKLOADS[args.config](torch, args.config, dev, rank, world, dist)

    # One step = one frame pair (or one batch of pairs): build, then the lookups.  The pair's block is a
    # local of the step, released before the next pair's is built -- as in RAFT.forward
    # (core/raft.py:103-129) -- so the caching allocator hands every build the same block.  HIP events
    # (phase split for the roofline figures) are recorded on every EV_EVERY-th step of the timed
    # region: each record costs ~3 us of GPU time.
    EV_EVERY = 4
    grad_ctx = torch.enable_grad if getattr(wl, "needs_grad", False) else torch.no_grad
    with grad_ctx():
        t_w = time.perf_counter()
        nwarm = 0
        while nwarm < warmup or (not dry and time.perf_counter() - t_w < args.min_warm_seconds):
            wl.step(None)
            nwarm += 1
            if nwarm % 8 == 0:
                sync()
        ev = [] if dry else [[torch.cuda.Event(enable_timing=True) for _ in range(wl.n_events)]
                             for _ in range((steps + EV_EVERY - 1) // EV_EVERY)]
        barrier()
        t0 = time.perf_counter()
        for s in range(steps):
            wl.step(ev[s // EV_EVERY] if (ev and s % EV_EVERY == 0) else None)
        barrier()
        elapsed = time.perf_counter() - t0

    elapsed = max_over_ranks(elapsed, dev)      # slowest rank defines the job time
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    (img_h, img_w), H, W, B, iters, bidx = CONFIGS[args.config]
    line = {"metric": METRIC, "value": None if dry else ranks_seen * steps * B / elapsed, "unit": "frame pairs/sec",
            "n_gpus": ranks_seen, "ranks_seen": ranks_seen, "steps": steps, "warmup": warmup, "warmup_steps_run": nwarm,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    if dry:
        line.update({"dry_run": True, "config": {"workload": "DRY RUN: launcher and rank plumbing only, no kernels"},
                     "allreduces_launched": getattr(getattr(wl, "bucket", None), "launched", 0)})
        print(json.dumps(line), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    phases = [float(np.median([e[i].elapsed_time(e[i + 1]) for e in ev])) * 1e3 for i in range(wl.n_events - 1)]
    kernels, dom = wl.kernels(phases)
    what = {"sintel": f"fp32 all-pairs build + {LEVELS}-level pyramid + {iters}-iter canonical lookup fwd (r={RADIUS})",
            "kitti_alt": f"AlternateCorrBlock (on-the-fly, query tiles by {KittiAlt.tiles} position in both passes), {iters} lookups fwd + bwd "
                         f"(r={RADIUS}, {LEVELS} levels)",
            "things_train": f"fp32 all-pairs build + {iters} lookups, fwd + bwd (deferred lookup backward)"
                            + (f" + all-reduce of {GRAD_BUCKET} fp32 (RCCL)" if world > 1 else ""),
            "1080p": f"bf16-operand MFMA all-pairs build (bf16 pyramid) + {iters}-iter lookup fwd"}[args.config]
    line["config"] = {"workload": f"{args.config}: {img_h}x{img_w} -> {H}x{W} (N={H * W}), batch {B}/GPU, D={D}, {what} "
                                  f"[BASELINE.json configs[{bidx}]]",
                      "parallelism": f"{ranks_seen} independent replica(s) (batch-sharded pairs"
                                     + (f", one async all-reduce per step of a DUMMY bucket of {GRAD_BUCKET} fp32 = "
                                        f"{GRAD_BUCKET * 4 / 1e6:.0f} MB (zeros, not gradients: DDP-sized traffic), issued before "
                                        "the backward kernels and waited for after them)"
                                        if args.config == "things_train" and world > 1 else ", no collective)")}
    if args.config == "1080p":
        line["dtype"] = "bf16"
    d = kernels[dom]
    line["roofline"] = {k: d.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                  "cap", "cap_derivation") if k in d or k in ("traffic",)}
    line["kernels"] = kernels
    if args.config in ("sintel", "1080p"):
        line["us_per_build"] = phases[0]
        line["us_per_lookup"] = phases[1] / iters
        line["build_plus_12_lookups_us"] = phases[0] + 12 * phases[1] / iters     # the north star's composite
    else:
        line["us_fwd"], line["us_bwd"] = phases[0], phases[1]
    if ranks_seen == 1 and not args.no_extras:
        line["extras"] = EXTRAS[args.config](torch, wl, dev)
        fill = line["extras"].get("hbm_achievable_GBs", {}).get("fill")
        if fill and "lookup" in kernels:
            kernels["lookup"]["frac_of_measured_fill_rate"] = kernels["lookup"]["achieved"] / fill
    if ranks_seen == 1 and not args.no_cpu_baseline:
        heavy = args.config in ("things_train", "1080p")
        if args.config == "1080p":
            line["cpu_baseline"] = {"value": None, "note": "the reference's CorrBlock needs 5.7 GB per level-0 volume "
                                    "and minutes per pair on CPU at this size; see --config sintel for the CPU baseline"}
        else:
            line["cpu_baseline_single_core"] = cpu_baseline(torch, args.config, 1234, budget_s=6.0, cores=1,
                                                            max_pairs=2 if heavy else 8)
            line["cpu_baseline"] = cpu_baseline(torch, args.config, 1234, budget_s=12.0 if heavy else 15.0)
            if args.config == "sintel":
                try:
                    line["cpu_baseline_config0"] = cpu_baseline_config0(torch)
                except Exception as e:                       # the side figure must not sink the bench line
                    line["cpu_baseline_config0"] = {"value": None, "error": repr(e)}
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
