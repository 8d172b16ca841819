#!/usr/bin/env python3
"""bench.py -- megapixels/s of the FFT Gaussian blur (sigma=20, 4K RGB u8) on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (pffft_(), Source.cpp:429-570: row pass + column pass)
over this rank's batch of FRAMES synthetic 4K RGB frames that are already resident in HBM.
Frames are independent, so ranks shard the batch with no data-path collective (weak scaling:
FRAMES frames per GPU per step).  Rank 0 prints ONE JSON line.

Launch: with --gpus N > 1 and no WORLD_SIZE in the environment this script starts its own N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD process,
before anything in this process touches a GPU) and relays rank 0's line; it refuses when fewer than N devices are
visible.  `--rehearse` replaces the GPU step by a sleep so that the launch / reduce / report path runs on a CPU-only
box (tests/test_sharding_gloo.py).

Extra objects in that line:
  roofline     -- for the slower of the two kernels: algorithmic bytes per launch
                  (15 B/px x the pixels of the frames that launch covers: 3 u8 in + 12 f32 out
                  for the row pass, 12 + 3 for the column pass) / its average
                  launch duration from HIP events recorded on the launch stream inside the timed
                  region; peak 8 TB/s HBM3E; traffic = HBM bytes per launch from the rocprofv3
                  PMC passes committed under profiles/ (null until collected).
  cpu_baseline -- the CPU port of the same path (oracle/blur_oracle.c, float32, OpenMP, same
                  stage structure as the reference; NOT pffft) timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the CPU baseline's OpenMP threads stay where they start; set before any OpenMP runtime loads
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense f16 / bf16 MFMA (the 2:1-sparsity figure is not used)
ALG_BYTES_PER_PX_KERNEL = 15   # per kernel of a two-pass engine; 30 B/px for the frame
FUSED_BYTES_PER_PX = 6         # what the fused kernel has to move: 3 B/px in + 3 B/px out (+ 3 B/px read by the quirk's pre-pass)

# --config: the BASELINE.json configurations that fit one GPU (rows, cols, sigma, frames per step)
CONFIGS = {
    "metric": dict(rows=2160, cols=3840, sigma=20.0, frames=8, label="metric / C4 shard: 3840x2160 RGB u8, sigma=20"),
    "c2": dict(rows=1080, cols=1920, sigma=20.0, frames=8, label="C2: 1920x1080 RGB u8, sigma=20"),
    "c3": dict(rows=2160, cols=3840, sigma=50.0, frames=8, label="C3: 3840x2160 RGB u8, sigma=50"),
    "c5": dict(rows=4320, cols=7680, sigma=20.0, frames=1, label="C5: fastboxblur 7680x4320 RGB u8, 3 passes, k=41"),
}


def native_oracle():
    """the CPU port rebuilt on THIS host with -march=native (the shipped liboracle.so is x86-64-v3 so that the pinned
    functions round alike everywhere); falls back to the shipped library when no compiler is present"""
    import shutil
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle")
    cc = shutil.which("gcc")
    if not cc:
        return None
    out = os.path.join(tempfile.gettempdir(), "liboracle_native_%d.so" % os.getuid())
    cmd = [cc, "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", out,
           os.path.join(src, "blur_oracle.c"), os.path.join(src, "boxblur_oracle.c"), "-lm"]
    try:
        subprocess.run(cmd, check=True, capture_output=True, timeout=120)
        return out
    except Exception:
        return None


def cpu_baseline(rows, cols, sigma, budget_s=8.0):
    """time the CPU port on whole frames of the same workload for about `budget_s` seconds"""
    import numpy as np
    native = native_oracle()
    if native:
        os.environ["BLUR_ORACLE_LIB"] = native
    from oracle import oracle as O
    img = np.random.default_rng(0x5EED).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    O.pffft_blur_u8c3_f32(img, sigma)          # warm-up (page faults, OpenMP pool)
    n, t0 = 0, time.perf_counter()
    while True:
        O.pffft_blur_u8c3_f32(img, sigma)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s and n >= 3:
            break
    return {
        "value": round(n * rows * cols / 1e6 / dt, 2),
        "unit": "megapixels/s",
        "cores": O.num_threads(),
        "kind": "port",
        "sample": "%d frames of %dx%d RGB u8, sigma=%g, %.1f s wall, float32 OpenMP port of Source.cpp:429-570 (own radix-4/2/3/5 FFT, not pffft), "
                  "%s, OMP_PROC_BIND=%s" % (n, cols, rows, sigma, dt, "-march=native build on this host" if native else "shipped x86-64-v3 build",
                                           os.environ.get("OMP_PROC_BIND", "unset")),
    }


def baseline_metric():
    """the metric string of BASELINE.json (the driver matches on it)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "megapixels/sec Gaussian blur (\u03c3=20, 4K RGB) at 1/2/4/8 GPUs; % HBM roofline"


FAMILY_PREFIX = {0: "rowpass_kernel", 1: "fast_", 2: "wr_", 4: "mx_", 6: "fx_", 7: "wr_"}
FAMILY_NAME = {0: "run-time-planned FFT kernels", 1: "specialised rows-first FFT kernels", 2: "wave-resident FFT kernels", 3: "whole-image 2D FFT",
               4: "matrix-core kernels (two passes)", 6: "fused matrix-core kernel", 7: "tiled wave-resident FFT kernels"}
DTYPE = {4: "f16 hi+lo operands (u8 pixels exact, taps split in two binary16 halves), f32 accumulate, 24-bit fixed-point intermediate",
         6: "f16 hi+lo operands (u8 pixels exact, taps and intermediate split in two binary16 halves), f32 accumulate"}


def fused_launch_shape(rows, cols, pad, frames, num_cus=256):
    """what fx_launch_u8 (csrc/fx_kernels.hpp) launches for this call: window blocks, tasks, and the matrix instructions it executes"""
    nkb = next(n for n in (3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23) if 8 * (n - 2) >= pad)
    nt = (nkb - 1) // 2
    chunks, ntiles = -(-cols // 128), -(-rows // 32)
    wide = nkb > 11                                   # fw_kernels.hpp: one channel per workgroup, so three times the tasks
    stripes = chunks * frames * (3 if wide else 1)
    best = None
    for n in range(1, ntiles + 1):
        t = -(-ntiles // n)                          # (round 4: any number of tiles per segment)
        ns = -(-ntiles // t)
        span = -(-stripes * ns // num_cus) * (t + nt)
        if best is None or span < best[0]:
            best = (span, ns, t)
    _, nseg, tps = best
    steps = sum(min(tps, ntiles - sg * tps) + nt for sg in range(nseg))          # per strip of columns
    mfma_per_wave_step = (1 if wide else 3) * 5 * nkb                            # channels x (2 row + 3 column products) x window blocks
    # 4 waves per task; the prologue's first row pass; round 4: the first nt steps of a segment leave out the column products of the
    # tiles above it (nt^2 triples of three products per channel; SQ_INSTS_MFMA of profiles/r04fx_sq_counters.json agrees)
    mfmas = stripes * 4 * (steps * mfma_per_wave_step + nseg * 2 * nkb - nseg * (1 if wide else 3) * 3 * nt * nt)
    return {"nkb": nkb, "tasks": stripes * nseg, "segments_per_strip": nseg, "steps_per_strip": steps, "mfma_instructions": mfmas,
            "flops": mfmas * 2 * 32 * 32 * 16}


def pmc_traffic(role, frames_per_launch, family=2):
    """(kernel name, HBM bytes per launch) of the row / column kernel from the committed rocprofv3 PMC summary
    (separate FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950 corrections applied there); the
    bytes are None when the summary is missing or was taken with another batching"""
    prefix = FAMILY_PREFIX.get(family, "?")
    default = {"mx_": {"row": "mx_rowpass_u8", "col": "mx_colpass_u8"}, "wr_": {"row": "wr_rowpass_u8", "col": "wr_colpass_u8"},
               "fx_": {"blur": "fx_blur_u8"}}.get(prefix, {"row": "fast_rowpass3_u8", "col": "fast_colpass_u8"})[role]
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(path))
        name = [k for k in t if (role + "pass" in k or (prefix == "fx_" and "fx_blur" in k)) and k.startswith(prefix)][0]
        e = t[name]
        if abs(e["frames_per_launch"] - frames_per_launch) > 1e-9:
            return name, None
        return name, e["hbm_bytes_per_launch"]
    except Exception:
        return default, None


def self_launch(args, argv):
    """--gpus N without torchrun: start N ranks as a child torch.distributed.run and relay rank 0's JSON line.
    Nothing in THIS process may have initialised the GPU (device_count() does not, on this image)."""
    import socket
    import subprocess
    if not args.rehearse and not args.all_on_device0:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("--gpus %d but only %d device(s) visible" % (args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.run(cmd, env=env)
    raise SystemExit(p.returncode)


def percentiles(ms):
    """median, p10, p90 of per-step milliseconds"""
    a = sorted(ms)
    def q(f):
        if not a:
            return None
        x = f * (len(a) - 1)
        lo = int(x)
        hi = min(lo + 1, len(a) - 1)
        return a[lo] + (a[hi] - a[lo]) * (x - lo)
    return {"median": round(q(0.5), 4), "p10": round(q(0.1), 4), "p90": round(q(0.9), 4)}


def copy_bandwidth(ctx, mib=1024, reps=5):
    """the box's streaming rate in GB/s (bytes read + bytes written) for the library's own 16-byte-per-lane copy kernel
    (blur_copy_bandwidth), beside the 8 TB/s spec figure; MI355X_MICROARCH.md gives 6.29 TB/s for this access shape"""
    return ctx.copy_bandwidth(mib, reps)


def box_leg(ctx, torch, dev, steps, warmup, copy=False):
    """BASELINE config 5: fastboxblur (call site Source.cpp:587), 8K RGB u8, box width 41, 3 passes, in place, device-resident"""
    c = CONFIGS["c5"]
    rows, cols, k, passes = c["rows"], c["cols"], 41, 3
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0005)
    img = torch.randint(0, 256, (rows, cols, 3), dtype=torch.uint8, device=dev, generator=g)
    for _ in range(max(warmup, 3)):
        ctx.fastboxblur(img, k, passes)
    torch.cuda.synchronize(dev)
    # the contract's timed region without events (an event between two launches delays the second by about 3.5 us: 3 % here), then
    # the same K steps again with one event per step for the percentiles
    t0 = time.perf_counter()
    for i in range(steps):
        ctx.fastboxblur(img, k, passes)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    marks[0].record()
    for i in range(steps):
        ctx.fastboxblur(img, k, passes)
        marks[i + 1].record()
    torch.cuda.synchronize(dev)
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    px = rows * cols
    ms = 1e3 * dt / steps
    # The kernels as built move the image twice (all horizontal sweeps in one launch, all vertical ones in another): 12 B/px, and that
    # is what `achieved` / `frac` price (nothing in the record can exceed 1).  BASELINE.md section 3 prices a P-pass box blur at
    # 12 P B/px (every sweep reads and writes the image): kept as `two_pass_equiv`.
    moved, equiv = 12 * px, 12 * passes * px
    achieved = moved / (ms * 1e-3) / 1e9
    rec = {"metric": "megapixels/sec fastboxblur (8K RGB, 3-pass box k=41) at 1 GPU; % HBM roofline", "value": round(steps * px / 1e6 / dt, 1), "unit": "megapixels/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8 (i8 matrix-core sums in i32, 24-bit multiply-high rounding per sweep)", "data": "synthetic",
           "config": {"workload": c["label"] + ", in place, device-resident", "frames_per_gpu": 1},
           "ms_per_step_gpu": percentiles(per_step),      # of a second pass of the same K steps with one event per step
           "roofline": {"bound": "hbm", "kernel": "bx_horz3_kernel<1,3> + bx_vert_kernel<1,3,4> (whole call)", "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "alg_bytes_per_launch": moved,
                        "two_pass_equiv": {"bytes_per_launch": equiv, "achieved": round(equiv / (ms * 1e-3) / 1e9, 1),
                                           "frac": round(equiv / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}}
    if copy:
        rec["roofline"]["copy_peak"] = round(copy_bandwidth(ctx), 1)
    return rec


def box_config(args, torch, B):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = B.BlurContext(0)
    print(json.dumps(box_leg(ctx, torch, dev, args.steps, args.warmup, copy=not args.no_copy)), flush=True)


def reference_sweep(args, torch, B):
    """Source.cpp:627-635 / py/performance.ipynb:24: x = 1500, y = 1000, +225 / +150 per step, cv::resize(Size(y, x)) -> rows = x,
    cols = y, Test(img, flag, sqrt(x)).  Device-resident, one image per call, HIP events around the calls.  Each size also
    checks that the kernels it ran agree with the run-time-planned kernels of the same engine (at most one grey level apart,
    on fewer than 2e-3 of the bytes: rounding ties); parity against the float64 oracle per size is tests/test_gpu_sweep.py."""
    import ctypes as C
    import math
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = B.BlurContext(0)
    lib = B._lib.load()
    lib.blur_debug_last_family.argtypes = [C.c_void_p]
    lib.blur_debug_last_family.restype = C.c_int
    published_ms = [11.52, None]          # py/performance.ipynb:24, entry 1 (Apple M3 Pro, whole function incl. set-up)
    rows_out = []
    x, y = 1500, 1000
    for i in range(45):
        rows, cols, sigma = x, y, math.sqrt(x)
        img = torch.randint(0, 256, (rows, cols, 3), dtype=torch.uint8, device=dev)
        out = torch.empty_like(img)
        sz = B.pffft_sizing(rows, cols, sigma)
        rec = {"cols": cols, "rows": rows, "sigma": round(sigma, 2), "kSize": sz["kSize"], "N1": sz["N1"], "N0": sz["N0"]}
        try:
            for _ in range(2):
                ctx.pffft_(img, sigma, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                ctx.pffft_(img, sigma, out=out)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / n
            fam = lib.blur_debug_last_family(ctx._h)
            gen_out = torch.empty_like(img)
            gen = ctx.pffft_(img, sigma, out=gen_out, force_generic=True)
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            for _ in range(2):
                ctx.pffft_(img, sigma, out=gen_out, force_generic=True)
            g1.record()
            torch.cuda.synchronize(dev)
            rec["generic_ms"] = round(g0.elapsed_time(g1) / 2, 4)          # the run-time-planned FFT kernels on the same image, for comparison
            d = (out.to(torch.int16) - gen.to(torch.int16))
            d = (d + 128) % 256 - 128
            rec.update({"ms": round(ms, 4), "megapixels_per_s": round(rows * cols / 1e6 / (ms * 1e-3), 1),
                        "roofline_frac": round(30.0 * rows * cols / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "kernels": FAMILY_NAME.get(fam, "?"),
                        "max_abs_diff_vs_generic": int(d.abs().max().item()), "frac_diff_vs_generic": round(float((d != 0).float().mean().item()), 6)})
            rec["agrees"] = rec["max_abs_diff_vs_generic"] <= 1 and rec["frac_diff_vs_generic"] < 2e-3
        except B.BlurError as e:
            rec["error"] = str(e)
        rows_out.append(rec)
        del img, out
        x += 225
        y += 150
    ok = [r for r in rows_out if "ms" in r]
    print(json.dumps({"preset": "reference-sweep", "unit": "megapixels/s", "data": "synthetic",
                      "published_first_entry_ms_m3pro": published_ms[0],
                      "min_megapixels_per_s": min(r["megapixels_per_s"] for r in ok) if ok else None,
                      "all_agree": all(r.get("agrees", False) for r in ok), "sizes": rows_out}), flush=True)


class Runner:
    """fences and timed regions of one rank: W warm-up steps, then exactly K timed steps between two fences (barrier + synchronize
    on both sides), the maximum over ranks"""

    def __init__(self, torch, dist, dev, world, backend, ctx):
        self.torch, self.dist, self.dev, self.world, self.backend, self.ctx = torch, dist, dev, world, backend, ctx

    def fence(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def settle(self, step, seconds):
        """run the workload untimed for `seconds` (first touch of the workspaces, clock and power state)"""
        if seconds <= 0:
            return
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            for _ in range(4):
                step()
            self.torch.cuda.synchronize(self.dev)

    def timed(self, step, steps, warmup, events, step_events=False):
        """events: 0 none, 1 HIP events around every timed kernel launch, 2 around the dominant kernel only (blur_ctx_timing_enable);
        step_events: one event per step as well.  Returns (seconds, per-kernel timing or None, per-step milliseconds)"""
        torch, ctx = self.torch, self.ctx
        for _ in range(warmup):
            step()
        self.fence()
        if events:
            ctx.timing_enable(2 if events == 2 else True)
            ctx.timing(reset=True)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if step_events else []
        t0 = time.perf_counter()
        if step_events:
            marks[0].record()
        for i in range(steps):
            step()
            if step_events:
                marks[i + 1].record()
        self.fence()
        dt = time.perf_counter() - t0
        tm = ctx.timing(reset=True) if events else None
        ctx.timing_enable(False)
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)] if step_events else [1e3 * dt / steps]
        t = torch.tensor([dt], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item()), tm, per_step


def make_frames(torch, dev, g, kind, F, rows, cols):
    """F device-resident frames: i.i.d. uniform u8 or a committed natural-image crop tiled to the frame size"""
    if kind == "synthetic":
        return torch.randint(0, 256, (F, rows, cols, 3), dtype=torch.uint8, device=dev, generator=g)
    import numpy as np
    tile = np.load(os.path.join(ROOT, "tests", "golden", "img_collage_top.npz"))["src"]
    reps = (-(-rows // tile.shape[0]), -(-cols // tile.shape[1]), 1)
    one = torch.from_numpy(np.ascontiguousarray(np.tile(tile, reps)[:rows, :cols])).to(dev)
    return torch.stack([torch.roll(one, shifts=(17 * i, 31 * i), dims=(0, 1)) for i in range(F)]).contiguous()


def engine_text(B, family, rows, cols, sz, F):
    """what ran, in words (config.engine)"""
    import ctypes as C
    lib = B._lib.load()
    if family == 2:
        n_row, n_col = lib.blur_wr_length(cols + 2 * sz["pad"], 0), lib.blur_wr_length(rows + 2 * sz["pad"], 1)
        return "wave-resident FFT kernels, columns first, engine FFT lengths %d (rows) / %d (columns)" % (n_row, n_col)
    if family == 4:
        lib.blur_mx_window_blocks.argtypes = [C.c_int]
        lib.blur_mx_window_blocks.restype = C.c_int
        nkb = lib.blur_mx_window_blocks(sz["pad"])
        return ("%s: both passes as banded Toeplitz products on v_mfma_f32_32x32x16_f16 (window %d positions for %d taps, taps and "
                "intermediate in hi + lo binary16 halves, f32 accumulation), Nyquist-slot quirk as rank-one terms"
                % (FAMILY_NAME[family], 16 * nkb, sz["kSize"]))
    if family == 6:
        shape = fused_launch_shape(rows, cols, sz["pad"], F)
        return ("%s: row pass, register hand-off, column pass and byte emission in one launch on v_mfma_f32_32x32x16_f16 (window %d positions "
                "for %d taps; taps and intermediate in hi + lo binary16 halves, f32 accumulation; no intermediate in memory), Nyquist-slot quirk as "
                "rank-one terms from an integer pre-pass over the image" % (FAMILY_NAME[family], 16 * shape["nkb"], sz["kSize"]))
    return FAMILY_NAME.get(family, "?") + " at the reference's FFT lengths"


def roofline_record(family, tm, steps, rows, cols, sz, copy_gbs):
    """the `roofline` object for the dominant kernel of a Gaussian leg from the per-kernel HIP events of its instrumented pass"""
    px = rows * cols
    if family == 6 and tm and tm["row_launches"]:
        # One launch does both passes and keeps the intermediate on chip: the kernel is bound by the matrix pipe, not by HBM.
        # achieved = the matrix instructions the launch executes x 32768 flop / its average duration (HIP events on the launch
        # stream over the K steps of the instrumented pass); the HBM side is reported with what the kernel has to move (6 B/px).
        k_ms = tm["row_ms"] / tm["row_launches"]
        fpl = tm["row_frames"] / tm["row_launches"]
        shape = fused_launch_shape(rows, cols, sz["pad"], int(round(fpl)))
        name = ("fw_blur_u8<%d, %s>" if shape["nkb"] > 11 else "fx_blur_u8<%d, %s>") % (shape["nkb"], "true")
        achieved = shape["flops"] / (k_ms * 1e-3) / 1e12
        name2, traffic = pmc_traffic("blur", fpl, family) if shape["nkb"] <= 11 else (None, None)      # counters are committed for fx_blur_u8 only
        side_ms = tm["col_ms"] / max(steps, 1)                     # per step: the quirk's pre-pass (with the edge strips) and term kernel
        hbm_alg = FUSED_BYTES_PER_PX * px * fpl
        rl = {
            "bound": "mfma", "kernel": name, "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
            "avg_launch_ms": {name: round(k_ms, 4), "side kernels per step (the quirk's pre-pass with the edge strips; instrumented pass)": round(side_ms, 4)},
            "frames_per_launch": fpl,
            "flops_per_launch": shape["flops"], "mfma_instructions_per_launch": shape["mfma_instructions"], "tasks": shape["tasks"],
            "useful_flop_frac": round((2 * sz["pad"] + 1) / (16.0 * shape["nkb"]), 4),     # taps / window positions the products cover
            "hbm": {"alg_bytes_per_launch": hbm_alg, "achieved": round(hbm_alg / (k_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(hbm_alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "two_pass_alg_bytes_per_launch": 2 * ALG_BYTES_PER_PX_KERNEL * px * fpl,
                    "two_pass_frac": round(2 * ALG_BYTES_PER_PX_KERNEL * px * fpl / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if copy_gbs:
            rl["hbm"]["copy_peak"] = round(copy_gbs, 1)
        return rl
    if tm and tm["row_launches"] and tm["col_launches"]:
        row_ms = tm["row_ms"] / tm["row_launches"]
        col_ms = tm["col_ms"] / tm["col_launches"]
        fpl = tm["row_frames"] / tm["row_launches"]            # frames one launch covers
        role, dur = ("col", col_ms) if col_ms >= row_ms else ("row", row_ms)
        name, traffic = pmc_traffic(role, fpl, family)
        alg = ALG_BYTES_PER_PX_KERNEL * px * fpl
        achieved = alg / (dur * 1e-3) / 1e9
        rl = {
            "bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "avg_launch_ms": {pmc_traffic("row", fpl, family)[0]: round(row_ms, 4), pmc_traffic("col", fpl, family)[0]: round(col_ms, 4)},
            "frames_per_launch": fpl,
            "alg_bytes_per_launch": alg,
        }
        if copy_gbs:
            rl["copy_peak"] = round(copy_gbs, 1)           # measured streaming copy on this box, GB/s
            rl["frac_of_copy"] = round(achieved / copy_gbs, 4)
        return rl
    return None


def config_record(B, ctx, c, elapsed, tm, steps, copy_gbs):
    """compact record of one BASELINE configuration's short leg (the `configs` object of the default run)"""
    rows, cols, sigma, F = c["rows"], c["cols"], c["sigma"], c["frames"]
    px = rows * cols
    sz = B.pffft_sizing(rows, cols, sigma)
    family = ctx.last_family()
    rec = {"workload": "%s, %d frames per step, device-resident" % (c["label"], F), "value": round(steps * F * px / 1e6 / elapsed, 1), "unit": "megapixels/s",
           "ms_per_step": round(1e3 * elapsed / steps, 4),
           "frame_roofline_frac": round(steps * F * 2 * ALG_BYTES_PER_PX_KERNEL * px / elapsed / 1e9 / HBM_PEAK_GBS, 4),
           "engine": FAMILY_NAME.get(family, "?")}
    rl = roofline_record(family, tm, steps, rows, cols, sz, None)
    if rl:
        rec.update({"kernel": rl["kernel"], "bound": rl["bound"], "frac": rl["frac"], "achieved": rl["achieved"], "achieved_unit": rl["unit"],
                    "avg_launch_ms": rl["avg_launch_ms"]})
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (default 8; BASELINE C4: 64 frames over 8 GPUs)")
    ap.add_argument("--config", default="metric", choices=sorted(CONFIGS), help="which BASELINE.json configuration (metric = the headline one; "
                    "c2 1080p sigma 20, c3 4K sigma 50, c5 fastboxblur 8K): sets --rows / --cols / --sigma / --frames unless given")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--cols", type=int, default=None)
    ap.add_argument("--sigma", type=float, default=None)
    ap.add_argument("--col-group", type=int, default=0)
    ap.add_argument("--frames-per-launch", type=int, default=0, help="0 = the library's choice")
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "natural"],
                    help="synthetic: i.i.d. uniform u8 (worst case for round-off and for DVFS); natural: a committed crop of the "
                         "reference's test_images tiled to the frame size")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path on one GPU)")
    ap.add_argument("--all-on-device0", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--engine", default="auto", choices=["auto", "fused", "matrix", "fft", "wave-resident", "rows-first"],
                    help="auto: the library's choice (the fused matrix-core kernel where it applies); the others force an engine / FFT family")
    ap.add_argument("--wave-resident", default="auto", choices=["auto", "on", "off"],
                    help="A/B: the wave-resident kernels (columns first, N = 256 R0) or the rows-first kernels of round 1")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--no-step-events", action="store_true", help="do not record one event per step in the timed region (no ms_per_step_gpu percentiles)")
    ap.add_argument("--no-natural", action="store_true", help="skip the second (natural-image) timed run")
    ap.add_argument("--no-copy", action="store_true", help="skip the streaming-copy bandwidth measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip the short C2 / C3 / C5 legs of the default run (the `configs` object)")
    ap.add_argument("--settle", type=float, default=0.4,
                    help="seconds of the same workload run BEFORE the W warm-up steps (untimed): first-touch of the workspaces, "
                         "clock and power state; the first few milliseconds after an idle gap run 5-10 %% slower (DESIGN.md)")
    ap.add_argument("--preset", default="", choices=["", "reference-sweep"],
                    help="reference-sweep: the reference's own benchmark (Source.cpp:627-635): 45 RGB images 1000x1500 ... 7600x11400, "
                         "sigma = sqrt(longer side), one image per call; prints one JSON line with a row per size")
    ap.add_argument("--rehearse", action="store_true", help="no GPU work: the step is a sleep (launch / reduce / report path on a CPU-only box)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    for k in ("rows", "cols", "sigma", "frames"):
        if getattr(args, k) is None:
            setattr(args, k, cfg[k])
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])          # does not return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rows, cols, sigma, F = args.rows, args.cols, args.sigma, args.frames

    if args.rehearse:
        # launch / barrier / reduce / report path only: no GPU, no library; the "step" sleeps 1 ms
        if world > 1:
            dist.init_process_group("gloo")
        for _ in range(args.warmup):
            time.sleep(1e-3)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(1e-3)
        if world > 1:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": baseline_metric(), "value": round(world * args.steps * F * rows * cols / 1e6 / float(t.item()), 1),
                              "unit": "megapixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(1e3 * float(t.item()) / args.steps, 4), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "f32", "data": "rehearsal (no GPU work: sleep)", "rehearsal": True,
                              "config": {"workload": "rehearsal of the launch path", "frames_per_gpu": F}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    import blur_algorithms_amd as B
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.preset == "reference-sweep":
        return reference_sweep(args, torch, B)
    if args.config == "c5":
        return box_config(args, torch, B)
    if args.all_on_device0:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit("rank %d wants cuda:%d but only %d device(s) are visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED0000 + rank)
    ctx = B.BlurContext(local)
    wr = {"auto": None, "on": True, "off": False}[args.wave_resident]
    eng = None if args.engine == "auto" else args.engine
    if wr is not None and eng is None:
        eng = "wave-resident" if wr else "rows-first"          # the round-2 A/B switch still selects an FFT family
    run = Runner(torch, dist, dev, world, args.dist_backend, ctx)

    frames = make_frames(torch, dev, g, args.data, F, rows, cols)
    out = torch.empty_like(frames)

    def step_on(src, dst, sg, engine="default"):
        def step():
            ctx.pffft_(src, sg, out=dst, col_group=args.col_group, frames_per_launch=args.frames_per_launch, wave_resident=wr, engine=(eng if engine == "default" else engine))
        return step

    step = step_on(frames, out, sigma)
    run.settle(step, args.settle)                          # untimed, before the contract's W warm-up steps
    # An event between two kernels keeps the second from starting while the first drains (the fused engine's step is one big kernel
    # and the quirk's side kernels; measured per step on one box: 0.353 ms without events, 0.361 with the fused kernel's pair, 0.367
    # with every kernel's and one per step).  So the contract's timed region carries no events, and a second pass of the same K steps
    # right after it carries all of them: kernel durations for `roofline` (they agree with rocprofv3's kernel trace) and the
    # per-step percentiles (`instrumented_pass`).
    two_regions = not args.no_events
    elapsed, tm, per_step = run.timed(step, args.steps, args.warmup, 0, step_events=not two_regions and not args.no_step_events)
    instrumented_ms = None
    if two_regions:
        inst_elapsed, tm, per_step = run.timed(step, args.steps, 0, 1, step_events=not args.no_step_events)
        instrumented_ms = 1e3 * inst_elapsed / args.steps
    family = ctx.last_family()
    natural = None
    if args.data == "synthetic" and not args.no_natural:
        # the same workload on natural-image frames, with the same settle as the synthetic leg (round 3 ran it right after the host had
        # built the frames, with ten warm-up steps = 3.6 ms: the first ~30 ms after an idle gap run slow)
        nat = make_frames(torch, dev, g, "natural", F, rows, cols)
        nstep = step_on(nat, out, sigma)
        run.settle(nstep, args.settle)
        natural, _, _ = run.timed(nstep, args.steps, args.warmup, 0)
        del nat
    # the same workload on the FFT kernels (the path north_star describes), a shorter run, for the record
    fft_value = None
    if args.engine == "auto" and args.wave_resident == "auto" and not args.no_natural:
        fsteps = max(3, args.steps // 2)
        fft_elapsed, _, _ = run.timed(step_on(frames, out, sigma, engine="fft"), fsteps, 2, 0)
        fft_value = world * fsteps * F * rows * cols / 1e6 / fft_elapsed
    copy_gbs = None if (args.no_copy or rank != 0) else copy_bandwidth(ctx)

    # BASELINE.json's other single-GPU configurations, short legs of the same K steps (rank 0 of a 1-GPU run only; no CPU leg):
    # the driver runs `python bench.py --gpus 1` only, so this is where C2 / C3 / C5 reach its record
    configs = None
    if world == 1 and args.config == "metric" and args.engine == "auto" and args.wave_resident == "auto" and not args.no_configs \
            and (rows, cols, sigma) == (cfg["rows"], cfg["cols"], cfg["sigma"]):
        configs = {}
        for name in ("c2", "c3"):
            c = CONFIGS[name]
            cf = make_frames(torch, dev, g, "synthetic", c["frames"], c["rows"], c["cols"])
            co = torch.empty_like(cf)
            cstep = step_on(cf, co, c["sigma"])
            run.settle(cstep, min(args.settle, 0.2))
            c_el, _, _ = run.timed(cstep, args.steps, args.warmup, 0)
            _, c_tm, _ = run.timed(cstep, args.steps, 0, 1)
            configs[name] = config_record(B, ctx, c, c_el, c_tm, args.steps, copy_gbs)
            del cf, co
        b5 = box_leg(ctx, torch, dev, args.steps, args.warmup)
        configs["c5"] = {"workload": b5["config"]["workload"], "value": b5["value"], "unit": "megapixels/s", "ms_per_step": b5["ms_per_step"],
                         "kernel": b5["roofline"]["kernel"], "bound": "hbm", "frac": b5["roofline"]["frac"],
                         "achieved_GBs": b5["roofline"]["achieved"], "two_pass_equiv_frac": b5["roofline"]["two_pass_equiv"]["frac"]}
    # the cold number: no settle, 5 warm-up steps, then the first K steps after the GPU has been idle for a second (what a caller
    # sees who blurs one batch now and then; `value` is the steady state of back-to-back batches)
    cold = None
    if args.data == "synthetic" and not args.no_natural:
        run.fence()
        time.sleep(1.0)
        cold, _, _ = run.timed(step, args.steps, 5, 0)
    else:
        run.timed(step, 1, 0, 0)                            # leave the context on the default engine

    if rank == 0:
        px = rows * cols
        mp_total = world * args.steps * F * px / 1e6
        sz = B.pffft_sizing(rows, cols, sigma)
        rec = {
            "metric": baseline_metric(),
            "value": round(mp_total / elapsed, 1),
            "unit": "megapixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE.get(family, "f32"),
            "data": args.data,
            "config": {
                "workload": "%s: %dx%d RGB u8 frames, sigma=%g (kSize %d), reference FFT row/col lengths %d/%d, %d frames per GPU per step, device-resident"
                            % (cfg["label"] if (rows, cols, sigma) == (cfg["rows"], cfg["cols"], cfg["sigma"]) else "custom", cols, rows, sigma, sz["kSize"],
                               sz["N1"], sz["N0"], F),
                "engine": engine_text(B, family, rows, cols, sz, F),
                "frames_per_gpu": F,
                "sharding": "frames over ranks, no data-path collective",
            },
            "ms_per_step_gpu": percentiles(per_step),       # rank 0, HIP events around every step (fused engine: of the instrumented pass)
            "settle_s": args.settle,                        # untimed run of the same workload before the W warm-up steps
        }
        if instrumented_ms is not None:
            # the second pass of the same K steps with HIP events around every kernel and every step (they cost about 3.5 us each)
            rec["instrumented_pass"] = {"ms_per_step": round(instrumented_ms, 4), "value": round(mp_total / (instrumented_ms * 1e-3 * args.steps), 1)}
        if natural is not None:
            rec["value_natural"] = round(mp_total / natural, 1)      # same workload on natural-image frames (tests/golden crop, tiled), same settle
        if cold is not None:
            rec["value_cold"] = round(mp_total / cold, 1)            # no settle: 1 s idle, 5 warm-up steps, the first K steps
        if fft_value is not None:
            rec["value_fft_kernels"] = round(fft_value, 1)           # same workload, --engine fft (wave-resident FFT kernels)
        # the metric's "% HBM roofline" as BASELINE.md section 3 defines it: 30 algorithmic B/px of the two-pass algorithm / whole-job time / 8 TB/s
        rec["frame_roofline_frac"] = round((world * args.steps * F * 2 * ALG_BYTES_PER_PX_KERNEL * px / elapsed / 1e9) / (HBM_PEAK_GBS * world), 4)
        rl = roofline_record(family, tm, args.steps, rows, cols, sz, copy_gbs)
        if rl:
            rec["roofline"] = rl
        if configs:
            rec["configs"] = configs
        if world == 1 and not args.no_cpu:
            rec["cpu_baseline"] = cpu_baseline(rows, cols, sigma)
        print(json.dumps(rec), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
