#!/usr/bin/env python3
"""tools/kbench.py -- per-kernel timing of the row / column pass on one GPU (developer tool).
   BLUR_AMD_LIB=<variant .so> python tools/kbench.py [--rows R --cols C --sigma S --frames F --iters K --col-group G]
Prints one line: row/col kernel average ms (HIP events on the launch stream), frame rate, parity spot check."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import blur_algorithms_amd as B

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2160); ap.add_argument("--cols", type=int, default=3840)
ap.add_argument("--sigma", type=float, default=20.0); ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--iters", type=int, default=10); ap.add_argument("--col-group", type=int, default=0)
ap.add_argument("--wave-resident", default="auto", choices=["auto", "on", "off"]); ap.add_argument("--check", action="store_true"); ap.add_argument("--chunk", type=int, default=0); ap.add_argument("--row-major", action="store_true"); ap.add_argument("--tag", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (a.frames, a.rows, a.cols, 3), dtype=torch.uint8, device=dev, generator=g)
out = torch.empty_like(frames)
ctx = B.BlurContext(0)
for _ in range(3):
    ctx.pffft_(frames, a.sigma, out=out, col_group=a.col_group, frames_per_launch=a.chunk, row_major_planes=a.row_major, wave_resident={'auto': None, 'on': True, 'off': False}[a.wave_resident])
torch.cuda.synchronize()
ctx.timing_enable(True); ctx.timing(reset=True)
t0 = time.perf_counter()
for _ in range(a.iters):
    ctx.pffft_(frames, a.sigma, out=out, col_group=a.col_group, frames_per_launch=a.chunk, row_major_planes=a.row_major, wave_resident={'auto': None, 'on': True, 'off': False}[a.wave_resident])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tm = ctx.timing()
row = tm["row_ms"] / max(1, tm["row_frames"]); col = tm["col_ms"] / max(1, tm["col_frames"])   # per frame
px = a.rows * a.cols
msg = "%-28s %dx%d s=%g  row %.4f ms  col %.4f ms  sum %.4f ms  | wall/frame %.4f ms  %.0f MP/s  frac %.3f" % (
    a.tag or os.environ.get("BLUR_AMD_LIB", "default")[-28:], a.cols, a.rows, a.sigma, row, col, row + col,
    1e3 * dt / (a.iters * a.frames), a.iters * a.frames * px / 1e6 / dt, 30 * px / ((row + col) * 1e-3) / 8e12)
if a.check:
    from oracle import oracle as O
    img = frames[0].cpu().numpy()
    want, planes = O.pffft_blur_u8c3_f64(img, a.sigma, True, want_planes=True)
    got = out[0].cpu().numpy()
    d = got.astype(int) - want.astype(int)
    mism = d != 0
    v = np.moveaxis(planes.astype(np.float64), 0, -1) + 0.5
    dist = np.abs(v - np.round(v))
    msg += "  | max|d| %d  mism %d  worst tie dist %.2e" % (abs(d).max(), mism.sum(), dist[mism].max() if mism.any() else 0)
print(msg, flush=True)
