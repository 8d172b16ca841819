"""tools/mx_bench.py -- the matrix-core engine against the FFT engines on the metric's frame (GPU box).
usage: python tools/mx_bench.py [-m] [frames] [rows cols sigma]      (-m: the matrix engine only)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import blur_algorithms_amd as B

only = "-m" in sys.argv
if only:
    sys.argv.remove("-m")
fpl = 0
for a in list(sys.argv):
    if a.startswith("--fpl="):
        fpl = int(a[6:])
        sys.argv.remove(a)
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows, cols, sigma = (int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (2160, 3840, 20.0)
ctx = B.BlurContext(0)
g = torch.Generator(device="cuda").manual_seed(1)
frames = torch.randint(0, 256, (nf, rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty_like(frames)
for eng in (("matrix",) if only else ("matrix", "fft")):
    for quirk in (True, False, True):
        for _ in range(10):
            ctx.pffft_(frames, sigma, out=out, nyquist_quirk=quirk, engine=eng, frames_per_launch=fpl)
        torch.cuda.synchronize()
        ctx.timing_enable(True)
        ctx.timing()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            ctx.pffft_(frames, sigma, out=out, nyquist_quirk=quirk, engine=eng, frames_per_launch=fpl)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        t = ctx.timing()
        ctx.timing_enable(False)
        print("engine %-12s quirk %d: %.3f ms/step  %.1f GP/s   row %.1f us/frame  col %.1f us/frame" % (
            eng, quirk, dt * 1e3, nf * rows * cols / dt / 1e9, t["row_ms"] / max(t["row_frames"], 1) * 1e3, t["col_ms"] / max(t["col_frames"], 1) * 1e3), flush=True)
if only:
    sys.exit(0)
a = ctx.pffft_(frames.clone(), sigma, engine="matrix")
b = ctx.pffft_(frames.clone(), sigma)
d = (a.int() - b.int()).abs()
print("matrix vs fft: max diff %d, fraction differing %.2e" % (int(d.max()), float((d != 0).float().mean())))
