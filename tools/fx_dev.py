"""tools/fx_dev.py -- the fused matrix-core kernel against the two-kernel matrix engine (GPU box): bytes and time.
usage: python tools/fx_dev.py [--quirk] [--no-check] [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import blur_algorithms_amd as B

quirk = "--quirk" in sys.argv
check = "--no-check" not in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
nf = int(args[0]) if args else 8
ctx = B.BlurContext(0)
g = torch.Generator(device="cuda").manual_seed(1)
if check:
    for rows, cols, sigma, n in ((100, 100, 20.0, 1), (100, 100, 20.0, 2), (70, 68, 20.0, 2), (100, 300, 20.0, 1), (100, 128, 20.0, 1), (160, 100, 20.0, 1), (80, 200, 20.0, 1), (300, 72, 20.0, 1), (67, 256, 19.0, 2), (200, 300, 20.0, 1), (256, 384, 20.0, 2), (332, 516, 18.5, 1), (70, 68, 20.0, 1), (66, 132, 20.0, 3), (1080, 1920, 20.0, 1), (2160, 3840, 20.0, 2), (90, 4004, 19.0, 1)):
        fr = torch.randint(0, 256, (n, rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g)
        a = ctx.pffft_(fr, sigma, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="matrix")
        b = ctx.pffft_(fr, sigma, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="fused")
        d = (a.int() - b.int()).abs()
        bad = (d > 1).nonzero()
        print("%4d x %4d sigma %.1f n %d: max diff %d, differing %.2e, family %d%s" % (
            rows, cols, sigma, n, int(d.max()), float((d != 0).float().mean()), ctx.last_family(),
            "" if bad.numel() == 0 else "  first bad at %s (of %d)" % (bad[0].tolist(), bad.shape[0])), flush=True)
if check:
    # extreme images: alternating 0 / 255 columns, rows, checkerboard (the quirk's terms overflow the byte range: wrap semantics)
    for name, fn in (("cols", lambda y, x: (x & 1) * 255), ("rows", lambda y, x: (y & 1) * 255), ("checker", lambda y, x: ((x ^ y) & 1) * 255)):
        rows, cols = 300, 520
        y = torch.arange(rows, device="cuda").view(rows, 1, 1).expand(rows, cols, 3)
        x = torch.arange(cols, device="cuda").view(1, cols, 1).expand(rows, cols, 3)
        fr = fn(y, x).to(torch.uint8).contiguous().view(1, rows, cols, 3)
        a = ctx.pffft_(fr, 20.0, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="matrix")
        b = ctx.pffft_(fr, 20.0, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="fused")
        d = (a.int() - b.int()).abs()
        d = torch.minimum(d, 256 - d)
        print("extreme %-8s: max diff (mod 256) %d, differing %.2e, output range %d..%d" % (name, int(d.max()), float((d != 0).float().mean()), int(b.min()), int(b.max())), flush=True)
rows, cols, sigma = 2160, 3840, 20.0
frames = torch.randint(0, 256, (nf, rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty_like(frames)
for eng in (("fused", "fused") if "--fused-only" in sys.argv else ("fused", "matrix", "fused")):
    for _ in range(10):
        ctx.pffft_(frames, sigma, out=out, nyquist_quirk=quirk, engine=eng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        ctx.pffft_(frames, sigma, out=out, nyquist_quirk=quirk, engine=eng)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ctx.timing_enable(True)
    ctx.timing()
    for _ in range(n):
        ctx.pffft_(frames, sigma, out=out, nyquist_quirk=quirk, engine=eng)
    torch.cuda.synchronize()
    t = ctx.timing()
    ctx.timing_enable(False)
    print("engine %-8s quirk %d: %.3f ms/step  %.1f us/frame  %.1f GP/s   [events: slot0 %.1f us/frame, slot1 %.1f us/frame]" % (
        eng, quirk, dt * 1e3, dt / nf * 1e6, nf * rows * cols / dt / 1e9, t["row_ms"] / max(t["row_frames"], 1) * 1e3, t["col_ms"] / max(t["col_frames"], 1) * 1e3), flush=True)
