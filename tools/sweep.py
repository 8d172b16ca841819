#!/usr/bin/env python3
"""tools/sweep.py -- the reference's own benchmark shape (Source.cpp:627-635, README.md:137): 45 RGB images
from 1500x1000 to 11400x7600 (+225/+150 per step), true Gaussian, sigma = sqrt(longer side), one image per call.
Prints per-size milliseconds of the device-resident blur (and of the host-pointer call incl. PCIe) next to the
reference's published M3 Pro timings (py/performance.ipynb:24; whole-function wall time incl. setup).
   python tools/sweep.py [--steps 45] [--host]"""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import blur_algorithms_amd as B

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=45)
ap.add_argument("--host", action="store_true")
a = ap.parse_args()
ctx = B.BlurContext(0)
print("%-12s %6s %7s %11s %9s %9s %10s" % ("size", "sigma", "kSize", "N1/N0", "dev ms", "MP/s", "host ms" if a.host else ""))
x, y = 1500, 1000
for i in range(a.steps):
    rows, cols = x, y                     # cv::resize(Size(y, x)): y columns?  Source.cpp:630 passes Size(y, x) = (width=y, height=x)
    sigma = math.sqrt(x)
    img = torch.randint(0, 256, (rows, cols, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty_like(img)
    s = B.pffft_sizing(rows, cols, sigma)
    try:
        for _ in range(2):
            ctx.pffft_(img, sigma, out=out)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.pffft_(img, sigma, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        hs = ""
        if a.host:
            h = img.cpu().numpy()
            ctx.pffft_(h, sigma)
            t0 = time.perf_counter()
            ctx.pffft_(h, sigma)
            hs = "%10.2f" % ((time.perf_counter() - t0) * 1e3)
        print("%5dx%-6d %6.1f %7d %5d/%-5d %9.3f %9.0f %s" % (cols, rows, sigma, s["kSize"], s["N1"], s["N0"], dt * 1e3, rows * cols / 1e6 / dt, hs), flush=True)
    except B.BlurError as e:
        print("%5dx%-6d %6.1f %7d %5d/%-5d   %s" % (cols, rows, sigma, s["kSize"], s["N1"], s["N0"], e), flush=True)
    x += 225
    y += 150
