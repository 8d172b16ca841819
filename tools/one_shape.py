#!/usr/bin/env python3
"""tools/one_shape.py ROWS COLS SIGMA [FRAMES] [ENGINE] -- run pffft_ on one shape a few times (for rocprofv3 --kernel-trace --stats) and print the event time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blur_algorithms_amd as B
rows, cols, sigma = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 1
engine = sys.argv[5] if len(sys.argv) > 5 else None
ctx = B.BlurContext(0)
img = torch.randint(0, 256, (frames, rows, cols, 3), dtype=torch.uint8, device="cuda")
out = torch.empty_like(img)
for _ in range(2):
    ctx.pffft_(img, sigma, out=out, engine=engine)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    ctx.pffft_(img, sigma, out=out, engine=engine)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("%dx%d sigma %g frames %d: %.3f ms  %.0f MP/s  family %d  sizing %s" % (rows, cols, sigma, frames, ms, frames * rows * cols / 1e3 / ms, ctx.last_family(), B.pffft_sizing(rows, cols, sigma)))
