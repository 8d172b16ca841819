#!/usr/bin/env python3
"""time fastboxblur (BASELINE config 5: 8K RGB, k=41, 3 passes) on one GPU"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blur_algorithms_amd as B
h, w, k, p = 4320, 7680, 41, 3
img = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda")
ctx = B.BlurContext(0)
for _ in range(2):
    ctx.fastboxblur(img, k, p)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    ctx.fastboxblur(img, k, p)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
px = h * w
print("fastboxblur 8K k=%d passes=%d: %.3f ms/frame  %.0f MP/s  achieved %.0f GB/s of 12*P B/px (frac %.3f)" % (k, p, dt * 1e3, px / 1e6 / dt, 12 * p * px / dt / 1e9, 12 * p * px / dt / 8e12))
