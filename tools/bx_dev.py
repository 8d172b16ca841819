#!/usr/bin/env python3
"""fastboxblur on the integer matrix cores (csrc/bx_box.hip): bytes against oracle/boxblur_oracle.c on a list of shapes, then the time of BASELINE config 5
(8K RGB, k = 41, 3 passes) with and without those kernels.   tools/bx_dev.py [--no-check] [--shapes "w,h,c,k,p;..."]"""
import os, sys, time, argparse, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import blur_algorithms_amd as B

ap = argparse.ArgumentParser()
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--shapes", default="")
ap.add_argument("--time-only", action="store_true")
args = ap.parse_args()
ctx = B.BlurContext(0)
SHAPES = [(640, 480, 3, 41, 3), (333, 517, 3, 41, 3), (256, 300, 1, 9, 2), (128, 200, 4, 15, 1), (1920, 1080, 3, 41, 3), (1000, 700, 3, 49, 3), (600, 900, 3, 65, 2),
          (512, 400, 3, 3, 3), (512, 400, 3, 5, 3), (7680, 4320, 3, 41, 3)]
if args.shapes:
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in args.shapes.split(";")]
if not args.no_check and not args.time_only:
    from oracle import oracle as O
    for (w, h, c, k, p) in SHAPES:
        img = np.random.default_rng(w + h).integers(0, 256, (h, w, c), dtype=np.uint8)
        want = O.fastboxblur_u8(img, k, p)
        got = ctx.fastboxblur(torch.from_numpy(img.copy()).cuda(), k, p).cpu().numpy()
        bad = int((got != want).sum())
        msg = ""
        if bad:
            ys, xs, cs = np.nonzero(got != want)
            msg = "  first at row %d col %d ch %d: got %d want %d; rows %d..%d cols %d..%d; max |d| %d" % (ys[0], xs[0], cs[0], got[ys[0], xs[0], cs[0]], want[ys[0], xs[0], cs[0]],
                                                                                                ys.min(), ys.max(), xs.min(), xs.max(), int(np.abs(got.astype(int) - want).max()))
        print("w %5d h %5d c %d k %3d p %d: %s%s" % (w, h, c, k, p, "equal" if not bad else "%d bytes differ" % bad, msg), flush=True)
h, w, k, p = 4320, 7680, 41, 3
img = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.fastboxblur(img, k, p)
torch.cuda.synchronize()
n = 20
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    ctx.fastboxblur(img, k, p)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
ms = ts[n // 2]
print("fastboxblur 8K k=%d passes=%d: median %.3f ms (min %.3f)  %.0f MP/s  frac of 36 B/px at 8 TB/s: %.3f" % (k, p, ms, ts[0], h * w / 1e3 / ms, 36 * h * w / (ms * 1e-3) / 8e12), flush=True)
