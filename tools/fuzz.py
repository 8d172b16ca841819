#!/usr/bin/env python3
"""tools/fuzz.py -- time-boxed random parity run on one GPU (developer tool; the oracle is the checker).
   python tools/fuzz.py [--seconds 240] [--seed 1] [--max-side 2600]
Draws (rows, cols, sigma, quirk, generic/specialised, batch) at random, runs the engine and the float64 oracle,
applies the parity contract of tests/conftest.py and stops at the first violation."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import blur_algorithms_amd as B
from oracle import oracle as O
from conftest import assert_u8_parity

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=240); ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-side", type=int, default=2600)
ap.add_argument("--targeted", action="store_true", help="aim one side at the FFT lengths that have specialised kernels")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
ctx = B.BlurContext(0)
t_end = time.time() + a.seconds
n = ties = 0
lengths = set()
last = time.time()
while time.time() < t_end:
    if a.targeted:
        # one side lands on a specialised length (as row or as column role), the other stays small (cheap oracle)
        n_target = int(rng.choice([4000, 2304, 4320, 1280, 2560]))
        sigma = float(rng.choice([2.0, 5.0, 11.0, 20.0, 33.0, 50.0]))
        long_side = None
        for _ in range(50):
            cand = int(rng.integers(max(8, n_target - 400), n_target))
            if O.pffft_sizing(cand, cand, sigma)["N0"] == n_target:
                long_side = cand
                break
        if long_side is None:
            continue
        short = int(rng.integers(8, 260))
        rows, cols = (short, long_side) if rng.random() < 0.5 else (long_side, short)
    else:
        big = rng.random() < 0.25
        hi = a.max_side if big else 500
        rows, cols = int(rng.integers(4, hi)), int(rng.integers(4, hi))
        sigma = float(np.exp(rng.uniform(np.log(0.3), np.log(60.0))))
    s = O.pffft_sizing(rows, cols, sigma)
    if s["pad"] > min(rows, cols) - 1:
        continue
    quirk = bool(rng.integers(0, 2)); generic = bool(rng.integers(0, 2)); nb = int(rng.choice([1, 1, 2, 3]))
    kind = rng.choice(["uniform", "binary", "smooth"])
    if kind == "uniform":
        imgs = rng.integers(0, 256, (nb, rows, cols, 3), dtype=np.uint8)
    elif kind == "binary":
        imgs = (rng.integers(0, 2, (nb, rows, cols, 3)) * 255).astype(np.uint8)
    else:
        yy, xx = np.mgrid[0:rows, 0:cols]
        imgs = np.stack([np.stack([(127.5 + 127.4 * np.sin(xx * rng.uniform(0.01, 0.5) + yy * rng.uniform(0.01, 0.5) + c)) for c in range(3)], -1)
                         for _ in range(nb)]).astype(np.uint8)
    t = torch.from_numpy(imgs).cuda()
    got = ctx.pffft_(t, sigma, out=torch.empty_like(t), nyquist_quirk=quirk, force_generic=generic).cpu().numpy()
    for i in range(nb):
        want, planes = O.pffft_blur_u8c3_f64(imgs[i], sigma, quirk, want_planes=True)
        try:
            ties += assert_u8_parity(got[i], want, planes)
        except AssertionError as e:
            print("FAIL rows=%d cols=%d sigma=%r quirk=%d generic=%d batch=%d frame=%d kind=%s: %s" % (rows, cols, sigma, quirk, generic, nb, i, kind, e), flush=True)
            sys.exit(1)
    lengths.update((s["N0"], s["N1"]))
    n += 1
    if time.time() - last > 30:
        print("  %d cases so far" % n, flush=True); last = time.time()
print("fuzz ok: %d cases, %d distinct FFT lengths, %d tie-break differences in total" % (n, len(lengths), ties), flush=True)
