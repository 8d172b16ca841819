#!/usr/bin/env python3
"""tools/mx_fuzz.py [seconds] [seed] -- the time-boxed fuzz of tests/test_gpu_matrix_engine.py, longer and wider: random shapes
(including large ones), every window size, both quirk settings, batches, in place / out of place, redzones; float64 oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import blur_algorithms_amd as B
from oracle import oracle as O
from conftest import assert_u8_parity

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = B.BlurContext(0)
t_end = time.time() + secs
cases = 0
while time.time() < t_end:
    sigma = float(rng.choice([0.6, 1.0, 2.0, 2.5, 4.0, 7.5, 9.0, 12.5, 15.0, 17.0, 20.0, 22.0, 24.0, 26.5, 29.0, 31.5, 34.0, 36.5, 39.0, 41.0, 44.0, 46.0, 48.0, 50.5]))
    pad = O.pffft_sizing(4096, 4096, sigma)["pad"]
    big = rng.random() < 0.15
    rows = int(rng.integers(pad + 1, pad + (1400 if big else 300)))
    cols = int(rng.integers(pad + 1, pad + (1800 if big else 500)))
    if rng.random() < 0.5:
        cols = (cols + 3) & ~3
    if O.pffft_sizing(rows, cols, sigma)["pad"] > min(rows, cols) - 1:
        continue
    quirk = bool(rng.integers(0, 2))
    nfr = int(rng.choice([1, 1, 2, 3]))
    kind = rng.choice(["uniform", "binary", "smooth"])
    if kind == "uniform":
        imgs = rng.integers(0, 256, (nfr, rows, cols, 3), dtype=np.uint8)
    elif kind == "binary":
        imgs = (rng.integers(0, 2, (nfr, rows, cols, 3)) * 255).astype(np.uint8)
    else:
        y, x = np.mgrid[0:rows, 0:cols]
        imgs = np.stack([np.clip(128 + 100 * np.sin(x / (7.0 + i))[..., None] * np.cos(y / 11.0)[..., None] + rng.normal(0, 5, (rows, cols, 3)), 0, 255).astype(np.uint8) for i in range(nfr)])
    guard = 4096
    n = imgs.size
    buf = torch.full((2 * n + 3 * guard,), 0xA5, dtype=torch.uint8, device="cuda")
    src = buf[guard:guard + n].view(nfr, rows, cols, 3)
    dst = buf[2 * guard + n:2 * guard + 2 * n].view(nfr, rows, cols, 3)
    src.copy_(torch.from_numpy(imgs))
    inplace = rng.random() < 0.3
    got = ctx.pffft_(src, sigma, out=src if inplace else dst, nyquist_quirk=quirk, engine="matrix").cpu().numpy()
    tag = "rows=%d cols=%d sigma=%r quirk=%d frames=%d kind=%s inplace=%d" % (rows, cols, sigma, quirk, nfr, kind, inplace)
    for i in range(nfr):
        want, planes = O.pffft_blur_u8c3_f64(imgs[i], sigma, quirk, want_planes=True)
        try:
            assert_u8_parity(got[i], want, planes)
        except AssertionError as e:
            print("FAIL", tag, "frame", i, str(e)[:200], flush=True)
            sys.exit(1)
    red = torch.cat([buf[:guard], buf[guard + n:2 * guard + n], buf[2 * guard + 2 * n:]])
    if int((red != 0xA5).sum()) != 0 or ctx._lib.blur_debug_check_workspace_guards(ctx._h) != 0:
        print("FAIL redzone", tag, flush=True)
        sys.exit(1)
    cases += 1
    if cases % 20 == 0:
        print("%d cases ok, last: %s" % (cases, tag), flush=True)
print("fuzz ok: %d cases in %.0f s" % (cases, secs))
