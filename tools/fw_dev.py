"""tools/fw_dev.py -- the wide fused matrix-core kernels (csrc/fw_kernels.hpp) against the float64 oracle (small shapes) and the two-kernel
matrix engine (large ones): bytes, then time on 8 x 4K frames for a few sigmas.   usage: python tools/fw_dev.py [--no-check] [--quirk0] [--fused-only] [--s50]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import blur_algorithms_amd as B

ctx = B.BlurContext(0)
quirk = "--quirk0" not in sys.argv
g = torch.Generator(device="cuda").manual_seed(1)
if "--no-check" not in sys.argv:
    from oracle import oracle as O
    from conftest import assert_u8_parity
    for rows, cols, sigma in ((300, 400, 25.0), (200, 332, 30.0), (400, 520, 50.0), (340, 132, 36.0), (180, 644, 44.0), (350, 256, 50.0)):
        img = np.random.default_rng(rows + cols).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
        want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=quirk, want_planes=True)
        got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk, engine="fused").cpu().numpy()
        d = np.abs(got.astype(int) - want.astype(int))
        try:
            assert_u8_parity(got, want, planes)
            verdict = "parity ok"
        except AssertionError as e:
            ys, xs, cs = np.nonzero(d > 1) if (d > 1).any() else np.nonzero(d)
            verdict = "FAILED: %s; first at row %d col %d ch %d; rows %d..%d cols %d..%d" % (str(e)[:80], ys[0], xs[0], cs[0], ys.min(), ys.max(), xs.min(), xs.max())
        print("%4d x %4d sigma %.1f pad %d: max diff %d, differing %.2e, family %d: %s" % (rows, cols, sigma, B.pffft_sizing(rows, cols, sigma)["pad"], int(d.max()),
              float((d != 0).mean()), ctx.last_family(), verdict), flush=True)
    for rows, cols, sigma, n in ((1080, 1920, 30.0, 2), (2160, 3840, 50.0, 2)):
        fr = torch.randint(0, 256, (n, rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g)
        a = ctx.pffft_(fr, sigma, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="matrix")
        b = ctx.pffft_(fr, sigma, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="fused")
        d = (a.int() - b.int()).abs()
        print("%4d x %4d sigma %.1f n %d vs the two-kernel engine: max diff %d, differing %.2e" % (rows, cols, sigma, n, int(d.max()), float((d != 0).float().mean())), flush=True)
rows, cols, nf = 2160, 3840, 8
frames = torch.randint(0, 256, (nf, rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty_like(frames)
for sigma in ((50.0,) if "--s50" in sys.argv else (26.0, 30.0, 36.0, 44.0, 50.0)):
    line = "4K x 8 sigma %.0f (pad %d):" % (sigma, B.pffft_sizing(rows, cols, sigma)["pad"])
    for eng in (("fused",) if ("--s50" in sys.argv or "--fused-only" in sys.argv) else ("fused", "matrix", "fft")):
        for _ in range(3):
            ctx.pffft_(frames, sigma, out=out, nyquist_quirk=quirk, engine=eng)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            ctx.pffft_(frames, sigma, out=out, nyquist_quirk=quirk, engine=eng)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        line += "  %s %.3f ms %.1f GP/s" % (eng, dt * 1e3, nf * rows * cols / 1e9 / dt)
    print(line, flush=True)
