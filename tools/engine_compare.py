"""tools/engine_compare.py -- time a few shapes with wide kernels on every engine (fused, the library's choice, two-kernel matrix, FFT)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import blur_algorithms_amd as B
ctx = B.BlurContext(0)
SHAPES = ((1000, 1500, 38.7, 1), (1600, 2400, 49.0, 1), (1080, 1920, 30.0, 1), (1080, 1920, 30.0, 8), (1080, 1920, 50.0, 8), (2160, 3840, 50.0, 1), (2160, 3840, 30.0, 1), (2160, 3840, 50.0, 8), (4320, 7680, 40.0, 1))
if "--sweep" in sys.argv:      # the first sizes of the reference's own benchmark (Source.cpp:627-635): one image per call, sigma = sqrt(cols)
    SHAPES = tuple((1500 + 225 * i, 1000 + 150 * i, (1500 + 225 * i) ** 0.5, 1) for i in range(12))      # tall: cv::Size(y, x), x > y
for rows, cols, sigma, nf in SHAPES:
    img = torch.randint(0, 256, (nf, rows, cols, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty_like(img)
    line = "%4d x %4d sigma %.1f n %d (pad %d):" % (rows, cols, sigma, nf, B.pffft_sizing(rows, cols, sigma)["pad"])
    for eng in ("fused", None, "matrix", "fft"):
        try:
            for _ in range(3): ctx.pffft_(img, sigma, out=out, engine=eng)
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
            for _ in range(n): ctx.pffft_(img, sigma, out=out, engine=eng)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            line += "  %s %.3f ms (%.1f GP/s, fam %d)" % (eng or "auto", dt * 1e3, nf * rows * cols / 1e9 / dt, ctx.last_family())
        except Exception as e:
            line += "  %s: %s" % (eng, str(e)[:40])
    print(line, flush=True)
