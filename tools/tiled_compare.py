"""tools/tiled_compare.py [--wide] -- the reference's sweep sizes between 8 and 19 MP (one image per call, sigma = sqrt(longer side)): the library's own choice
against the tiled wave-resident path forced (tile_points = 4096); time and the largest byte difference between the two (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blur_algorithms_amd as B
ctx = B.BlurContext(0)
TALL = "--wide" not in sys.argv          # the reference's sweep (Source.cpp:627-635) has tall images: cv::Size(y, x) with x > y
for i in range(9, 18):
    rows, cols = (1500 + 225 * i, 1000 + 150 * i) if TALL else (1000 + 150 * i, 1500 + 225 * i)
    sigma = max(rows, cols) ** 0.5
    img = torch.randint(0, 256, (1, rows, cols, 3), dtype=torch.uint8, device="cuda")
    outs = []
    line = "%5d x %5d sigma %5.1f (%4.1f MP):" % (cols, rows, sigma, rows * cols / 1e6)
    for tp in (0, 4096):
        out = torch.empty_like(img)
        try:
            for _ in range(2): ctx.pffft_(img, sigma, out=out, tile_points=tp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # as bench.py --preset reference-sweep times it
            e0.record()
            for _ in range(5): ctx.pffft_(img, sigma, out=out, tile_points=tp)
            e1.record(); torch.cuda.synchronize(); ev = e0.elapsed_time(e1) / 5
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
            for _ in range(n): ctx.pffft_(img, sigma, out=out, tile_points=tp)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            line += "  %s %.3f ms [5 calls between events: %.3f] (%.1f GP/s, fam %d)" % ("forced" if tp else "auto  ", dt * 1e3, ev, rows * cols / 1e9 / dt, ctx.last_family())
            outs.append(out)
        except Exception as e:
            line += "  tile_points %d: %s" % (tp, str(e)[:60])
    if len(outs) == 2:
        line += "  max diff %d" % int((outs[0].int() - outs[1].int()).abs().max())
    print(line, flush=True)
