"""tools/fx_fixed_cost.py -- the fused kernel's fixed cost per launch (GPU box, under rocprofv3 --kernel-trace):
8 frames of 3840 x H for several H (240 tasks of H / 32 + 5 steps each), quirk off and on; `--parse DIR` fits
duration = a + b steps over the kernel trace's dispatches.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/fx_fixed_cost.py ; python3 tools/fx_fixed_cost.py --parse DIR"""
import os
import sys

HS = (288, 544, 1088, 2176, 4352)
REPS = 24


def parse(d):
    import csv
    import glob
    import statistics
    f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    fused = [r for r in rows if "fx_blur_u8" in r["Kernel_Name"]]
    pre = [r for r in rows if "fx_prepass" in r["Kernel_Name"]]
    print("fused dispatches", len(fused), "prepass dispatches", len(pre))
    i = 0
    pts = {0: [], 1: []}
    for quirk in (0, 1):
        for h in HS:
            grp = fused[i:i + REPS]
            pg = pre[i:i + REPS]
            i += REPS
            med = statistics.median(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp[4:]) / 1e3
            pm = statistics.median(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in pg[4:]) / 1e3
            steps = (h + 31) // 32 + 5
            pts[quirk].append((steps, med))
            print("quirk %d  H %5d  steps %4d  fused %8.1f us  (%.3f us/step)  prepass %6.1f us" % (quirk, h, steps, med, med / steps, pm))
    for quirk in (0, 1):
        xs = [p[0] for p in pts[quirk]]
        ys = [p[1] for p in pts[quirk]]
        n = len(xs)
        mx, my = sum(xs) / n, sum(ys) / n
        b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
        a = my - b * mx
        print("quirk %d: duration = %.1f us + %.3f us x steps" % (quirk, a, b))


if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    parse(sys.argv[2])
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import blur_algorithms_amd as B

ctx = B.BlurContext(0)
g = torch.Generator(device="cuda").manual_seed(1)
for quirk in (False, True):
    for h in HS:
        fr = torch.randint(0, 256, (8, h, 3840, 3), dtype=torch.uint8, device="cuda", generator=g)
        out = torch.empty_like(fr)
        for _ in range(REPS):
            ctx.pffft_(fr, 20.0, out=out, nyquist_quirk=quirk, engine="fused")
        torch.cuda.synchronize()
        print("done", quirk, h, ctx.last_family(), flush=True)
