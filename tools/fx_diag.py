import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blur_algorithms_amd as B
ctx = B.BlurContext(0)
g = torch.Generator(device="cuda").manual_seed(1)
np.set_printoptions(linewidth=250)
quirk = "--quirk" in sys.argv
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(70, 68)]
for rows, cols in shapes:
    fr = torch.randint(0, 256, (1, rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g)
    a = ctx.pffft_(fr, 20.0, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="matrix")[0].cpu().numpy().astype(int)
    b = ctx.pffft_(fr, 20.0, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="fused")[0].cpu().numpy().astype(int)
    d = np.abs(a - b).max(axis=2)
    print(rows, cols, "max", d.max(), "bad", int((d > 1).sum()))
    if d.max() > 1:
        print("per 32-col tile x per 8-row group: count of |d|>1")
        for r0 in range(0, rows, 8):
            print("%4d " % r0 + " ".join("%3d" % (d[r0:r0 + 8, c0:c0 + 32] > 1).sum() for c0 in range(0, cols, 32)))
        print("b - a, channel 0, rows 0..15 x cols 0..40")
        print((b - a)[0:16, 0:40, 0])
        print("a rows 0..3:", a[0:4, 0:12, 0])
        print("b rows 0..3:", b[0:4, 0:12, 0])
