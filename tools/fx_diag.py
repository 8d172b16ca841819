import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blur_algorithms_amd as B
ctx = B.BlurContext(0)
g = torch.Generator(device="cuda").manual_seed(1)
for rows, cols in ((256, 384), (70, 68)):
    fr = torch.randint(0, 256, (1, rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g)
    a = ctx.pffft_(fr, 20.0, out=torch.empty_like(fr), nyquist_quirk=False, engine="matrix")[0].cpu().numpy().astype(int)
    b = ctx.pffft_(fr, 20.0, out=torch.empty_like(fr), nyquist_quirk=False, engine="fused")[0].cpu().numpy().astype(int)
    d = np.abs(a - b).max(axis=2)
    print(rows, cols, "max", d.max())
    print("per 32-col tile (cols) x per 8-row group: count of |d|>1")
    for r0 in range(0, rows, 8):
        print("%4d " % r0 + " ".join("%3d" % (d[r0:r0 + 8, c0:c0 + 32] > 1).sum() for c0 in range(0, cols, 32)))
    print("max diff per column (first 160):", d.max(axis=0)[:160].tolist())
    np.set_printoptions(linewidth=250)
    dd = (b - a)[:, :, 0]
    print("b - a, channel 0, rows 0..11 x cols 60..100")
    print(dd[0:12, 60:100])
    print("rows 28..44 x cols 60..100")
    print(dd[28:44, 60:100])
