import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blur_algorithms_amd as B
g = torch.Generator(device="cuda").manual_seed(1)
for order in ("fused-first", "matrix-first", "fused-only-twice"):
    ctx = B.BlurContext(0)
    fr = torch.randint(0, 256, (1, 100, 100, 3), dtype=torch.uint8, device="cuda", generator=g)
    res = {}
    seq = {"fused-first": ("fused", "matrix", "fused"), "matrix-first": ("matrix", "fused", "matrix"), "fused-only-twice": ("fused", "fused", "fft")}[order]
    outs = []
    for eng in seq:
        o = ctx.pffft_(fr, 20.0, out=torch.empty_like(fr), nyquist_quirk=False, engine=eng)[0].cpu().numpy().astype(int)
        outs.append((eng, o))
    ref = [o for e, o in outs if e in ("matrix", "fft")][0]
    print(order, [(e, int(np.abs(o - ref).max())) for e, o in outs])
    ctx.close()
