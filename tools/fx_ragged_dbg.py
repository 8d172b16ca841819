"""tools/fx_ragged_dbg.py -- the fused kernel on widths that are no multiples of 4 against the two-kernel engine: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blur_algorithms_amd as B
ctx = B.BlurContext(0)
g = torch.Generator(device="cuda").manual_seed(1)
for rows, cols, sigma in ((210, 333, 20.0), (210, 332, 20.0), (131, 154, 18.0), (70, 69, 20.0), (100, 257, 12.0)):
    for quirk in (False, True):
        fr = torch.randint(0, 256, (1, rows, cols, 3), dtype=torch.uint8, device="cuda", generator=g)
        a = ctx.pffft_(fr, sigma, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="matrix")[0].cpu().numpy().astype(int)
        b = ctx.pffft_(fr, sigma, out=torch.empty_like(fr), nyquist_quirk=quirk, engine="fused")[0].cpu().numpy().astype(int)
        d = np.abs(a - b)
        bad = np.argwhere(d > 1)
        print("%d x %d sigma %g quirk %d: max diff %d, >1 at %d places, fam %d" % (rows, cols, sigma, quirk, d.max(), len(bad), ctx.last_family()))
        if len(bad):
            print("   rows", np.unique(bad[:, 0])[:12], "cols", np.unique(bad[:, 1])[:24], "ch", np.unique(bad[:, 2]))
            r, x, c = bad[0]
            print("   first: (%d, %d, %d): matrix %d fused %d; d by col at that row:" % (r, x, c, a[r, x, c], b[r, x, c]), d[r, :, c].tolist()[:40])
