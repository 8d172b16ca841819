"""tools/mx_diff.py -- where does an engine differ from the oracle?  (GPU box)  usage: python tools/mx_diff.py ENGINE rows cols sigma quirk"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blur_algorithms_amd as B
from oracle import oracle as O

eng, rows, cols, sigma, quirk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), bool(int(sys.argv[5]))
img = np.random.default_rng(rows + 3 * cols).integers(0, 256, (rows, cols, 3), dtype=np.uint8)
want, planes = O.pffft_blur_u8c3_f64(img, sigma, quirk=quirk, want_planes=True)
ctx = B.BlurContext(0)
got = ctx.pffft_(torch.from_numpy(img).cuda(), sigma, nyquist_quirk=quirk, engine=eng).cpu().numpy()
d = got.astype(int) - want.astype(int)
print("mismatches", int((d != 0).sum()), "of", d.size, "max", int(np.abs(d).max()))
bad = np.argwhere(d != 0)
if len(bad):
    print("rows with mismatches (count per 32-row band):", np.bincount(bad[:, 0] // 32, minlength=(rows + 31) // 32))
    print("cols with mismatches (count per 32-col band):", np.bincount(bad[:, 1] // 32, minlength=(cols + 31) // 32))
    print("channels:", np.bincount(bad[:, 2], minlength=3))
    print("row mod 32:", np.bincount(bad[:, 0] % 32, minlength=32))
    print("col mod 32:", np.bincount(bad[:, 1] % 32, minlength=32))
