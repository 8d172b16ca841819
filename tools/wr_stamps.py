#!/usr/bin/env python3
"""read the -DWR_STAMPS phase stamps of the wave-resident kernels (diagnostic build only):
   BLUR_AMD_LIB=blur_algorithms_amd/variants/libblur_amd_stamps.so python tools/wr_stamps.py [frames]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blur_algorithms_amd as B
from blur_algorithms_amd import _lib
rows, cols, sigma = 2160, 3840, 20.0
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = torch.randint(0, 256, (nfr, rows, cols, 3), dtype=torch.uint8, device="cuda")
ctx = B.BlurContext(0)
out = torch.empty_like(frames)
for _ in range(3):
    ctx.pffft_(frames, sigma, out=out)
torch.cuda.synchronize()
lib = _lib.load()
lib.blur_debug_read_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
NAMES = {
    2304: ("column kernel (9 waves)", 9, ["pass 0", "barrier 1", "middle", "barrier 2 (+claim)", "inverse pass 0 + stores", "barrier 3", "commit/request next strip", "prologue"]),
    4096: ("row kernel (12 waves)", 12, ["pass 0", "barrier 1", "middle", "barrier 2", "inverse pass 0", "claim + barrier 3", "write-out", "prologue"]),
}
for n, (title, nw, names) in NAMES.items():
    buf = np.zeros(2048 * 8, np.uint64)
    rc = lib.blur_debug_read_stamps(ctx._h, n, -1, buf.ctypes.data, buf.size)
    st = buf.reshape(2048, 8).astype(np.float64)
    st = st[st.sum(1) > 0]
    tot = st.sum(1)
    print("%s: rc %d, waves with stamps %d; cycles per wave: mean %.0f min %.0f max %.0f" % (title, rc, len(st), tot.mean(), tot.min(), tot.max()))
    for i, nm in enumerate(names):
        col = st[:, i]
        print("  %-28s mean %9.0f (%5.1f%%)   min %9.0f  max %9.0f" % (nm, col.mean(), 100 * col.mean() / tot.mean(), col.min(), col.max()))
    # per wave position inside the workgroup (SIMD placement effects)
    w = st[: (len(st) // nw) * nw].reshape(-1, nw, 8)
    print("  middle by wave index:", " ".join("%.0f" % x for x in w[:, :, 2].mean(0)))
    print("  pass 0 by wave index:", " ".join("%.0f" % x for x in w[:, :, 0].mean(0)))
