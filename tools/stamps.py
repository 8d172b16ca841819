#!/usr/bin/env python3
"""read the -DFK_STAMPS phase stamps of the column kernel (diagnostic build only; the unit of s_memtime is NOT the shader clock)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blur_algorithms_amd as B
from blur_algorithms_amd import _lib
rows, cols, sigma = 2160, 3840, 20.0
frames = torch.randint(0, 256, (2, rows, cols, 3), dtype=torch.uint8, device="cuda")
ctx = B.BlurContext(0)
for _ in range(3):
    ctx.pffft_(frames, sigma, out=torch.empty_like(frames), frames_per_launch=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
torch.cuda.synchronize()
lib = _lib.load()
lib.blur_debug_read_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
nblk = 4096
buf = np.zeros(nblk * 8, np.uint64)
for n_fft, names in ((2304, ["prologue/loop/writeout", "gather+barriers", "pass0", "last writeout", "fwd inner", "mid", "inv inner", "inv pass0+stage"]),):
    buf = np.zeros(nblk * 8, np.uint64)
    rc = lib.blur_debug_read_stamps(ctx._h, n_fft, 1 if n_fft == 4000 else 2, buf.ctypes.data, buf.size)
    st = buf.reshape(nblk, 8).astype(np.float64)
    st = st[st.sum(1) > 0]
    tot = st.sum(1)
    print("N=%d: blocks with stamps: %d; cycles per block: mean %.0f  min %.0f  max %.0f" % (n_fft, len(st), tot.mean(), tot.min(), tot.max()))
    for i, n in enumerate(names):
        print("  %-24s mean %9.0f cyc  (%.1f%%)" % (n, st[:, i].mean(), 100 * st[:, i].mean() / tot.mean()))
