"""tools/bx_sizes.py -- fastboxblur k = 41, P = 3 on RGB images of several sizes (GPU box): ms per call and GB/s on the 12 B/px moved
(does the intermediate of a small image stay in cache between the horizontal and the vertical pass?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blur_algorithms_amd as B
ctx = B.BlurContext(0)
for h, w in ((1080, 1920), (2160, 3840), (3240, 5760), (4320, 7680), (6480, 11520)):
    img = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda")
    for _ in range(5): ctx.fastboxblur(img, 41, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 30
    for _ in range(n): ctx.fastboxblur(img, 41, 3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%5d x %5d: %.4f ms  %.0f MP/s  %.0f GB/s of 12 B/px" % (w, h, dt * 1e3, h * w / 1e6 / dt, 12 * h * w / dt / 1e9), flush=True)
