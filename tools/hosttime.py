import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch, blur_algorithms_amd as B
ctx = B.BlurContext(0)
img = np.random.default_rng(0).integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
for _ in range(3): ctx.pffft_(img, 20.0)
t0 = time.perf_counter(); n = 20
for _ in range(n): ctx.pffft_(img, 20.0)
dt = (time.perf_counter() - t0) / n
print("host-pointer 4K frame (pageable H2D + blur + D2H, incl. malloc/free): %.3f ms -> %.0f MP/s" % (dt * 1e3, 2160 * 3840 / 1e6 / dt))
pin = torch.from_numpy(img).pin_memory(); out = torch.empty_like(pin).pin_memory(); d = torch.empty((2160, 3840, 3), dtype=torch.uint8, device='cuda')
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    d.copy_(pin, non_blocking=True); ctx.pffft_(d, 20.0); out.copy_(d, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("pinned H2D + blur + D2H on one stream: %.3f ms -> %.0f MP/s" % (dt * 1e3, 2160 * 3840 / 1e6 / dt))

# pipelined host batch (blur_gaussian_u8c3_host_batch): copies in / kernels / copies out overlapped over three slots
nb = 24
for name, alloc in (("pageable", lambda shp: np.empty(shp, np.uint8)), ("pinned", ctx.pinned_empty)):
    a = alloc((nb, 2160, 3840, 3)); a[...] = img[None]; o = alloc(a.shape)
    ctx.pffft_host_batch(a, 20.0, out=o)
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): ctx.pffft_host_batch(a, 20.0, out=o)
    dt = (time.perf_counter() - t0) / (reps * nb)
    print("host batch pipeline, %s memory, %d frames: %.3f ms/frame -> %.0f MP/s (%.1f GB/s each way)" % (
        name, nb, dt * 1e3, 2160 * 3840 / 1e6 / dt, 2160 * 3840 * 3 / dt / 1e9))
    assert np.array_equal(o[5], ctx.pffft_(img, 20.0))
