import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch, blur_algorithms_amd as B
from oracle import oracle as O
ctx = B.BlurContext(0)
rng = np.random.default_rng(3)
for rows, cols, sigma in [(2160, 3840, 20.0), (1080, 1920, 20.0), (2160, 3840, 50.0), (500, 748, 20.0), (512, 512, 5.0)]:
    img = rng.integers(0, 256, (rows, cols, 3), dtype=np.uint8)
    rp = ctx.rowpass(torch.from_numpy(img).cuda(), sigma).cpu().numpy()
    _, inter = O.pffft_plane_f64(img[:, :, 1].astype(np.float32), sigma, True, want_inter=True)
    e_row = np.abs(rp[1].astype(np.float64) - inter).max()
    plane = img[:, :, 2].astype(np.float32)
    want = O.pffft_plane_f64(plane, sigma, True)
    got = ctx.pffft_plane(torch.from_numpy(plane).cuda(), sigma).cpu().numpy()
    e_fin = np.abs(got.astype(np.float64) - want).max()
    print("%dx%d s=%g: max |row-pass err| %.3e   max |final float err| %.3e (generic f32 path)" % (cols, rows, sigma, e_row, e_fin))
