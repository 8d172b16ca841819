"""tools/fx_prof.py [engine] [quirk] [sigma] -- a fixed number of 8-frame 4K blurs on one engine, for rocprofv3 runs (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import blur_algorithms_amd as B

eng = sys.argv[1] if len(sys.argv) > 1 else "fused"
quirk = len(sys.argv) > 2 and sys.argv[2] == "1"
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
ctx = B.BlurContext(0)
g = torch.Generator(device="cuda").manual_seed(1)
frames = torch.randint(0, 256, (8, 2160, 3840, 3), dtype=torch.uint8, device="cuda", generator=g)
out = torch.empty_like(frames)
for _ in range(12):
    ctx.pffft_(frames, sigma, out=out, nyquist_quirk=quirk, engine=eng)
torch.cuda.synchronize()
