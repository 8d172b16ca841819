import sys, os, time, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import blur_algorithms_amd as B
ctx = B.BlurContext(0)
for rows, cols in ((1500, 1000), (2400, 1600), (3300, 2200), (4425, 2950), (6000, 4000), (11400, 7600)):
    sigma = math.sqrt(rows)
    img = torch.randint(0, 256, (rows, cols, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty_like(img)
    res = []
    for reps in (5, 50, 500):
        if rows > 5000 and reps > 50: continue
        for _ in range(2): ctx.pffft_(img, sigma, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ctx.pffft_(img, sigma, out=out)
        e1.record(); torch.cuda.synchronize()
        res.append("%d reps %.3f ms" % (reps, e0.elapsed_time(e1) / reps))
        time.sleep(0.3)
    print(rows, cols, ctx.last_family(), " | ".join(res), flush=True)
