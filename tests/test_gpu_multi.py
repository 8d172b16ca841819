"""The N > 1 path with real kernels on the one GPU a test box has: bench.py's own launch of two ranks (both on cuda:0, gloo for the
barrier and the max-reduce; the data path has no collective) and blur_multi_* with the frames of BASELINE config 4's per-GPU shard
(8 x 4K per shard) on two logical shards.  The driver measures the real 1 / 2 / 4 / 8 GPU scaling curve; these tests keep that
path correct (Utils.hpp:16-55 is the reference's whole parallel runtime)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_real_kernels_on_one_gpu():
    """`python bench.py --gpus 2 --all-on-device0 --dist-backend gloo`: a fresh child process starts two ranks itself; each blurs its
    own 8 frames per step with the real kernels; rank 0 prints ONE JSON line with n_gpus = 2 and the whole job's rate"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--all-on-device0", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--no-cpu", "--no-natural", "--no-copy", "--settle", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["config"]["frames_per_gpu"] == 8
    assert "fused" in rec["config"]["engine"]


def test_multi_handle_on_the_c4_shard_shape(ctx):
    """blur_multi_* (one host thread, one context and stream per shard) with host frames of the C4 shape -- two logical shards of
    8 x 3840 x 2160 frames on device 0 -- against the bytes of the single-context call"""
    import torch
    import blur_algorithms_amd as B
    rng = np.random.default_rng(64)
    frames = rng.integers(0, 256, (16, 2160, 3840, 3), dtype=np.uint8)
    m = B.BlurMulti([0, 0])
    got = m.pffft_(frames, 20.0)
    m.close()
    for lo in (0, 8):
        t = torch.from_numpy(frames[lo:lo + 8]).cuda()
        want = ctx.pffft_(t, 20.0, out=torch.empty_like(t)).cpu().numpy()
        assert np.array_equal(got[lo:lo + 8], want)
        del t


def test_two_contexts_two_host_threads_fused_kernel_concurrently():
    """two blur_ctx on device 0, each with its own stream, driven from two host threads at the same time (ctypes releases the GIL
    during a call): the closest a one-GPU box gets to blur_multi_* on distinct devices.  Every context sets the kernels' LDS
    attribute for its device itself (fx_attr_needed: per device, not per process) and the two launches share the chip; each thread's
    bytes equal the single-context result, over several rounds and three kernel widths"""
    import threading
    import torch
    import blur_algorithms_amd as B
    ref = B.BlurContext(0)
    work = []
    for i, (rows, cols, sigma) in enumerate(((540, 964, 20.0), (360, 641, 12.0), (700, 1283, 30.0))):
        img = torch.from_numpy(np.random.default_rng(100 + i).integers(0, 256, (2, rows, cols, 3), dtype=np.uint8)).cuda()
        want = ref.pffft_(img, sigma, out=torch.empty_like(img), engine="fused").cpu().numpy()
        work.append((img, sigma, want))
    torch.cuda.synchronize()
    errors = []

    def run(tid):
        try:
            ctx = B.BlurContext(0)
            st = torch.cuda.Stream(device=0)
            ctx.set_stream(st.cuda_stream)
            for rnd in range(6):
                img, sigma, want = work[(rnd + tid) % len(work)]
                out = torch.empty_like(img)
                o = ctx._opts(True, 0, False, 0, False, None, "fused")
                import ctypes as C
                ctx._check(ctx._lib.blur_gaussian_u8c3_batch_dev(ctx._h, img.data_ptr(), out.data_ptr(), img.shape[0], img.shape[1], img.shape[2], float(sigma), C.byref(o)))
                ctx.synchronize()
                if not np.array_equal(out.cpu().numpy(), want):
                    errors.append("thread %d round %d: bytes differ" % (tid, rnd))
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (tid, e))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    ref.close()
    assert not errors, errors
