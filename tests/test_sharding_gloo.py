"""The N>1 path on CPU: world_size-2 gloo processes shard a batch of frames, blur their slice
(the oracle's float32 port stands in for the GPU kernels, which need a device) and fan the
result back in; must equal the unsharded result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blur_algorithms_amd.distributed import blur_batch_sharded, frame_shard, shard_sizes


def test_frame_shard_partitions_exactly():
    for n in range(0, 70):
        for w in (1, 2, 3, 4, 8):
            spans = [frame_shard(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            s = shard_sizes(n, w)
            assert sum(s) == n and max(s) - min(s) <= 1
    assert shard_sizes(64, 8) == [8] * 8                       # BASELINE C4: 64 frames over 8 GPUs


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nframes, shape, sigma, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    frames = None
    if rank == 0:
        frames = torch.from_numpy(np.random.default_rng(8).integers(0, 256, (nframes,) + shape, dtype=np.uint8))

    def blur(t):
        return torch.from_numpy(np.stack([O.pffft_blur_u8c3_f32(f.numpy(), sigma) for f in t]))

    out = blur_batch_sharded(blur, frames, nframes, shape, src=0)
    if rank == 0:
        want = torch.from_numpy(np.stack([O.pffft_blur_u8c3_f32(f.numpy(), sigma) for f in frames]))
        q.put(bool(torch.equal(out, want)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nframes", [5, 2, 1])
def test_two_rank_gloo_fanout_blur_fanin(nframes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nframes, (40, 56, 3), 3.0, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun must start two ranks itself (torch.distributed.run as a child) and
    report n_gpus = 2; --rehearse replaces the GPU step by a sleep so that the launch / barrier / max-reduce / report
    path runs on this CPU-only box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 prints ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    # a world size that does not match --gpus is an error, not a silent 1-GPU number
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse"], capture_output=True, text=True, timeout=120, env=env2)
    assert q.returncode != 0
