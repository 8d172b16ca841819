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
