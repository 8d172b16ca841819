"""Frame sharding across the GPUs of one node.

Frames of a batch are independent, so the hot path shards with NO data-path collective: rank r
blurs its own contiguous slice.  Collectives (RCCL through torch.distributed's "nccl" backend on
the GPU box, gloo on CPU in the tests) are used only to fan whole frames out from / back in to one
rank when the batch lives there.  The blur itself is passed in as a callable so that the same
plumbing is exercised on CPU in tests/ with the oracle standing in for the GPU.
"""
import torch
import torch.distributed as dist


def frame_shard(nframes, world, rank):
    """[begin, end) of the contiguous slice of frames owned by `rank`; sizes differ by at most one"""
    base, extra = divmod(int(nframes), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(nframes, world):
    return [frame_shard(nframes, world, r)[1] - frame_shard(nframes, world, r)[0] for r in range(world)]


def scatter_frames(frames, nframes, frame_shape, src=0, device=None, group=None):
    """rank `src` holds uint8 [nframes, *frame_shape]; every rank gets its frame_shard() slice.
    Point-to-point sends (xGMI is point to point; a ring would be per-link bound)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b, e = frame_shard(nframes, world, rank)
    if rank == src:
        reqs = []
        for r in range(world):
            rb, re = frame_shard(nframes, world, r)
            if r != src and re > rb:
                reqs.append(dist.isend(frames[rb:re].contiguous(), dst=r, group=group))
        mine = frames[b:e].clone()
        for q in reqs:
            q.wait()
        return mine
    mine = torch.empty((e - b,) + tuple(frame_shape), dtype=torch.uint8, device=device)
    if e > b:
        dist.recv(mine, src=src, group=group)
    return mine


def gather_frames(local, nframes, dst=0, group=None):
    """inverse of scatter_frames: rank `dst` returns uint8 [nframes, ...], the others None"""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == dst:
        out = torch.empty((nframes,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        b, e = frame_shard(nframes, world, rank)
        out[b:e] = local
        for r in range(world):
            rb, re = frame_shard(nframes, world, r)
            if r != dst and re > rb:
                dist.recv(out[rb:re], src=r, group=group)
        return out
    if local.shape[0]:
        dist.send(local.contiguous(), dst=dst, group=group)
    return None


def blur_batch_sharded(blur_fn, frames, nframes, frame_shape, src=0, device=None, group=None):
    """fan out from `src`, blur the local shard with blur_fn(tensor)->tensor, fan back in"""
    mine = scatter_frames(frames, nframes, frame_shape, src, device, group)
    done = blur_fn(mine) if mine.shape[0] else mine
    return gather_frames(done, nframes, src, group)
