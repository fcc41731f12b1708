"""Frame-batch data parallelism for pose2vid on one node: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-process (`scripts/pose2vid.py:59-110`).  What shards naturally:
  * independent clips                      -> no communication at all (the `bench.py --gpus N` mode);
  * the context windows of ONE long clip   -> windows of a DDIM step are independent UNet calls
    (src/pipelines/pipeline_pose2vid_long.py:519-548); the per-frame sums `noise_pred`/`counter` need one
    all-reduce(sum) per step, only when there is more than one window;
  * per-frame VAE decode (:119-120)        -> frames are independent; gathered to rank 0.
The ReferenceNet bank (16 tensors, 46 MB fp16 at 512x512) is computed on rank 0 and broadcast once per
clip as ONE flat buffer (xGMI is point-to-point: one large message instead of 16 small ones).
CFG pairs stay on one GPU (same batch), so no CFG all-reduce exists.

All functions are no-ops for world size 1 / group None and work on CPU tensors (gloo) for the tests.
"""
import torch
import torch.distributed as dist


def world(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_round_robin(n_items, rank, world_size):
    """indices of the items owned by `rank` (window k -> rank k mod G)"""
    return list(range(rank, n_items, world_size))


def shard_balanced(costs, world_size):
    """Greedy longest-processing-time assignment of items with `costs` to ranks; returns a list of index
    lists.  Used for windows of unequal frame count (the last window of a non-closed loop)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    return [sorted(x) for x in out]


def allreduce_window_sums(acc, counter, group=None):
    """Sum the per-frame accumulators of one DDIM step over ranks (each rank ran its own windows).
    One flat message: acc and counter are packed together."""
    rank, ws = world(group)
    if ws == 1:
        return acc, counter
    flat = torch.cat([acc.reshape(-1), counter.reshape(-1).to(acc.dtype)])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    n = acc.numel()
    acc.copy_(flat[:n].view_as(acc))
    counter.copy_(flat[n:].view_as(counter).to(counter.dtype))
    return acc, counter


def broadcast_tensors(tensors, src=0, group=None):
    """Broadcast a list of same-dtype tensors (shapes known on every rank) as one flat buffer, in place."""
    rank, ws = world(group)
    if ws == 1 or not tensors:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    o = 0
    for t in tensors:
        t.copy_(flat[o:o + t.numel()].view_as(t))
        o += t.numel()
    return tensors


def gather_frames(local_frames, local_idx, n_frames, dst=0, group=None):
    """local_frames (n_local, ...) holding global frame indices `local_idx`; returns the (n_frames, ...)
    tensor on `dst` (None elsewhere).  Ranks may own different frame counts."""
    rank, ws = world(group)
    if ws == 1:
        out = torch.empty((n_frames,) + tuple(local_frames.shape[1:]), dtype=local_frames.dtype,
                          device=local_frames.device)
        out[torch.as_tensor(local_idx, dtype=torch.long, device=local_frames.device)] = local_frames
        return out
    per = -(-n_frames // ws)
    pad = torch.zeros((per,) + tuple(local_frames.shape[1:]), dtype=local_frames.dtype, device=local_frames.device)
    pad[: local_frames.shape[0]] = local_frames
    idx = torch.full((per,), -1, dtype=torch.long, device=local_frames.device)
    idx[: len(local_idx)] = torch.as_tensor(local_idx, dtype=torch.long, device=local_frames.device)
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    ibufs = [torch.empty_like(idx) for _ in range(ws)]
    dist.all_gather(bufs, pad, group=group)
    dist.all_gather(ibufs, idx, group=group)
    if rank != dst:
        return None
    out = torch.empty((n_frames,) + tuple(local_frames.shape[1:]), dtype=local_frames.dtype,
                      device=local_frames.device)
    for b, i in zip(bufs, ibufs):
        keep = i >= 0
        out[i[keep]] = b[keep]
    return out
