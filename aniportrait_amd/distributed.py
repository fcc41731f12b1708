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


def window_sum_buffers(S, L, HWC, device):
    """(flat, acc, counter): the per-step window sums of one clip as views of ONE persistent fp32 buffer, so the
    per-step exchange is a single in-place all-reduce of `flat` — no packing copy before it and none after."""
    n = S * L * HWC
    flat = torch.empty((n + L,), dtype=torch.float32, device=device)
    return flat, flat[:n].view(S, L, HWC), flat[n:]


def allreduce_flat(flat, group=None):
    """in-place sum over ranks of the buffer made by `window_sum_buffers` (each rank ran its own windows)"""
    rank, ws = world(group)
    if ws > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def allreduce_window_sums(acc, counter, group=None):
    """Sum the per-frame accumulators of one DDIM step over ranks (each rank ran its own windows), for callers whose
    acc / counter are separate tensors: packed into one flat message (the pipeline itself uses `window_sum_buffers` +
    `allreduce_flat`, which needs no packing)."""
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
    """Broadcast a list of same-dtype tensors (shapes known on every rank) as one flat buffer, in place.  Once per clip
    (the 16 ReferenceNet banks, 46 MB at 512x512): one large message suits the point-to-point xGMI links better than
    16 small ones; the packing copy is paid once per clip, not per step."""
    rank, ws = world(group)
    if ws == 1 or not tensors:
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    if rank != src:
        o = 0
        for t in tensors:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()
    return tensors


def gather_frames(local_frames, local_idx, n_frames, dst=0, group=None, owner_fn=None):
    """local_frames (n_local, ...) holding global frame indices `local_idx`; returns the (n_frames, ...) tensor on
    `dst` (None elsewhere).  Ranks may own different frame counts.  gloo: every rank sends EXACTLY its share
    point-to-point (`batch_isend_irecv` into exact-size staging buffers on `dst`); nccl (= RCCL): one `dist.gather` with
    every share padded to the largest (see the branch).  The frame indices of every rank follow from the sharding rule
    (`owner_fn(rank) -> indices`, default round robin), so they are not communicated."""
    rank, ws = world(group)
    dev = local_frames.device
    if ws == 1:
        out = torch.empty((n_frames,) + tuple(local_frames.shape[1:]), dtype=local_frames.dtype, device=dev)
        out[torch.as_tensor(local_idx, dtype=torch.long, device=dev)] = local_frames
        return out
    owner_fn = owner_fn or (lambda r: shard_round_robin(n_frames, r, ws))
    assert list(local_idx) == list(owner_fn(rank)), "gather_frames: local_idx does not follow the sharding rule"
    g = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    tail = tuple(local_frames.shape[1:])
    if dist.get_backend(group) == "nccl":
        # RCCL: one collective with every share padded to the largest (the exact-size point-to-point form below has only
        # ever run on gloo — no multi-GPU box was available to the builds; it stays the CPU path until it has been measured
        # over xGMI, where per-pair communicator setup and zero-share ranks are the open questions)
        most = max(len(list(owner_fn(r))) for r in range(ws))
        send = torch.zeros((most,) + tail, dtype=local_frames.dtype, device=dev)
        send[: len(local_idx)] = local_frames
        bufs = [torch.empty_like(send) for _ in range(ws)] if rank == dst else None
        dist.gather(send, bufs, dst=g(dst), group=group)
        if rank != dst:
            return None
        out = torch.empty((n_frames,) + tail, dtype=local_frames.dtype, device=dev)
        for r in range(ws):
            idx = list(owner_fn(r))
            if idx:
                out[torch.as_tensor(idx, dtype=torch.long, device=dev)] = bufs[r][: len(idx)]
        return out
    if rank != dst:
        if len(local_idx):
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_frames.contiguous(), g(dst), group)]):
                w.wait()
        return None
    out = torch.empty((n_frames,) + tail, dtype=local_frames.dtype, device=dev)
    ops_, staged = [], []
    for r in range(ws):
        idx = list(owner_fn(r))
        if not idx:
            continue
        if r == dst:
            out[torch.as_tensor(idx, dtype=torch.long, device=dev)] = local_frames
            continue
        buf = torch.empty((len(idx),) + tail, dtype=local_frames.dtype, device=dev)
        staged.append((idx, buf))
        ops_.append(dist.P2POp(dist.irecv, buf, g(r), group))
    if ops_:
        for w in dist.batch_isend_irecv(ops_):
            w.wait()
    for idx, buf in staged:
        out[torch.as_tensor(idx, dtype=torch.long, device=dev)] = buf
    return out
