"""Sliding-window ("context") scheduler of the long-clip pipeline — same call surface and results as
src/pipelines/context.py:7-49 (`uniform`, `ordered_halving`, `get_context_scheduler`), restated.

A clip longer than `context_size` frames is denoised as overlapping windows that wrap around the end
of the clip (closed loop); per-window predictions are averaged per frame before CFG
(src/pipelines/pipeline_pose2vid_long.py:546-552).  Windows inside one DDIM step are independent UNet
calls: they are the data-parallel unit `aniportrait_amd.distributed` shards across GPUs.
"""
import math


def ordered_halving(val):
    """bit-reversal of a 64-bit integer mapped to [0, 1): van-der-Corput offset for window phase"""
    rev = 0
    v = int(val)
    for _ in range(64):
        rev = (rev << 1) | (v & 1)
        v >>= 1
    return rev / float(1 << 64)


def uniform(step=0, num_steps=None, num_frames=0, context_size=None, context_stride=3, context_overlap=4,
            closed_loop=True):
    """Yield lists of frame indices.  One window `[0..L-1]` if the clip fits; otherwise, for each
    dilation 2^k (k < context_stride, capped by log2(L/size)+1), windows of `context_size` frames taken
    every 2^k-th frame, advancing by `size*2^k - overlap`, indices modulo L."""
    L = int(num_frames)
    if L <= context_size:
        yield list(range(L))
        return
    n_dil = min(int(context_stride), int(math.ceil(math.log2(L / context_size))) + 1)
    frac = ordered_halving(step)
    shift = int(round(L * frac))
    for k in range(n_dil):
        dil = 1 << k
        start = int(frac * dil) + shift
        stop = L + shift - (0 if closed_loop else context_overlap)
        hop = context_size * dil - context_overlap
        span = context_size * dil
        j = start
        while j < stop:
            yield [e % L for e in range(j, j + span, dil)]
            j += hop


def get_context_scheduler(name):
    if name == "uniform":
        return uniform
    raise ValueError(f"Unknown context_overlap policy {name}")
