"""aniportrait_amd.distributed on 2 CPU processes (gloo): the N>1 path of the pipeline — window sharding,
per-step all-reduce of the window sums, bank broadcast, frame gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from aniportrait_amd import distributed as D
        from aniportrait_amd.context import uniform
        L, HWC, S = 40, 24, 2
        windows = [list(w) for w in uniform(0, 25, L, 16, 1, 4)]
        g = torch.Generator().manual_seed(7)
        preds = [torch.randn((S, len(w), HWC), generator=g) for w in windows]   # same on every rank

        def accumulate(idx):
            acc, cnt = torch.zeros(S, L, HWC), torch.zeros(L)
            for k in idx:
                for j, fr in enumerate(windows[k]):
                    acc[:, fr] += preds[k][:, j]
                    cnt[fr] += 1
            return acc, cnt

        full_acc, full_cnt = accumulate(range(len(windows)))
        mine = D.shard_round_robin(len(windows), rank, world)
        acc, cnt = accumulate(mine)
        D.allreduce_window_sums(acc, cnt)
        ok1 = torch.allclose(acc, full_acc, atol=1e-5) and torch.equal(cnt, full_cnt)
        # the pipeline's form: acc / counter as views of one persistent flat buffer, one in-place all-reduce
        flat, acc2, cnt2 = D.window_sum_buffers(S, L, HWC, "cpu")
        for _ in range(2):                       # reused across steps
            flat.zero_()
            a_, c_ = accumulate(mine)
            acc2.copy_(a_); cnt2.copy_(c_)
            D.allreduce_flat(flat)
            ok1 = ok1 and torch.allclose(acc2, full_acc, atol=1e-5) and torch.equal(cnt2, full_cnt)

        banks = [torch.full((2, 5, 3), float(i + 1)).half() if rank == 0 else torch.zeros(2, 5, 3).half()
                 for i in range(4)]
        D.broadcast_tensors(banks, 0)
        ok2 = all(torch.equal(b, torch.full((2, 5, 3), float(i + 1)).half()) for i, b in enumerate(banks))

        frames_idx = D.shard_round_robin(7, rank, world)          # uneven: 4 + 3
        local = torch.stack([torch.full((3, 2), float(i)) for i in frames_idx])
        out = D.gather_frames(local, frames_idx, 7, 0)
        ok3 = (out is None) if rank != 0 else bool((out[:, 0, 0] == torch.arange(7.0)).all())
        q.put((rank, ok1, ok2, ok3, len(mine)))
    finally:
        dist.destroy_process_group()


def test_two_rank_window_parallelism():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res
    assert sum(r[4] for r in res) == 4  # L=40 -> 4 windows


def test_sharding_helpers_single_process():
    from aniportrait_amd import distributed as D
    assert D.world() == (0, 1)
    assert D.shard_round_robin(13, 5, 8) == [5]
    assert D.shard_round_robin(13, 0, 8) == [0, 8]
    parts = D.shard_balanced([16] * 13, 8)
    assert sorted(sum(parts, [])) == list(range(13)) and max(len(p) for p in parts) == 2
    a, c = torch.ones(2, 3, 4), torch.ones(3)
    assert D.allreduce_window_sums(a, c)[0] is a
    out = D.gather_frames(torch.arange(6.0).reshape(3, 2), [2, 0, 1], 3)
    assert out.tolist() == [[2.0, 3.0], [4.0, 5.0], [0.0, 1.0]]


class _Patch:
    """minimal stand-in for pytest's monkeypatch inside a spawned worker"""

    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def _pipeline_worker(rank, world, port, q, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here)]
    torch.set_num_threads(max(2, (os.cpu_count() or 4) // 2))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import emu_hipops
        emu_hipops.install(_Patch())
        from aniportrait_amd import configs as C
        from aniportrait_amd.scheduling_ddim import DDIMScheduler
        from golden_inputs import pipe_inputs
        from src.pipelines.pipeline_pose2vid_long import Pose2VideoPipeline
        from util import build_hip_models, load_golden, psnr, small_clip_encoder
        m, _ = build_hip_models(True, device="cpu")
        # long_L10_ctx8: L = 10, 8-frame windows: two windows per step -> one per rank
        # long_L4:       ONE window: rank 1 owns none (its sums stay zero) and still decodes its share of the frames
        i = pipe_inputs(case)
        pipe = Pose2VideoPipeline(vae=m["vae"], image_encoder=small_clip_encoder("cpu"), reference_unet=m["reference_unet"],
                                  denoising_unet=m["denoising_unet"], pose_guider=m["pose_guider"],
                                  scheduler=DDIMScheduler(**C.DDIM_V2))
        pipe.set_progress_bar_config(disable=True)
        if rank != 0:  # only rank 0's ReferenceNet banks may be used: break the others' ReferenceNet
            for p_ in m["reference_unet"].parameters():
                p_.data.zero_()
            m["reference_unet"]._invalidate()
        out = pipe(i["ref_image"], list(i["poses"]), i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"],
                   latents=i["latents"], dp_group=dist.group.WORLD, **i["kw"])
        if rank == 0:
            gold = load_golden("small_pipeline.pt")
            q.put((rank, float(psnr(out.videos, gold[case + "/video_f16"].float())), tuple(out.videos.shape)))
        else:
            q.put((rank, out is None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,L", [("long_L10_ctx8", 10), ("long_L4", 4)])
def test_two_rank_long_clip_pipeline_matches_reference(case, L):
    """One long clip on 2 ranks (gloo, CPU, kernel wrappers emulated — tests/emu_hipops.py): windows sharded round
    robin, rank 0's ReferenceNet banks broadcast, per-step all-reduce of the window sums, frames decoded per rank
    and gathered — the decoded video on rank 0 matches the reference's single-process pipeline.  The single-window
    case covers a rank without any window (round 1 produced 0/0 = NaN latents there)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert res[0][0] == 0 and res[0][1] >= 40.0 and res[0][2] == (1, 3, L, 128, 128), res
    assert res[1][0] == 1 and res[1][1] is True, res


def _long_clip_inputs():
    """L = 78 at 64x64 with 8-frame windows overlapping by 2: 13 windows per DDIM step (the window count of BASELINE
    configs[3]: L = 150, 16-frame windows), the last one wrapping around the clip end (the same windows at every DDIM step: step 0 is what the reference passes)"""
    from aniportrait_amd.synthetic import synth_latents, synth_pose_frames, synth_ref_image
    H = W = 64
    L = 78
    return dict(H=H, W=W, L=L, steps=2, cfg=3.5, kw=dict(context_frames=8, context_overlap=2),
                poses=synth_pose_frames(L, H, W), ref_pose=synth_pose_frames(1, H, W, 999)[0],
                ref_image=synth_ref_image(H, W), latents=synth_latents(L, H // 8, W // 8, 42))


def _run_long_clip(dp_group=None):
    from aniportrait_amd import configs as C
    from aniportrait_amd.scheduling_ddim import DDIMScheduler
    from src.pipelines.pipeline_pose2vid_long import Pose2VideoPipeline
    from util import build_hip_models, small_clip_encoder
    m, _ = build_hip_models(True, device="cpu")
    i = _long_clip_inputs()
    pipe = Pose2VideoPipeline(vae=m["vae"], image_encoder=small_clip_encoder("cpu"), reference_unet=m["reference_unet"],
                              denoising_unet=m["denoising_unet"], pose_guider=m["pose_guider"],
                              scheduler=DDIMScheduler(**C.DDIM_V2))
    pipe.set_progress_bar_config(disable=True)
    # decode_chunk=1: every frame is decoded alone on whatever rank owns it, so the decoder sees the same shapes in the
    # sharded and in the single-process run (the CPU convolutions of the kernel emulation round differently per batch size)
    kw = dict(i["kw"], decode_chunk=1)
    if dp_group is not None:
        kw["dp_group"] = dp_group
    return pipe(i["ref_image"], list(i["poses"]), i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"],
                latents=i["latents"], **kw)


def _eight_rank_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here)]
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import emu_hipops
        emu_hipops.install(_Patch())
        out = _run_long_clip(dist.group.WORLD)
        # by value (numpy): a torch tensor travels through the queue as a shared-memory handle served by THIS process,
        # which may have exited before the parent reads it
        q.put((rank, None if out is None else out.videos.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.slow
def test_eight_rank_long_clip_equals_single_process_bit_for_bit(monkeypatch):
    """BASELINE configs[3]'s sharding on the rank count it is meant for: ONE long clip, 13 windows per step dealt to 8
    ranks (`shard_balanced`: five ranks run two windows, three run one), rank 0's ReferenceNet banks broadcast, one
    in-place all-reduce of the window sums per step, 78 frames decoded 10 / 10 / ... / 9 per rank and gathered
    point-to-point with uneven shares — 8 gloo processes on the kernel emulator.  Every frame is covered by at most two
    windows, so its window sum has at most two non-zero terms and the all-reduce order cannot change it: the video on
    rank 0 must EQUAL the single-process result bit for bit."""
    import emu_hipops
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    emu_hipops.install(monkeypatch)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)                   # as in the workers: the CPU GEMMs' summation order follows the thread count
    try:
        ref = _run_long_clip(None).videos      # world size 1, in this process, while the workers run
    finally:
        torch.set_num_threads(nthreads)
    res = dict(q.get(timeout=1500) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(res[r] is None for r in range(1, world))
    got = torch.from_numpy(res[0])
    assert tuple(got.shape) == tuple(ref.shape) == (1, 3, 78, 64, 64)
    assert torch.isfinite(got).all() and float(got.std()) > 1e-3
    assert torch.equal(got, ref), f"8-rank video differs from world size 1: max |d| = {float((got - ref).abs().max()):.3e}"
    # the sharding the run used: every window owned exactly once, no rank with more than two
    from aniportrait_amd import distributed as D
    parts = D.shard_balanced([8] * 13, world)
    assert sorted(sum(parts, [])) == list(range(13)) and max(len(p_) for p_ in parts) == 2 and min(len(p_) for p_ in parts) == 1
