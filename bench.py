"""bench.py — generated frames/s of the pose2vid hot path on MI355X (BASELINE.json metric).

One "step" = one full `Pose2VideoPipeline.__call__` on a synthetic 16-frame clip at BASELINE configs[1]:
512x512, L=16, 25 DDIM steps, CFG 3.5, fp16 (CLIP embed + VAE encode + ReferenceNet + 25 x UNet3D on the
CFG batch of 32 frames + 16 VAE decodes + D2H of the frames).  Weights are random (no checkpoints: no
network), shapes are the real SD-1.5 / sd-vae-ft-mse / CLIP ViT-L/14 ones.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, every rank generates its own clips (frame-batch data parallelism over
independent clips: no data-path collective, "weak" scaling); timing is bracketed by barrier +
synchronize and the MAX over ranks is reported.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, outside the timed region):
  roofline      per-kernel HIP-event timing of one more clip (events on the launch stream, recorded by
                the library around every launch) -> dominant kernel family's achieved TFLOP/s vs the
                2.5 PFLOP/s dense fp16 MFMA peak; the full per-kernel table goes to
                gpurun_out/bench_kernels_table.json.
  cpu_baseline  (N == 1 only) the CPU oracle (oracle/ref_torch.py, a restatement of the reference's
                PyTorch path) timed on the host cores on a bounded sample of the same workload and
                extrapolated by algorithmic FLOPs.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# algorithmic work model (SURVEY.md §8d / BASELINE.md §2), TFLOP
TF_UNET = {512: 36.43, 256: 2.00}       # one UNet3D call (CFG batch 2 x L frames; 256: L=4)
TF_REFNET = {512: 1.59, 256: 0.35}
TF_VAE_FRAME = {512: 2.515, 256: 0.622}
TF_PER_FRAME_C2 = 59.54                  # (25 * 36.43 + 1.59 + 16 * 2.515) / 16
MFMA_PEAK_TFLOPS = 2500.0                # dense fp16, MI355X_MICROARCH.md


def build_pipeline(device, H=512, W=512, small=False, seed=0):
    from aniportrait_amd import configs as C
    from aniportrait_amd.autoencoder_kl import AutoencoderKL
    from aniportrait_amd.params import skip_init
    from aniportrait_amd.pipeline_pose2vid_long import Pose2VideoPipeline
    from aniportrait_amd.pose_guider import PoseGuider
    from aniportrait_amd.scheduling_ddim import DDIMScheduler
    from aniportrait_amd.synthetic import fast_fill_
    from aniportrait_amd.unet import UNet2DConditionModel, UNet3DConditionModel
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    dt = torch.float16
    with skip_init():
        den = UNet3DConditionModel(**C.unet3d_kwargs(small))
        ref = UNet2DConditionModel(**C.unet2d_kwargs(small))
        vae = AutoencoderKL(**(C.SD_VAE_SMALL if small else C.SD_VAE_FT_MSE))
        ch0 = (C.SD15_UNET_SMALL if small else C.SD15_UNET)["block_out_channels"][0]
        pg = PoseGuider(noise_latent_channels=ch0, use_ca=True)
    enc = CLIPVisionModelWithProjection(CLIPVisionConfig(**(C.CLIP_SMALL if small else C.CLIP_VIT_L14))).eval()
    mods = dict(denoising_unet=den, reference_unet=ref, vae=vae, pose_guider=pg, image_encoder=enc)
    for i, (k, m) in enumerate(mods.items()):
        m.to(device=device, dtype=dt)
        fast_fill_(m, seed * 16 + i)
    pipe = Pose2VideoPipeline(vae=vae, image_encoder=enc, reference_unet=ref, denoising_unet=den, pose_guider=pg,
                              scheduler=DDIMScheduler(**C.DDIM_V2))
    pipe.set_progress_bar_config(disable=True)
    return pipe


def clip_inputs(H, W, L, seed):
    from aniportrait_amd.synthetic import synth_latents, synth_pose_frames, synth_ref_image
    return dict(ref_image=synth_ref_image(H, W, seed + 1), poses=list(synth_pose_frames(L, H, W, 1234 + 100 * seed)),
                ref_pose=synth_pose_frames(1, H, W, 999)[0], latents=synth_latents(L, H // 8, W // 8, 42 + seed))


def run_clip(pipe, inp, H, W, L, steps, cfg):
    return pipe(inp["ref_image"], inp["poses"], inp["ref_pose"], W, H, L, steps, cfg, generator=None,
                latents=inp["latents"]).videos


def cpu_baseline(budget_s=20.0):
    """Oracle (oracle/ref_torch.py, fp32) on the host cores, on a bounded sample of the same workload:
    real-width UNet3D calls with reference banks (CFG batch) and one VAE frame decode at reduced spatial
    size / clip length, escalated while the time budget lasts; the largest sample measured is extrapolated
    to the 512x512 / L=16 / 25-step clip by algorithmic FLOPs (conv / linear FLOPs scale with pixels x
    frames; the attention share grows faster, so this favours the CPU)."""
    from aniportrait_amd import configs as C
    from aniportrait_amd.params import unet_shapes, vae_shapes
    from aniportrait_amd.pipeline_pose2vid_long import bank_shapes
    from oracle import ref_torch as O

    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count()
    threads = max(1, min(32, ncpu))
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)

    def rand_sd(shapes):
        sd = {}
        for k, s in shapes.items():
            t = torch.randn(tuple(s), generator=g)
            if len(s) >= 2:
                t *= 1.0 / math.sqrt(float(torch.tensor(s[1:]).prod()))
            elif k.endswith("weight"):
                t = 1 + 0.1 * t
            else:
                t *= 0.1
            sd[k] = t
        return sd

    ucfg = C.unet3d_kwargs(False)
    sd_u = rand_sd(unet_shapes(ucfg, True)[0])
    sd_v = rand_sd(vae_shapes(C.SD_VAE_FT_MSE)[0])
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn((1, 1, 768), generator=g)])
    spent, best = 0.0, None
    # (latent side, frames): UNet3D TFLOP ~ 36.43 * (h/64)^2 * (f/16) (lower bound: attention grows faster)
    for h, f in ((8, 2), (16, 2), (16, 4), (32, 4)):
        tf = TF_UNET[512] * (h / 64.0) ** 2 * (f / 16.0)
        if best is not None and spent + best[1] * tf / best[0] > budget_s:
            break
        banks = {p: torch.randn(s, generator=g).half().float() for p, s in bank_shapes(ucfg, 2, h, h).items()}
        lat = torch.randn((1, 4, f, h, h), generator=g).repeat(2, 1, 1, 1, 1)
        with torch.no_grad():
            t0 = time.time()
            O.unet3d_forward(sd_u, ucfg, lat, 959, ehs, None, banks, True)
            dt = time.time() - t0
        spent += dt
        best = (tf, dt, h, f)
    tf_u, t_u, h_u, f_u = best
    hv = 8 if t_u * (TF_VAE_FRAME[512] / 64.0) / tf_u > 10 else 16
    tf_v = TF_VAE_FRAME[512] * (hv / 64.0) ** 2
    with torch.no_grad():
        t0 = time.time()
        O.vae_decode(sd_v, C.SD_VAE_FT_MSE, torch.randn((1, 4, hv, hv), generator=g))
        t_v = time.time() - t0
    rate_u, rate_v = tf_u / t_u, tf_v / t_v
    t_clip = (25 * TF_UNET[512] + TF_REFNET[512]) / rate_u + 16 * TF_VAE_FRAME[512] / rate_v
    return dict(value=16.0 / t_clip, unit="frames/s", cores=threads, kind="port",
                sample=f"oracle/ref_torch.py fp32, {threads} torch threads ({ncpu} schedulable cores): real-width UNet3D "
                       f"call with reference banks at {8 * h_u}x{8 * h_u} px, L={f_u}, CFG batch ({t_u:.2f} s = "
                       f"{rate_u:.3f} TFLOP/s) + 1 VAE frame decode at {8 * hv}x{8 * hv} px ({t_v:.2f} s = "
                       f"{rate_v:.3f} TFLOP/s); extrapolated by algorithmic FLOPs to the 512x512 L=16 25-step clip "
                       f"({t_clip:.0f} s per clip)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed clips per rank")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--ddim-steps", type=int, default=25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    n_gpus = world
    assert a.gpus == n_gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if world > 1:  # the per-clip host work (PIL resize, CLIP preprocessing) is tiny: do not oversubscribe the host cores
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // (2 * world)))
    device = torch.device("cuda", local_rank if world > 1 else 0)

    H = W = a.size
    L = a.frames
    pipe = build_pipeline(device, H, W, seed=rank)
    inputs = [clip_inputs(H, W, L, seed=rank * 100 + i) for i in range(2)]

    for i in range(a.warmup):
        run_clip(pipe, inputs[i % 2], H, W, L, a.ddim_steps, 3.5)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        vid = run_clip(pipe, inputs[i % 2], H, W, L, a.ddim_steps, 3.5)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert tuple(vid.shape) == (1, 3, L, H, W) and bool(torch.isfinite(vid).all())

    if rank == 0:
        frames = n_gpus * a.steps * L
        fps = frames / elapsed
        out = {
            "metric": "generated frames/sec, 512x512 L=16 25-step pose2vid",
            "value": fps, "unit": "frames/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"pose2vid {H}x{W}, L={L}, {a.ddim_steps} DDIM steps, CFG=3.5, fp16, one 16-frame clip "
                                   "per step per GPU (BASELINE.json configs[1])",
                       "frames_per_step": L, "parallelism": f"dp{n_gpus} over independent clips"},
        }
        is_c2 = (H == 512 and L == 16 and a.ddim_steps == 25)
        if not a.no_roofline:
            from aniportrait_amd import hipops
            with hipops.profile() as prof:
                run_clip(pipe, inputs[0], H, W, L, a.ddim_steps, 3.5)
            table = prof.result
            mf = {k: v for k, v in table.items() if v["unit"] == "TFLOP/s"}
            dom = max(mf, key=lambda k: mf[k]["ms"])
            d = mf[dom]
            total_ms = sum(v["ms"] for v in table.values())
            out["roofline"] = {
                "bound": "mfma", "kernel": dom, "achieved": d["rate"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": d["rate"] / MFMA_PEAK_TFLOPS, "traffic": None,
                "launches": d["launches"], "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                "share_of_gpu_kernel_time": d["ms"] / total_ms,
                "whole_clip_frac": (fps / n_gpus * TF_PER_FRAME_C2 / MFMA_PEAK_TFLOPS) if is_c2 else None,
            }
            os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
            with open(os.path.join(REPO, "gpurun_out", "bench_kernels_table.json"), "w") as f:
                json.dump({"clip_kernel_ms": total_ms, "kernels": table, "by_shape": prof.by_shape}, f, indent=1)
        if n_gpus == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
