"""Generate tests/golden/real_pipeline_<case>.pt by running the REFERENCE's own Pose2VideoPipeline
(/root/reference/src/pipelines/pipeline_pose2vid_long.py:339-584, unmodified, via oracle/ref_harness.py) at the REAL
SD-1.5 / sd-vae-ft-mse widths on PyTorch-CPU fp32 — the north star's parity criterion ("outputs match the reference
PyTorch-CPU pipeline ... PSNR >= 40 dB on decoded frames") at BASELINE.json's own geometries.

TEST INFRASTRUCTURE.  Build container only (needs /root/reference; hours of CPU on 8 cores for the 512 / 768 cases):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_real_pipeline.py c2_4step c5_1step l40_windows

Cases and seeded inputs: tests/golden_inputs.py REAL_PIPE_CASES / real_pipe_inputs.  Per case the fixture holds
  latents_f16      (steps, 1, 4, L, h, w)  latents after every DDIM step (the pipeline's `callback`), fp16
  frames_u8        (n, H, W, 3)            decoded frames `frames` of the returned video as display bytes:
                                           round(255 x) (quantisation floor 58.9 dB, far above the 40 dB bar)
  frames           indices of those frames
  video_mean       mean of the full fp32 video (sanity)
  clip_embeds      (1, 768) CLIP image embedding the pipeline computed (tiny name-hash CLIP tower, as in tests/util.py)
Weights: aniportrait_amd.synthetic name-hash values, seed 0 (bit-identical on every machine).
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

from oracle import ref_harness as R  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


@torch.no_grad()
def run_case(models, name):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from golden_inputs import real_pipe_inputs
    from src.pipelines.pipeline_pose2vid_long import Pose2VideoPipeline as LongPipe
    R.assert_reference("src.pipelines.pipeline_pose2vid_long", "src.models.unet_3d", "src.models.mutual_self_attention")
    i = real_pipe_inputs(name)
    pipe = LongPipe(vae=models["vae"], image_encoder=models["image_encoder"], reference_unet=models["reference_unet"],
                    denoising_unet=models["denoising_unet"], pose_guider=models["pose_guider"],
                    scheduler=models["scheduler"])
    lat_steps = []
    t0 = time.time()

    def cb(step, t, lat):
        lat_steps.append(lat.detach().clone().half())
        print(f"  {name}: step {step} (t={int(t)}) done at {time.time() - t0:.0f} s", flush=True)
        # hours-long cases: keep what exists (an interrupted run is finished by `--finish <name>`, below)
        torch.save(dict(latents_f16=torch.stack(lat_steps), seconds=time.time() - t0), os.path.join(GOLD, f".partial_{name}.pt"))

    vid = pipe(i["ref_image"], list(i["poses"]), i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"],
               generator=torch.manual_seed(i["gen_seed"]), callback=cb).videos
    from transformers import CLIPImageProcessor
    clip = models["image_encoder"](CLIPImageProcessor().preprocess(
        i["ref_image"].resize((224, 224)), return_tensors="pt").pixel_values).image_embeds
    fr = list(i["frames"])
    u8 = (vid[0, :, fr].permute(1, 2, 3, 0) * 255.0).round().clamp(0, 255).to(torch.uint8).contiguous()
    out = dict(latents_f16=torch.stack(lat_steps), frames_u8=u8, frames=torch.tensor(fr),
               video_mean=vid.double().mean().float(), clip_embeds=clip.float(), seconds=time.time() - t0)
    print(f"{name}: video {tuple(vid.shape)} mean {float(vid.mean()):.6f} in {time.time() - t0:.0f} s", flush=True)
    return out


@torch.no_grad()
def finish_case(models, name):
    """An interrupted run whose DDIM loop completed (every step's latents are in tests/golden/.partial_<name>.pt): decode the
    stored frames with the reference's own `decode_latents` arithmetic (pipeline_pose2vid_long.py:113-126: 1 / 0.18215, the
    VAE frame by frame, / 2 + 0.5, clamp) — the pipeline decodes every frame independently, so decoding only the stored ones
    gives the same bytes — and compute the CLIP embedding as `run_case` does."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from golden_inputs import real_pipe_inputs
    i = real_pipe_inputs(name)
    part = torch.load(os.path.join(GOLD, f".partial_{name}.pt"))
    lat = part["latents_f16"]
    assert lat.shape[0] == i["steps"], f"only {lat.shape[0]} of {i['steps']} steps were completed"
    final = lat[-1].float()                                   # (1, 4, L, h, w): what the pipeline hands to decode_latents
    fr = list(i["frames"])
    frames = []
    for k in fr:
        z = final[:, :, k] / 0.18215
        frames.append((models["vae"].decode(z).sample / 2 + 0.5).clamp(0, 1)[0])
    u8 = (torch.stack(frames).permute(0, 2, 3, 1) * 255.0).round().clamp(0, 255).to(torch.uint8).contiguous()
    from transformers import CLIPImageProcessor
    clip = models["image_encoder"](CLIPImageProcessor().preprocess(
        i["ref_image"].resize((224, 224)), return_tensors="pt").pixel_values).image_embeds
    return dict(latents_f16=lat, frames_u8=u8, frames=torch.tensor(fr), video_mean=torch.stack(frames).double().mean().float(),   # (of the stored frames only)
                clip_embeds=clip.float(), seconds=float(part["seconds"]), finished_from_partial=True)


def main():
    args = sys.argv[1:]
    finish = "--finish" in args
    names = [a for a in args if not a.startswith("--")] or ["l40_windows"]
    R.setup()
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    from aniportrait_amd import configs as C
    from aniportrait_amd.synthetic import fill_module_
    m = R.build_models(small=False, with_clip=False)
    enc = CLIPVisionModelWithProjection(CLIPVisionConfig(**dict(C.CLIP_SMALL, projection_dim=768)))
    m["image_encoder"] = fill_module_(enc, 0, "image_encoder768.").eval()   # tests/util.py clip_encoder_for(False)
    for name in names:
        res = finish_case(m, name) if finish else run_case(m, name)
        torch.save(res, os.path.join(GOLD, f"real_pipeline_{name}.pt"))
        part = os.path.join(GOLD, f".partial_{name}.pt")
        if os.path.isfile(part):
            os.remove(part)


if __name__ == "__main__":
    main()
