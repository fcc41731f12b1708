"""End-to-end parity of the HIP pipeline against the golden videos produced by the REFERENCE's own
pipelines (tests/golden/small_pipeline.pt).  GPU box: python tools/gpu_check_pipeline.py"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from aniportrait_amd import configs as C  # noqa: E402
from aniportrait_amd.scheduling_ddim import DDIMScheduler  # noqa: E402
from aniportrait_amd.synthetic import fill_module_  # noqa: E402
from golden_inputs import PIPE_CASES, pipe_inputs  # noqa: E402
from gpu_check_models import build  # noqa: E402
from util import load_golden  # noqa: E402

DEV = "cuda"


from util import small_clip_encoder as small_clip  # noqa: E402,F401


def psnr(a, b):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * __import__("math").log10(mse)


@torch.no_grad()
def main():
    from src.pipelines.pipeline_pose2vid import Pose2VideoPipeline as ShortPipe
    from src.pipelines.pipeline_pose2vid_long import Pose2VideoPipeline as LongPipe
    gold = load_golden("small_pipeline.pt")
    m, _ = build(True)
    enc = small_clip()
    for name in PIPE_CASES:
        i = pipe_inputs(name)
        cls = LongPipe if i["long"] else ShortPipe
        pipe = cls(vae=m["vae"], image_encoder=enc, reference_unet=m["reference_unet"],
                   denoising_unet=m["denoising_unet"], pose_guider=m["pose_guider"], scheduler=DDIMScheduler(**C.DDIM_V2))
        pipe.set_progress_bar_config(disable=True)
        t0 = time.time()
        vid = pipe(i["ref_image"], list(i["poses"]), i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"],
                   generator=torch.manual_seed(42), latents=i["latents"], **i["kw"]).videos
        dt = time.time() - t0
        ref = gold[name + "/video_f16"].float()
        print(f"{name:16s} shape={tuple(vid.shape)} psnr={psnr(vid, ref):.2f} dB  mean={vid.mean():.5f} "
              f"ref_mean={gold[name + '/video_mean'].item():.5f}  {dt:.2f}s", flush=True)


if __name__ == "__main__":
    main()
