"""One eager (un-captured) denoising UNet3D forward at the BASELINE C2 shapes — CFG batch of 2 x 16 frames at 64x64
latents, reference banks attached, motion modules live — for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE): 25 of
these are 92 % of a clip's GPU time and carry every hot kernel family at its pipeline shapes.  (rocprofv3's counter
collection segfaults on the full bench.py process — the transformers CLIP tower — and cannot follow hipGraph replays.)
Every wrapper call (family, shape descriptor, algorithmic FLOP / bytes) is traced in launch order and written to
$ANIP_CALL_TRACE (default gpurun_out/pmc_calls.json): tools/pmc_summarize.py --calls pairs that list with the counter
rows by dispatch order, which is what separates the 3x3 convs from the Linears and one shape from another.
usage: python tools/pmc_unet_step.py [n_forwards]"""
import json
import os
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from aniportrait_amd import configs as C  # noqa: E402
from aniportrait_amd import hipops  # noqa: E402
from aniportrait_amd.mutual_self_attention import ReferenceAttentionControl  # noqa: E402
from aniportrait_amd.params import skip_init  # noqa: E402
from aniportrait_amd.pipeline_pose2vid_long import bank_shapes  # noqa: E402
from aniportrait_amd.synthetic import fast_fill_  # noqa: E402
from aniportrait_amd.unet import UNet3DConditionModel  # noqa: E402

DEV = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
with skip_init():
    net = UNet3DConditionModel(**C.unet3d_kwargs(False))
net = fast_fill_(net.to(DEV, torch.float16), 3)
rd = ReferenceAttentionControl(net, do_classifier_free_guidance=True, mode="read", fusion_blocks="full")
g = torch.Generator(device=DEV).manual_seed(0)
f, h = 16, 64
for p, shp in bank_shapes(net.config, 2, h, h).items():
    net._ref_blocks[p].node.bank = [torch.randn(shp, generator=g, device=DEV).half()]
x = torch.randn((2 * f, h, h, 4), generator=g, device=DEV).half()
ehs = torch.cat([torch.zeros(1, 1, 768, device=DEV), torch.randn((1, 1, 768), generator=g, device=DEV)]).half()
pose = [torch.randn(s, generator=g, device=DEV).half() for s in
        ((2 * f, 64, 64, 320), (2 * f, 32, 32, 320), (2 * f, 16, 16, 640), (2 * f, 8, 8, 1280), (2 * f, 8, 8, 1280))]
with hipops.trace_calls() as tr:
    net.forward_nhwc(x, 2, f, 519, ehs, pose)          # warm-up: packs weights, projects the banks
    torch.cuda.synchronize()
    for _ in range(n):
        net.forward_nhwc(x, 2, f, 519, ehs, pose, attn2_refresh=False)
    torch.cuda.synchronize()
out = os.environ.get("ANIP_CALL_TRACE") or os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_calls.json")
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
with open(out, "w") as fh:
    json.dump(dict(forwards=n + 1, commit=os.environ.get("ANIP_COMMIT"), calls=tr.calls), fh)
print("done", n, len(tr.calls), "wrapper calls ->", out)
