"""Print model-level parity numbers of the HIP path against the golden fixtures made by the reference's
own code (tests/golden/*.pt).  GPU box: python tools/gpu_check_models.py [--real]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
from aniportrait_amd import configs as C  # noqa: E402
from aniportrait_amd.autoencoder_kl import AutoencoderKL  # noqa: E402
from aniportrait_amd.mutual_self_attention import ReferenceAttentionControl  # noqa: E402
from aniportrait_amd.pose_guider import PoseGuider  # noqa: E402
from aniportrait_amd.unet import UNet2DConditionModel, UNet3DConditionModel  # noqa: E402
from golden_inputs import unet_case, vae_case  # noqa: E402
from util import load_golden, oracle_state_dicts, rel_err  # noqa: E402

DEV = "cuda"


from util import build_hip_models as build  # noqa: E402,F401


def report(tag, out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    print(f"{tag:40s} rel_max_err={rel_err(out, ref):.3e}  rms_err/rms={((out-ref).pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item():.3e}",
          flush=True)


@torch.no_grad()
def check(small):
    name = "small" if small else "real"
    gold = load_golden(f"{name}_models.pt")
    m, sds = build(small)
    c = unet_case(small)
    wr = ReferenceAttentionControl(m["reference_unet"], do_classifier_free_guidance=True, mode="write", batch_size=1,
                                   fusion_blocks="full")
    rd = ReferenceAttentionControl(m["denoising_unet"], do_classifier_free_guidance=True, mode="read", batch_size=1,
                                   fusion_blocks="full")
    ehs = c["ehs"].to(DEV)
    t0 = time.time()
    m["reference_unet"](c["ref_lat"].repeat(2, 1, 1, 1).to(DEV), torch.zeros((), dtype=torch.long),
                        encoder_hidden_states=ehs, return_dict=False)
    rd.update(wr)
    torch.cuda.synchronize()
    print(f"[{name}] refnet {time.time()-t0:.2f}s")
    for p, rb in m["denoising_unet"]._ref_blocks.items():
        report(f"[{name}] bank {p}", rb.node.bank[0], gold["bank/" + p])
    if small:
        pose = [gold[f"pose_fea/{i}"].to(DEV) for i in range(5)]
        fea = m["pose_guider"](c["pose"].to(DEV, torch.float16), c["ref_pose"].to(DEV, torch.float16))
        for i, f_ in enumerate(fea):
            report(f"[{name}] pose_fea {i} (torch fp16)", f_, gold[f"pose_fea/{i}"])
    else:
        from oracle import ref_torch as O
        pose = [p.to(DEV) for p in O.pose_guider(sds["pose_guider"], c["pose"], c["ref_pose"])]
    for with_pose in (True, False):
        t0 = time.time()
        out = m["denoising_unet"](c["lat"].to(DEV), torch.tensor(c["t"]), encoder_hidden_states=ehs,
                                  pose_cond_fea=pose if with_pose else None, return_dict=False)[0]
        torch.cuda.synchronize()
        print(f"[{name}] unet3d {time.time()-t0:.2f}s")
        report(f"[{name}] unet_out pose={with_pose}", out, gold["unet_out" if with_pose else "unet_out_nopose"])
    rd.clear()
    wr.clear()
    v = vae_case(16, 16) if small else vae_case(32, 32)
    dec = m["vae"].decode(v["z"].to(DEV)).sample
    enc = m["vae"].encode(v["x"].to(DEV)).latent_dist.mean
    report(f"[{name}] vae_dec", dec, gold["vae_dec"])
    report(f"[{name}] vae_enc", enc, gold["vae_enc"])


if __name__ == "__main__":
    check(True)
    if "--real" in sys.argv:
        check(False)
