"""TEST INFRASTRUCTURE (run by hand, not collected by pytest) — where does the fp16 path lose its bits?

Round 2 measured, at real width and identical inputs, 61 dB (256x256, T = 1024 tokens) but 46 dB (512x512, T = 4096)
against the fp32 CPU oracle.  This tool takes ONE denoising UNet3D forward (CFG batch of 2 x f frames, reference
attention live with banks made by the oracle's ReferenceNet, motion modules live) at several latent sizes and compares
every block output of

  hip     the product (HIP kernels; needs the GPU)
  emu     tests/emu_hipops.py: the same engine walk on the CPU with fp32 arithmetic and ONE rounding to fp16 per
          kernel — i.e. an ideal fp16-storage implementation of the same op sequence (runs anywhere)
  o16     the fp32 oracle with every BLOCK output rounded to fp16 (the coarsest possible fp16 storage)

with the fp32 oracle, block by block:  rel_rms = |a - b|_2 / |b|_2,  rel_max = max|a - b| / max|b|.
If `hip` tracks `emu`, the kernels add nothing beyond what fp16 storage of this op sequence costs; if the growth with
the token count shows in `emu` / `o16` too, it belongs to the network (random weights), not to a kernel.

    python tests/bisect_parity.py --sizes 32 64 --frames 4 --backends emu o16 [hip] --out profiles/r03/bisect.json
    (ANIP_FUSED_FFN=0 selects the two-GEMM feed-forward in `hip`.)
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


class _Patch:
    """the two-method slice of pytest's monkeypatch that emu_hipops.install uses (nothing is undone: one process)"""

    @staticmethod
    def setattr(obj, name, value):
        setattr(obj, name, value)


def stats(a, b):
    a, b = a.double(), b.double()
    d = a - b
    return dict(rel_rms=float(d.norm() / b.norm().clamp_min(1e-30)), rel_max=float(d.abs().max() / b.abs().max().clamp_min(1e-30)),
                ref_rms=float(b.pow(2).mean().sqrt()))


@torch.no_grad()
def run_size(h, f, backends, seed=0):
    from aniportrait_amd import configs as C
    from oracle import ref_torch as O
    from util import build_hip_models, oracle_state_dicts

    cfg = C.unet3d_kwargs(False)
    sds = oracle_state_dicts(False, keys=["denoising_unet", "reference_unet"])
    g = torch.Generator().manual_seed(100 + seed)
    clip = torch.randn((1, 1, 768), generator=g)
    ehs = torch.cat([torch.zeros_like(clip), clip])
    lat = torch.randn((1, 4, f, h, h), generator=g)
    ref_lat = torch.randn((1, 4, h, h), generator=g) * 0.18215 * 4
    t = 519
    t0 = time.time()
    banks = O.refnet_forward(sds["reference_unet"], cfg, ref_lat.repeat(2, 1, 1, 1), 0, ehs)
    want = {}

    def rec(name, x):
        want[name] = x.clone()
        return x

    O.unet3d_forward(sds["denoising_unet"], cfg, lat.repeat(2, 1, 1, 1, 1), t, ehs, None, banks, True, tap=rec)
    t_oracle = time.time() - t0
    order = list(want)
    out = {"h": h, "frames": f, "tokens": h * h, "oracle_seconds": t_oracle, "blocks": order, "backends": {}}

    for be in backends:
        got = {}
        t0 = time.time()
        if be == "o16":
            def rec16(name, x):
                x = x.half().float()
                got[name] = x
                return x
            O.unet3d_forward(sds["denoising_unet"], cfg, lat.repeat(2, 1, 1, 1, 1), t, ehs, None, banks, True, tap=rec16)
        else:
            dev = "cuda" if be == "hip" else "cpu"
            if be == "emu":
                import emu_hipops
                emu_hipops.install(_Patch)
            from aniportrait_amd import hipops as ops
            from aniportrait_amd.mutual_self_attention import ReferenceAttentionControl
            m, _ = build_hip_models(False, keys=("denoising_unet",), device=dev)
            net = m["denoising_unet"]
            rd = ReferenceAttentionControl(net, do_classifier_free_guidance=True, mode="read", fusion_blocks="full")
            for p, bank in banks.items():
                net._ref_blocks[p].node.bank = [bank.to(dev, torch.float16)]
            x = ops.ncfhw_to_nhwc(lat.repeat(2, 1, 1, 1, 1).to(dev).half())

            def rec_nhwc(name, xx):
                got[name] = xx.detach().float().cpu().permute(0, 3, 1, 2)

            net.forward_nhwc(x, 2, f, t, ehs.to(dev), None, tap=rec_nhwc)
            rd.clear()
            del net, m
        rows = {}
        for name in order:
            if name in got:
                a = got[name]
                b = want[name]
                if a.dim() == 5:
                    a = a.permute(0, 2, 1, 3, 4).reshape(b.shape)
                rows[name] = stats(a.reshape(b.shape), b)
        out["backends"][be] = {"seconds": time.time() - t0, "rows": rows,
                               "fused_ffn": os.environ.get("ANIP_FUSED_FFN", "1") if be == "hip" else None}
        last = rows[order[-1]]
        print(f"h={h} f={f} {be:>4}: conv_out rel_rms {last['rel_rms']:.3e} rel_max {last['rel_max']:.3e} "
              f"({time.time() - t0:.0f} s; oracle {t_oracle:.0f} s)", flush=True)
    return out


@torch.no_grad()
def run_vae(h, n, backends, seed=0):
    """VAE decode of n latent frames (h x h): the stage that turns a 55-70 dB latent into the 46 dB frames at 512x512"""
    from aniportrait_amd import configs as C
    from oracle import ref_torch as O
    from util import build_hip_models, oracle_state_dicts

    cfg = C.SD_VAE_FT_MSE
    sds = oracle_state_dicts(False, keys=["vae"])
    g = torch.Generator().manual_seed(200 + seed)
    z = torch.randn((n, 4, h, h), generator=g)      # ~ latents / 0.18215 after a few DDIM steps: unit scale
    t0 = time.time()
    want = {}

    def rec(name, x):
        want[name] = x.clone()
        return x

    O.vae_decode(sds["vae"], cfg, z, tap=rec)
    t_oracle = time.time() - t0
    order = list(want)
    out = {"net": "vae_decode", "h": h, "frames": n, "tokens": h * h, "oracle_seconds": t_oracle, "blocks": order, "backends": {}}
    for be in backends:
        got = {}
        t0 = time.time()
        if be == "o16":
            def rec16(name, x):
                x = x.half().float()
                got[name] = x
                return x
            O.vae_decode(sds["vae"], cfg, z, tap=rec16)
        else:
            dev = "cuda" if be == "hip" else "cpu"
            if be == "emu":
                import emu_hipops
                emu_hipops.install(_Patch)
            from aniportrait_amd import hipops as ops
            m, _ = build_hip_models(False, keys=("vae",), device=dev)
            x = ops.ncfhw_to_nhwc(z.unsqueeze(2).to(dev).half())          # (n, 4, 1, h, h) -> (n, h, h, 4)

            def rec_nhwc(name, xx):
                got[name] = xx.detach().float().cpu().permute(0, 3, 1, 2)

            m["vae"].decode_nhwc(x, tap=rec_nhwc)
            del m
        rows = {name: stats(got[name].reshape(want[name].shape), want[name]) for name in order if name in got}
        # the pipeline's metric on this stage alone: PSNR of the decoded frames in [0, 1]
        a = (got[order[-1]].reshape(want[order[-1]].shape) / 2 + 0.5).clamp(0, 1)
        b = (want[order[-1]] / 2 + 0.5).clamp(0, 1)
        mse = float(((a.double() - b.double()) ** 2).mean())
        import math
        out["backends"][be] = {"seconds": time.time() - t0, "rows": rows, "frame_psnr_db": 10 * math.log10(1 / max(mse, 1e-30)),
                               "saturated_fraction": float(((want[order[-1]] / 2 + 0.5 <= 0) | (want[order[-1]] / 2 + 0.5 >= 1)).float().mean())}
        last = rows[order[-1]]
        print(f"VAE h={h} n={n} {be:>4}: conv_out rel_rms {last['rel_rms']:.3e} rel_max {last['rel_max']:.3e} frame PSNR "
              f"{out['backends'][be]['frame_psnr_db']:.1f} dB ({time.time() - t0:.0f} s; oracle {t_oracle:.0f} s)", flush=True)
    return out


@torch.no_grad()
def run_vae_batch(h, n, backends, seed=0):
    """frames are independent in the VAE decode: a batch of n frames must reproduce the n single-frame decodes block by
    block.  `hip` only (no oracle involved); rows: batched vs one-by-one outputs of frames 0 and n-1."""
    from aniportrait_amd import hipops as ops
    from util import build_hip_models
    g = torch.Generator().manual_seed(300 + seed)
    z = (torch.randn((n, 4, h, h), generator=g) * 5.0)
    m, _ = build_hip_models(False, keys=("vae",), device="cuda")
    x = ops.ncfhw_to_nhwc(z.unsqueeze(2).cuda().half())
    order, batch, single = [], {}, {}

    def rec_b(name, xx):
        order.append(name)
        batch[name] = xx[[0, n - 1]].detach().float().cpu()

    m["vae"].decode_nhwc(x, tap=rec_b)
    for k, fi in enumerate((0, n - 1)):
        def rec_s(name, xx, k=k):
            single.setdefault(name, [None, None])[k] = xx[0].detach().float().cpu()
        m["vae"].decode_nhwc(x[fi:fi + 1].contiguous(), tap=rec_s)
    rows = {name: stats(batch[name], torch.stack(single[name])) for name in order}
    a = (batch[order[-1]] / 2 + 0.5).clamp(0, 1)
    b = (torch.stack(single[order[-1]]) / 2 + 0.5).clamp(0, 1)
    mse = float(((a.double() - b.double()) ** 2).mean())
    import math
    ps = float("inf") if mse == 0 else 10 * math.log10(1 / mse)
    print(f"VAE batch-vs-single h={h} n={n}: frame PSNR {ps:.1f} dB; first differing block: "
          f"{next((nm for nm in order if rows[nm]['rel_max'] > 0), None)}", flush=True)
    return {"net": "vae_decode", "h": h, "frames": n, "tokens": h * h, "oracle_seconds": 0.0, "blocks": order,
            "backends": {"hip batch vs hip single": {"seconds": 0.0, "rows": rows, "frame_psnr_db": ps}}}


def table(res):
    lines = []
    for r in res:
        bes = list(r["backends"])
        if r.get("net") == "vae_decode":
            ps = ", ".join(f"{be}: {r['backends'][be]['frame_psnr_db']:.1f} dB" for be in bes)
            lines.append(f"## VAE decode of {r['frames']} frame(s), {r['h']}x{r['h']} latents -> {8 * r['h']}x{8 * r['h']} pixels — rel_rms "
                         f"(rel_max) vs fp32 oracle; frame PSNR {ps}")
        else:
            lines.append(f"## {r['h']}x{r['h']} latents (T = {r['tokens']}), CFG batch 2 x {r['frames']} frames — rel_rms (rel_max) vs fp32 oracle")
        lines.append("| block | " + " | ".join(bes) + " |")
        lines.append("|---|" + "---|" * len(bes))
        for name in r["blocks"]:
            cells = []
            for be in bes:
                s = r["backends"][be]["rows"].get(name)
                cells.append("-" if s is None else f"{s['rel_rms']:.2e} ({s['rel_max']:.2e})")
            lines.append(f"| {name} | " + " | ".join(cells) + " |")
        lines.append("")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[32, 64])
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--backends", nargs="+", default=["emu", "o16"])
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--net", default="unet", choices=["unet", "vae", "vaebatch"])
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    else:
        from util import oracle_threads
        oracle_threads()
    fn = {"vae": run_vae, "vaebatch": run_vae_batch, "unet": run_size}[a.net]
    res = [fn(h, a.frames, a.backends) for h in a.sizes]
    md = table(res)
    print(md)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)
        with open(os.path.splitext(a.out)[0] + ".md", "w") as fh:
            fh.write(md + "\n")


if __name__ == "__main__":
    main()
