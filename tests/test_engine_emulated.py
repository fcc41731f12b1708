"""Host logic above the C ABI, on the CPU: engine.py's walk of the networks, weight packing and the pipeline's
window / CFG / DDIM orchestration run on a plain-PyTorch emulation of the kernel wrappers (tests/emu_hipops.py,
test infrastructure) and are compared with the goldens made by the reference's own code.  The kernels
themselves are checked on the GPU (test_hip_ops.py, test_gpu_models.py); the product refuses to run off-GPU."""
import pytest
import torch

import emu_hipops
from util import build_hip_models, load_golden, psnr, rel_err, small_clip_encoder

TOL = 6e-3
# PoseGuider: 16 conv + batch-statistics BatchNorm layers and 4 transformer blocks, each rounding to fp16 on
# store, against the fp32 reference: observed 6e-3 of the tensor max at the deepest maps
POSE_TOL = 1.5e-2


@pytest.fixture()
def emu(monkeypatch):
    emu_hipops.install(monkeypatch)


@pytest.fixture(scope="module")
def gold():
    return load_golden("small_models.pt")


@torch.no_grad()
def test_pose_guider_engine_matches_reference(emu, gold):
    from golden_inputs import unet_case
    m, _ = build_hip_models(True, keys=("pose_guider",), device="cpu")
    pg = m["pose_guider"]
    c = unet_case(True)
    from aniportrait_amd import hipops as ops
    fea = pg.forward_nhwc(ops.ncfhw_to_nhwc(c["pose"].half()))
    assert len(fea) == 5
    for i, f_ in enumerate(fea):
        got = ops.nhwc_to_ncfhw(f_, 2, out_f32=True)
        assert rel_err(got, gold[f"pose_fea/{i}"].float()) < POSE_TOL, i
    # eval mode: running statistics, same as the module's own torch path
    pg.eval()
    ev = pg.forward_nhwc(ops.ncfhw_to_nhwc(c["pose"].half()))
    ref = pg.float()(c["pose"].half().float(), None)
    assert rel_err(ops.nhwc_to_ncfhw(ev[4], 2, out_f32=True), ref[4]) < POSE_TOL


@torch.no_grad()
def test_unets_and_vae_engine_match_reference(emu, gold):
    from golden_inputs import unet_case, vae_case
    from src.models.mutual_self_attention import ReferenceAttentionControl
    m, _ = build_hip_models(True, keys=("denoising_unet", "reference_unet", "vae"), device="cpu")
    c = unet_case(True)
    wr = ReferenceAttentionControl(m["reference_unet"], do_classifier_free_guidance=True, mode="write", batch_size=1,
                                   fusion_blocks="full")
    rd = ReferenceAttentionControl(m["denoising_unet"], do_classifier_free_guidance=True, mode="read", batch_size=1,
                                   fusion_blocks="full")
    m["reference_unet"](c["ref_lat"].repeat(2, 1, 1, 1), torch.zeros((), dtype=torch.long),
                        encoder_hidden_states=c["ehs"], return_dict=False)
    rd.update(wr)
    for p, rb in m["denoising_unet"]._ref_blocks.items():
        assert rel_err(rb.node.bank[0].float(), gold["bank/" + p].float()) < TOL, p
    pose = [gold[f"pose_fea/{i}"] for i in range(5)]
    out = m["denoising_unet"](c["lat"], torch.tensor(c["t"]), encoder_hidden_states=c["ehs"], pose_cond_fea=pose)
    assert rel_err(out.sample, gold["unet_out"]) < TOL
    rd.clear(); wr.clear()
    v = vae_case(16, 16)
    assert rel_err(m["vae"].decode(v["z"]).sample, gold["vae_dec"]) < TOL
    assert rel_err(m["vae"].encode(v["x"]).latent_dist.mean, gold["vae_enc"]) < TOL


@torch.no_grad()
@pytest.mark.parametrize("case", ["long_L4", "long_L10_ctx8"])
def test_pipeline_host_logic_matches_reference_video(emu, case):
    from aniportrait_amd import configs as C
    from aniportrait_amd.scheduling_ddim import DDIMScheduler
    from golden_inputs import pipe_inputs
    from src.pipelines.pipeline_pose2vid_long import Pose2VideoPipeline
    gold = load_golden("small_pipeline.pt")
    m, _ = build_hip_models(True, device="cpu")
    i = pipe_inputs(case)
    pipe = Pose2VideoPipeline(vae=m["vae"], image_encoder=small_clip_encoder("cpu"), reference_unet=m["reference_unet"],
                              denoising_unet=m["denoising_unet"], pose_guider=m["pose_guider"],
                              scheduler=DDIMScheduler(**C.DDIM_V2))
    pipe.set_progress_bar_config(disable=True)
    vid = pipe(i["ref_image"], list(i["poses"]), i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"],
               latents=i["latents"], **i["kw"]).videos
    assert psnr(vid, gold[case + "/video_f16"].float()) >= 40.0
    if case == "long_L4":
        # display bytes made on the device == what save_videos_grid computes from the fp32 video on the host
        # (src/utils/util.py:97-98: (x * 255).numpy().astype(np.uint8)); numpy output type as in the reference
        u8 = pipe(i["ref_image"], list(i["poses"]), i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"],
                  latents=i["latents"], output_type="uint8", **i["kw"]).videos
        want = (vid[0].permute(1, 2, 3, 0) * 255).numpy().astype("uint8")       # (L, H, W, 3)
        assert u8.dtype == torch.uint8 and tuple(u8.shape) == (i["L"], i["H"], i["W"], 3)
        assert (u8.numpy() == want).all()


@torch.no_grad()
def test_inference_v1_groupnorm_variant_engine_matches_reference(emu):
    """use_inflated_groupnorm=False (configs/inference/inference_v1.yaml): GroupNorm statistics over the sample's frames in
    the ResnetBlock3D norms and conv_norm_out — the engine's `frames_per_stat` path against the reference-made golden"""
    import copy

    from aniportrait_amd import configs as C
    from aniportrait_amd.unet import UNet3DConditionModel
    from golden_inputs import unet_case
    from util import oracle_state_dicts
    kw = copy.deepcopy(C.unet3d_kwargs(True))
    kw.update(use_inflated_groupnorm=False, motion_module_mid_block=False)
    kw["motion_module_kwargs"]["temporal_position_encoding_max_len"] = 24
    net = UNet3DConditionModel(**kw)
    sd = oracle_state_dicts(True, keys=["denoising_unet"])["denoising_unet"]
    missing, unexpected = net.load_state_dict({k: v for k, v in sd.items() if not k.startswith("mid_block.motion_modules")},
                                              strict=False)
    assert not unexpected and all(m.endswith(".pe") for m in missing), (missing[:3], unexpected[:3])
    net = net.to("cpu", torch.float16)
    c = unet_case(True)
    out = net(c["lat"], torch.tensor(c["t"]), encoder_hidden_states=c["ehs"], pose_cond_fea=None)
    gold = load_golden("small_models_v1.pt")
    assert rel_err(out.sample, gold["unet_out_v1"]) < TOL
    assert rel_err(out.sample, gold["unet_out_v1_if_inflated"]) > 1e-2


def _v1_models(device):
    """the product's modules under configs/inference/inference_v1.yaml (UNet3D variant), name-hash weights as the golden's"""
    from aniportrait_amd import configs as C
    from aniportrait_amd.unet import UNet3DConditionModel
    from util import oracle_state_dicts
    m, _ = build_hip_models(True, keys=("reference_unet", "vae", "pose_guider"), device=device)
    net = UNet3DConditionModel(**C.unet3d_kwargs_v1(True))
    sd = oracle_state_dicts(True, keys=["denoising_unet"])["denoising_unet"]
    missing, unexpected = net.load_state_dict({k: v for k, v in sd.items() if not k.startswith("mid_block.motion_modules")},
                                              strict=False)
    assert not unexpected and all(x.endswith(".pe") for x in missing)
    m["denoising_unet"] = net.to(device, torch.float16)
    return m


@torch.no_grad()
@pytest.mark.parametrize("case", ["long_L4", "long_L10_ctx8"])
def test_inference_v1_pipeline_matches_reference_video(emu, case):
    """configs/inference/inference_v1.yaml END TO END: its UNet3D variant and its scheduler (epsilon prediction, leading spacing,
    no zero-SNR) through the pipeline's fused CFG + DDIM step, against the video the reference's own pipeline produced with the
    same configuration (oracle/make_golden.py --v1-pipeline)"""
    from aniportrait_amd import configs as C
    from aniportrait_amd.scheduling_ddim import DDIMScheduler
    from golden_inputs import pipe_inputs
    from src.pipelines.pipeline_pose2vid_long import Pose2VideoPipeline
    gold = load_golden("small_pipeline_v1.pt")
    m = _v1_models("cpu")
    i = pipe_inputs(case)
    pipe = Pose2VideoPipeline(vae=m["vae"], image_encoder=small_clip_encoder("cpu"), reference_unet=m["reference_unet"],
                              denoising_unet=m["denoising_unet"], pose_guider=m["pose_guider"],
                              scheduler=DDIMScheduler(**C.DDIM_V1))
    pipe.set_progress_bar_config(disable=True)
    vid = pipe(i["ref_image"], list(i["poses"]), i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"],
               latents=i["latents"], **i["kw"]).videos
    assert psnr(vid, gold[case + "/video_f16"].float()) >= 40.0
    # and it is not the v2 schedule in disguise
    assert psnr(load_golden("small_pipeline.pt")[case + "/video_f16"].float(), gold[case + "/video_f16"].float()) < 30.0


@torch.no_grad()
def test_fused_temporal_branch_of_the_motion_module_matches_the_three_launch_branch(emu, monkeypatch):
    """C = 320, F = 16 (the 64x64 level): engine.motion_module through anip_temporal_qkv_attention's packing (head-pair row order
    of [to_q; to_k; to_v], bias + positional-encoding table) against the LayerNorm -> GEMM -> temporal attention branch and
    against plain fp32 arithmetic of src/models/motion_module.py:236-259,351-388"""
    import torch.nn.functional as Fn

    from aniportrait_amd import engine
    g = torch.Generator().manual_seed(5)
    C, heads, f, b, H, W = 320, 8, 16, 1, 2, 4
    r = lambda *s, scale=1.0: (torch.randn(s, generator=g) * scale).half().float()
    p = "mm.temporal_transformer"
    bp = p + ".transformer_blocks.0"
    sd = {p + ".norm.weight": 1 + 0.1 * r(C), p + ".norm.bias": 0.1 * r(C),
          p + ".proj_in.weight": r(C, C, scale=C ** -0.5), p + ".proj_in.bias": 0.1 * r(C),
          p + ".proj_out.weight": r(C, C, scale=C ** -0.5), p + ".proj_out.bias": 0.1 * r(C),
          bp + ".ff_norm.weight": 1 + 0.1 * r(C), bp + ".ff_norm.bias": 0.1 * r(C),
          bp + ".ff.net.0.proj.weight": r(8 * C, C, scale=C ** -0.5), bp + ".ff.net.0.proj.bias": 0.1 * r(8 * C),
          bp + ".ff.net.2.weight": r(C, 4 * C, scale=(4 * C) ** -0.5), bp + ".ff.net.2.bias": 0.1 * r(C)}
    for i in range(2):
        ap = bp + f".attention_blocks.{i}"
        for n in ("to_q", "to_k", "to_v"):
            sd[ap + f".{n}.weight"] = r(C, C, scale=C ** -0.5)
        sd[ap + ".to_out.0.weight"] = r(C, C, scale=C ** -0.5)
        sd[ap + ".to_out.0.bias"] = 0.1 * r(C)
        sd[ap + ".pos_encoder.pe"] = 0.5 * r(1, 24, C)
        sd[bp + f".norms.{i}.weight"] = 1 + 0.1 * r(C)
        sd[bp + f".norms.{i}.bias"] = 0.1 * r(C)
    x = r(b * f, H, W, C).half()
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(engine, "_FUSED_TEMPORAL", fused)
        outs[fused] = engine.motion_module(engine.PackedNet(sd, "cpu"), "mm", x, b, f, heads).float()
    assert rel_err(outs[True], outs[False]) < 2e-3
    # fp32 restatement of the reference block
    T, d = H * W, C // heads
    xs = x.float().reshape(b * f, T, C)
    h = Fn.group_norm(xs.permute(0, 2, 1), 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6).permute(0, 2, 1)
    h = h @ sd[p + ".proj_in.weight"].t() + sd[p + ".proj_in.bias"]                        # (b f, T, C)
    for i in range(2):
        ap = bp + f".attention_blocks.{i}"
        nh = Fn.layer_norm(h, (C,), sd[bp + f".norms.{i}.weight"], sd[bp + f".norms.{i}.bias"])
        t = nh.reshape(b, f, T, C).permute(0, 2, 1, 3).reshape(b * T, f, C) + sd[ap + ".pos_encoder.pe"][:, :f]
        q, k, v = ((t @ sd[ap + f".{n}.weight"].t()).reshape(b * T, f, heads, d).permute(0, 2, 1, 3) for n in ("to_q", "to_k", "to_v"))
        o = (torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, -1) @ v).permute(0, 2, 1, 3).reshape(b * T, f, C)
        o = o @ sd[ap + ".to_out.0.weight"].t() + sd[ap + ".to_out.0.bias"]
        h = h + o.reshape(b, T, f, C).permute(0, 2, 1, 3).reshape(b * f, T, C)
    nh = Fn.layer_norm(h, (C,), sd[bp + ".ff_norm.weight"], sd[bp + ".ff_norm.bias"])
    u = nh @ sd[bp + ".ff.net.0.proj.weight"].t() + sd[bp + ".ff.net.0.proj.bias"]
    h = h + (u[..., :4 * C] * Fn.gelu(u[..., 4 * C:])) @ sd[bp + ".ff.net.2.weight"].t() + sd[bp + ".ff.net.2.bias"]
    ref = (h @ sd[p + ".proj_out.weight"].t() + sd[p + ".proj_out.bias"] + xs).reshape(b * f, H, W, C)
    assert rel_err(outs[True], ref) < TOL


@torch.no_grad()
def test_row_stationary_branches_of_the_spatial_transformer_match_the_separate_launches(emu, monkeypatch):
    """C = 320, M % 128 == 0 (the 64x64 level): engine.spatial_transformer through anip_groupnorm_scale_shift +
    anip_affine_linear320 (GroupNorm inside proj_in) and anip_ln_qkv_projection (norm1 inside the q | k | v^T projections),
    against the groupnorm -> gemm / layernorm -> three GEMMs branches — plain, write and read mode"""
    from aniportrait_amd import engine
    g = torch.Generator().manual_seed(7)
    C, heads, N, H, W = 320, 8, 2, 8, 16
    r = lambda *s, scale=1.0: (torch.randn(s, generator=g) * scale).half().float()
    p = "st"
    bp = p + ".transformer_blocks.0"
    sd = {p + ".norm.weight": 1 + 0.1 * r(C), p + ".norm.bias": 0.1 * r(C),
          p + ".proj_in.weight": r(C, C, 1, 1, scale=C ** -0.5), p + ".proj_in.bias": 0.1 * r(C),
          p + ".proj_out.weight": r(C, C, 1, 1, scale=C ** -0.5), p + ".proj_out.bias": 0.1 * r(C),
          bp + ".norm1.weight": 1 + 0.1 * r(C), bp + ".norm1.bias": 0.1 * r(C),
          bp + ".norm3.weight": 1 + 0.1 * r(C), bp + ".norm3.bias": 0.1 * r(C),
          bp + ".attn1.to_out.0.weight": r(C, C, scale=C ** -0.5), bp + ".attn1.to_out.0.bias": 0.1 * r(C),
          bp + ".ff.net.0.proj.weight": r(8 * C, C, scale=C ** -0.5), bp + ".ff.net.0.proj.bias": 0.1 * r(8 * C),
          bp + ".ff.net.2.weight": r(C, 4 * C, scale=(4 * C) ** -0.5), bp + ".ff.net.2.bias": 0.1 * r(C)}
    for n in ("to_q", "to_k", "to_v"):
        sd[bp + f".attn1.{n}.weight"] = r(C, C, scale=C ** -0.5)
    x = (r(N, H, W, C) + 0.5).half()
    attn2 = 0.1 * r(N, C)
    bank = r(N, H * W, C).half()
    for mode in ("plain", "read", "write"):
        outs = {}
        for fused in (True, False):
            monkeypatch.setattr(engine, "_FUSED_ROWS", fused)
            ref = engine.RefState()
            ref.mode = mode
            ridx = None
            if mode == "read":
                ref.bank = bank
                ridx = (torch.tensor([-1, 1], dtype=torch.int32), 1)
            out = engine.spatial_transformer(engine.PackedNet(sd, "cpu"), p, x, heads, attn2, 1, ref=ref, ref_index=ridx)
            outs[fused] = (out.float(), None if ref.written is None else ref.written.float())
        assert rel_err(outs[True][0], outs[False][0]) < 3e-3, mode
        if mode == "write":
            assert rel_err(outs[True][1], outs[False][1]) < 3e-3


@torch.no_grad()
def test_cfg_shared_prefix_of_the_denoising_unet_changes_nothing(emu, gold, monkeypatch):
    """unet_forward(cfg_shared_input=True): the first ResnetBlock3D, the transformer's norm / proj_in, norm1 and the q / k / v
    projections of down_blocks.0 run once for the two CFG halves (identical latents, pose features and timestep), the reference
    attention reads them for both (frame n attends with the tokens of frame n % f) — same output as the full-batch walk, and
    the reference-made golden still matches"""
    from aniportrait_amd import engine
    from golden_inputs import unet_case
    from src.models.mutual_self_attention import ReferenceAttentionControl
    m, _ = build_hip_models(True, keys=("denoising_unet", "reference_unet"), device="cpu")
    c = unet_case(True)
    wr = ReferenceAttentionControl(m["reference_unet"], do_classifier_free_guidance=True, mode="write", batch_size=1,
                                   fusion_blocks="full")
    rd = ReferenceAttentionControl(m["denoising_unet"], do_classifier_free_guidance=True, mode="read", batch_size=1,
                                   fusion_blocks="full")
    m["reference_unet"](c["ref_lat"].repeat(2, 1, 1, 1), torch.zeros((), dtype=torch.long),
                        encoder_hidden_states=c["ehs"], return_dict=False)
    rd.update(wr)
    unet = m["denoising_unet"]
    b, _, f, h, w = c["lat"].shape
    x = c["lat"].permute(0, 2, 3, 4, 1).reshape(b * f, h, w, -1).half().contiguous()
    pose = [gold[f"pose_fea/{i}"] for i in range(5)]
    pose = [p_.permute(0, 2, 3, 4, 1).reshape(b * f, p_.shape[3], p_.shape[4], -1).half().contiguous() for p_ in pose]
    assert torch.equal(x[:f], x[f:]) and all(torch.equal(p_[:f], p_[f:]) for p_ in pose)     # the guarantee the flag states
    outs = {}
    for shared in (False, True):
        outs[shared] = unet.forward_nhwc(x, b, f, c["t"], c["ehs"], pose, cfg_shared_input=shared).float()
    assert not torch.equal(outs[True][:f], outs[True][f:])            # the halves do diverge (reference index, attn2)
    assert rel_err(outs[True], outs[False]) < 2e-3
    want = gold["unet_out"].permute(0, 2, 3, 4, 1).reshape(b * f, h, w, -1).float()
    assert rel_err(outs[True], want) < TOL
    monkeypatch.setattr(engine, "_SHARE_CFG_PREFIX", False)
    assert torch.equal(unet.forward_nhwc(x, b, f, c["t"], c["ehs"], pose, cfg_shared_input=True).float(), outs[False])
    rd.clear(); wr.clear()


@torch.no_grad()
@pytest.mark.parametrize("batch", [1, 2])
def test_clip_vision_engine_matches_transformers(emu, batch):
    """engine.clip_vision_forward (round 6: the CLIP image encoder on the HIP kernels; here on their emulation) against the
    transformers module it adopts, fp32 on the CPU (pipeline_pose2vid_long.py:379-385): host-side im2col incl. the class
    token's zero row and the K padding, class + position table as the patch GEMM's residual, pre-scaled q bias, quick-GELU,
    post-LayerNorm of the class token, projection"""
    from aniportrait_amd.clip_vision import CLIPVisionHip, patch_rows
    enc = small_clip_encoder("cpu")
    g = torch.Generator().manual_seed(11)
    px = torch.randn((batch, 3, 224, 224), generator=g)
    ref = enc(px).image_embeds
    hip = CLIPVisionHip.from_module(enc)
    rows = patch_rows(px, enc.config.patch_size)
    assert rows.shape == (batch * 50, 3072) and float(rows[0].abs().max()) == 0.0
    got = hip.image_embeds_from_rows(rows)
    assert got.shape == ref.shape and got.dtype == torch.float16
    assert rel_err(got.float(), ref) < 4e-3
    assert rel_err(hip.image_embeds(px).float(), ref) < 4e-3
    # 588 = 3 * 14 * 14 columns (ViT-L/14) are padded to a multiple of 64 with zeros
    assert patch_rows(torch.randn(1, 3, 28, 28), 14).shape == (5, 640)
    with pytest.raises(ValueError):
        hip.image_embeds(torch.randn(1, 3, 192, 192))
    bad = small_clip_encoder("cpu")
    bad.config.hidden_act = "gelu"
    with pytest.raises(NotImplementedError):
        CLIPVisionHip.from_module(bad)
