"""TEST INFRASTRUCTURE — the bodies of tests/test_dropin_surface.py, each run in its own interpreter
(`python tests/dropin_checks.py <check> [args]`) because they replace host-side modules (torchvision, cv2, ...) with
stubs and re-order `sys.path`, which must not leak into the rest of the test session."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = os.environ.get("ANIP_REFERENCE_ROOT", "/root/reference")


def _paths(with_reference):
    """repository FIRST, the reference checkout second: the deployment INTEGRATION.md §1 describes"""
    for p in (HERE, REPO) + ((REFERENCE,) if with_reference else ()):
        while p in sys.path:
            sys.path.remove(p)
    sys.path[:0] = [REPO, HERE] + ([REFERENCE] if with_reference else [])
    sys.dont_write_bytecode = True  # nothing may be written into the reference tree


def check_script_imports(script, last_line):
    """execute the reference script's own import block (e.g. scripts/pose2vid.py:1-30) with this repository first on
    the path: the shimmed modules resolve to aniportrait_amd, everything else falls through to the reference"""
    _paths(True)
    from dropin_driver import install_stub_modules
    install_stub_modules()
    with open(os.path.join(REFERENCE, "scripts", script)) as f:
        block = "".join(f.readlines()[: int(last_line)])
    ns = {}
    exec(compile(block, script, "exec"), ns)
    import diffusers
    for name in ("PoseGuider", "UNet2DConditionModel", "UNet3DConditionModel", "Pose2VideoPipeline"):
        mod = sys.modules[ns[name].__module__]
        assert ns[name].__module__.startswith("aniportrait_amd."), (name, ns[name].__module__)
        assert os.path.abspath(mod.__file__).startswith(REPO + os.sep), mod.__file__
    for shim in ("src.models.pose_guider", "src.models.unet_2d_condition", "src.models.unet_3d",
                 "src.pipelines.pipeline_pose2vid_long"):
        assert os.path.abspath(sys.modules[shim].__file__).startswith(os.path.join(REPO, "src") + os.sep)
    for name in ("LMKExtractor", "FaceMeshVisualizer"):
        mod = sys.modules[ns[name].__module__]
        assert os.path.abspath(mod.__file__).startswith(os.path.join(REFERENCE, "src") + os.sep), (name, mod.__file__)
    assert ns["save_videos_grid"].__module__ == "aniportrait_amd.video_io"   # the output side is shimmed too
    for name in ("get_fps", "read_frames"):                              # the pose-video readers: ours too (PyAV on first use),
        if name in ns:                                                   # so that the import block never executes the reference's util.py
            assert ns[name].__module__ == "aniportrait_amd.video_io", (name, ns[name].__module__)
    assert "src.utils._reference_util" not in sys.modules               # (cv2 / torchvision / einops at its module level)
    from src.utils.util import crop_face                                 # not ours: falls through to the reference's module
    assert os.path.abspath(sys.modules[crop_face.__module__].__file__).startswith(os.path.join(REFERENCE, "src") + os.sep)
    for name in ("init_frame_interpolation_model", "batch_images_interpolation_tool"):   # the `-acc` plumbing is shimmed too
        assert ns[name].__module__ == "aniportrait_amd.frame_interpolation", (name, ns[name].__module__)
    assert issubclass(ns["Pose2VideoPipeline"], diffusers.DiffusionPipeline)
    # the reader control the pipelines attach comes from the same shim; context scheduler too
    from src.models.mutual_self_attention import ReferenceAttentionControl
    from src.pipelines.context import get_context_scheduler
    assert ReferenceAttentionControl.__module__ == "aniportrait_amd.mutual_self_attention"
    assert get_context_scheduler.__module__ == "aniportrait_amd.context"
    print("OK", script, sorted(k for k in ns if not k.startswith("__"))[:5], "...")


def _inputs(case):
    from golden_inputs import pipe_inputs
    import numpy as np
    i = pipe_inputs(case)
    return i, np.array(list(i["poses"]))


def check_script_main(tmp, device, weight_dtype):
    """scripts/pose2vid.py:50-110,166-176 from an on-disk `pretrained_model/` tree -> video; vs the fixture the
    reference's own pipeline produced from the same weights / seed (fp32), or vs the CPU oracle on the latents the
    fp16 run draws (the reference draws them in the CLIP tower's dtype, pipeline_pose2vid_long.py:415)."""
    _paths(False)
    import torch
    from dropin_driver import install_stub_modules, script_main, write_pretrained_tree
    install_stub_modules()
    if device == "cpu":
        import emu_hipops

        class _P:
            def setattr(self, o, n, v):
                setattr(o, n, v)
        emu_hipops.install(_P())
    from util import load_golden, psnr
    cfg_path, sds = write_pretrained_tree(tmp, small=True)
    if weight_dtype == "fp32":
        import yaml
        with open(cfg_path) as f:
            c = yaml.safe_load(f)
        c["weight_dtype"] = "fp32"
        with open(cfg_path, "w") as f:
            yaml.safe_dump(c, f)
    i, pose_list = _inputs("long_L4")
    seen = {}

    def observe(name, obj):
        if name == "reference_unet_base":
            sd = obj.state_dict()
            assert not any(k.startswith("conv_out") for k in sd) and not obj.training
            k = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight"
            assert torch.equal(sd[k].float().cpu(), sds["base"][k].half().float())
        elif name == "denoising_unet_base":
            sd = obj.state_dict()
            k2, k3 = "conv_out.weight", "down_blocks.0.motion_modules.0.temporal_transformer.proj_out.weight"
            assert torch.equal(sd[k2].float().cpu(), sds["base"][k2].half().float())
            assert torch.equal(sd[k3].float().cpu(), sds["mm"][k3].half().float())
        elif name == "pipe":
            import diffusers
            assert isinstance(obj, diffusers.DiffusionPipeline)
            assert str(obj.device).startswith(device)
            seen["pipe"] = obj

    video = script_main(cfg_path, i["W"], i["H"], i["L"], i["steps"], i["cfg"], 42, i["ref_image"], pose_list,
                        i["ref_pose"], device=device, pose_channels=64, observe=observe)
    assert tuple(video.shape) == (1, 3, i["L"], i["H"], i["W"]) and video.dtype == torch.float32
    assert video.device.type == "cpu" and float(video.min()) >= 0 and float(video.max()) <= 1
    if weight_dtype == "fp32":
        ref = load_golden("small_pipeline.pt")["long_L4/video_f16"].float()
    else:
        from aniportrait_amd import configs as C
        from aniportrait_amd.image_processor import randn_tensor
        from oracle import ref_torch as O
        pipe = seen["pipe"]
        lat = randn_tensor((1, 4, i["L"], i["H"] // 8, i["W"] // 8), generator=torch.manual_seed(42),
                           device=torch.device("cpu"), dtype=torch.float16).float()
        clip = pipe._clip_embeds(i["ref_image"], torch.device("cpu")).float()
        sd4 = {k: sds[k] for k in ("denoising_unet", "reference_unet", "vae", "pose_guider")}
        ref = O.pose2vid(sd4, {"unet": C.unet3d_kwargs(True), "vae": C.SD_VAE_SMALL}, clip, i["ref_image"],
                         list(pose_list), i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"], lat, long=True)
    p = psnr(video, ref)
    # the script's tail (scripts/pose2vid.py:176-196): [reference image | poses | result] side by side through the shimmed
    # src.utils.util.save_videos_grid (a .gif here: PyAV is not installed)
    import numpy as np
    from PIL import Image
    from src.utils.util import save_videos_grid
    assert save_videos_grid.__module__ == "aniportrait_amd.video_io"
    to_t = lambda im: torch.from_numpy(np.asarray(im.resize((i["W"], i["H"])), dtype=np.float32) / 255).permute(2, 0, 1)  # noqa: E731
    ref_t = to_t(i["ref_image"])[None, :, None].repeat(1, 1, video.shape[2], 1, 1)
    pose_t = torch.stack([to_t(Image.fromarray(a)) for a in pose_list], 0).transpose(0, 1)[None]
    out = os.path.join(tmp, "out", "clip.gif")
    save_videos_grid(torch.cat([ref_t, pose_t[:, :, :video.shape[2]], video], dim=0), out, n_rows=3, fps=8)
    gif = Image.open(out)
    assert gif.n_frames == i["L"] and gif.size == (3 * (i["W"] + 2) + 2, i["H"] + 4), (gif.n_frames, gif.size)
    print(f"OK script_main device={device} dtype={weight_dtype} PSNR={p:.2f} dB")
    assert p >= 40.0, p


def check_reference_recipe():
    """re-run a slice of oracle/make_golden.py — AFTER importing the repository's own `src` shim, the hostile case —
    and compare with the committed fixtures: the recipe must still execute the reference's files (not the product's)
    and reproduce what is in tests/golden/."""
    _paths(False)
    import json
    import torch
    import src.models.unet_3d as shim  # noqa: F401  the shim is already imported when the harness is set up
    assert os.path.abspath(shim.__file__).startswith(REPO + os.sep)
    from oracle import make_golden as G
    from oracle import ref_harness as R
    R.setup()
    import src.models.unet_3d as ref_unet
    import src.pipelines.context as ref_ctx
    import src.pipelines.pipeline_pose2vid as ref_pipe
    for m in (ref_unet, ref_ctx, ref_pipe):
        assert os.path.abspath(m.__file__).startswith(REFERENCE + os.sep), m.__file__
    with open(os.path.join(G.GOLD, "context_windows.json")) as f:
        win = json.load(f)
    for L in (4, 16, 17, 24, 46, 150):
        assert win[str(L)] == [list(map(int, w)) for w in ref_ctx.uniform(0, 25, L, 16, 1, 4)], L
    models = R.build_models(small=True)
    assert type(models["denoising_unet"]).__module__ == "src.models.unet_3d"
    assert os.path.abspath(sys.modules["src.models.unet_3d"].__file__).startswith(REFERENCE + os.sep)
    with open(os.path.join(G.GOLD, "shapes_small.json")) as f:
        assert json.load(f) == G.manifest(models)
    got = G.pipeline_level(models, names=["short_L4"])
    gold = torch.load(os.path.join(G.GOLD, "small_pipeline.pt"), map_location="cpu")
    for k, v in got.items():
        err = (v.float() - gold[k].float()).abs().max().item()
        assert err <= 2e-3, (k, err)   # fp16-rounded frames: at most one ulp of [0, 1] apart across thread counts
    print("OK reference recipe reproduces", sorted(got))


if __name__ == "__main__":
    globals()["check_" + sys.argv[1]](*sys.argv[2:])
