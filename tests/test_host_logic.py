"""Host-side pieces of the path against the oracle / the reference-made goldens (CPU)."""
import json
import os

import numpy as np
import pytest
import torch

from util import GOLD, load_golden, oracle_state_dicts, rel_err


def test_context_windows_match_reference_goldens():
    from aniportrait_amd.context import get_context_scheduler, uniform
    with open(os.path.join(GOLD, "context_windows.json")) as f:
        win = json.load(f)
    for L in (4, 16, 17, 24, 46, 150):
        assert [list(w) for w in uniform(0, 25, L, 16, 1, 4)] == win[str(L)]
    assert [list(w) for w in get_context_scheduler("uniform")(0, 2, 10, 8, 1, 2)] == win["10_ctx8_ov2"]
    with pytest.raises(ValueError):
        get_context_scheduler("nope")
    # every frame of a long clip is covered at least once, windows are 16 long and wrap (closed loop)
    cover = np.zeros(150, int)
    for w in uniform(0, 25, 150, 16, 1, 4):
        assert len(w) == 16 and len(set(w)) == 16
        cover[w] += 1
    assert cover.min() >= 1 and cover.max() <= 2


def test_context_windows_match_oracle_for_other_steps():
    from aniportrait_amd.context import ordered_halving, uniform
    from oracle import ref_torch as O
    for step in (0, 1, 2, 5, 24):
        assert ordered_halving(step) == O._ordered_halving(step)
        for L in (20, 33, 150):
            for stride in (1, 2, 3):
                assert [list(w) for w in uniform(step, 25, L, 16, stride, 4)] == \
                    O.uniform_windows(step, L, 16, stride, 4)


def test_ddim_scheduler_matches_oracle():
    from aniportrait_amd import configs as C
    from aniportrait_amd.scheduling_ddim import DDIMScheduler
    from oracle import ref_torch as O
    s = DDIMScheduler(**C.DDIM_V2)
    assert torch.equal(s.alphas_cumprod, O.ddim_tables()) and float(s.alphas_cumprod[-1]) == 0.0
    for steps in (2, 10, 25):
        s.set_timesteps(steps)
        assert s.timesteps.tolist() == O.ddim_timesteps(steps)
    s.set_timesteps(25)
    assert s.timesteps[0] == 999 and s.timesteps[-1] == 39 and s.init_noise_sigma == 1.0 and s.order == 1
    g = torch.Generator().manual_seed(0)
    x, v = torch.randn((1, 4, 3, 8, 8), generator=g), torch.randn((1, 4, 3, 8, 8), generator=g)
    acp = O.ddim_tables()
    for t in (999, 519, 39):
        want = O.ddim_step_v(acp, t, 25, v, x)
        got = s.step(v, t, x).prev_sample
        assert torch.allclose(got, want, atol=1e-6)
        sa, sb, sap, sbp = s.coefficients(t)
        fused = sap * (sa * x - sb * v) + sbp * (sa * v + sb * x)   # what anip_cfg_ddim_step computes
        assert torch.allclose(fused, want, atol=1e-6)
    assert s.scale_model_input(x, 5) is x
    with pytest.raises(NotImplementedError):
        DDIMScheduler(**dict(C.DDIM_V2, clip_sample=True)).coefficients(999)
    eps0 = DDIMScheduler(**dict(C.DDIM_V2, prediction_type="epsilon"))
    eps0.set_timesteps(25)
    with pytest.raises(ValueError):            # epsilon prediction at alpha_bar = 0 divides by zero: zero-SNR needs v-prediction
        eps0.coefficients(999)


def test_ddim_scheduler_inference_v1_form():
    """configs/inference/inference_v1.yaml:18-23 — epsilon prediction, leading spacing with steps_offset 1, no zero-SNR rescale:
    timesteps as the reference's scheduler produced them (golden), and the fused kernel's four-scalar form reproduces step()
    for epsilon and sample prediction"""
    from aniportrait_amd import configs as C
    from aniportrait_amd.scheduling_ddim import DDIMScheduler
    from util import load_golden
    s = DDIMScheduler(**C.DDIM_V1)
    assert s.config.prediction_type == "epsilon" and s.config.timestep_spacing == "leading" and not s.config.rescale_betas_zero_snr
    s.set_timesteps(3)
    assert s.timesteps.tolist() == load_golden("small_pipeline_v1.pt")["timesteps_3"].tolist() == [667, 334, 1]
    g = torch.Generator().manual_seed(1)
    x, m = torch.randn((1, 4, 3, 8, 8), generator=g), torch.randn((1, 4, 3, 8, 8), generator=g)
    for kind in ("epsilon", "sample", "v_prediction"):
        s = DDIMScheduler(**dict(C.DDIM_V1, prediction_type=kind))
        s.set_timesteps(25)
        for t in s.timesteps.tolist()[::6] + [int(s.timesteps[-1])]:
            want = s.step(m, t, x).prev_sample
            sa, sb, sap, sbp = s.coefficients(t)
            fused = sap * (sa * x - sb * m) + sbp * (sa * m + sb * x)   # what anip_cfg_ddim_step computes
            assert torch.allclose(fused, want, atol=2e-5, rtol=1e-5), (kind, t, float((fused - want).abs().max()))


def test_image_processor_matches_oracle():
    from aniportrait_amd.image_processor import VaeImageProcessor, randn_tensor
    from aniportrait_amd.synthetic import synth_pose_frames, synth_ref_image
    from oracle import ref_torch as O
    ip = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True)
    img = synth_ref_image(100, 132)
    assert torch.equal(ip.preprocess(img, height=64, width=72), O.preprocess_pil(img, 64, 72))
    pose = synth_pose_frames(2, 64, 64)
    got = ip.preprocess(pose[0], height=64, width=64)
    assert torch.equal(got, O.preprocess_np(pose[0], 64, 64))
    assert got.max() > 1.0  # the numpy path is NOT divided by 255 (reference quirk)
    a = randn_tensor((1, 4, 2, 8, 8), generator=torch.Generator().manual_seed(42), device="cpu", dtype=torch.float32)
    b = torch.randn((1, 4, 2, 8, 8), generator=torch.Generator().manual_seed(42))
    assert torch.equal(a, b)


def test_pose_guider_matches_reference_golden():
    """adjacent torch module, fp32 on CPU, against features produced by the reference's own PoseGuider"""
    from aniportrait_amd.pose_guider import PoseGuider
    from golden_inputs import unet_case
    gold = load_golden("small_models.pt")
    pg = PoseGuider(noise_latent_channels=64)
    missing, unexpected = pg.load_state_dict(oracle_state_dicts(True, keys=["pose_guider"])["pose_guider"], strict=False)
    assert not unexpected
    assert pg.training  # the scripts never call .eval(): BatchNorm uses batch statistics
    c = unet_case(True)
    fea = pg(c["pose"], c["ref_pose"])
    for i, f in enumerate(fea):
        assert rel_err(f, gold[f"pose_fea/{i}"]) < 2e-4


def test_bank_shapes_and_pipeline_refuses_cpu():
    from aniportrait_amd import configs as C
    from aniportrait_amd.pipeline_pose2vid_long import bank_shapes
    sh = bank_shapes(C.unet3d_kwargs(False), 2, 64, 64)
    assert len(sh) == 16 and sh["down_blocks.0.attentions.0"] == (2, 4096, 320)
    assert sh["mid_block.attentions.0"] == (2, 64, 1280) and sh["up_blocks.3.attentions.2"] == (2, 4096, 320)
    assert sum(np.prod(s) for s in sh.values()) * 2 / 1e6 == pytest.approx(46.2, abs=0.3)  # 46 MB


def test_reference_attention_control_protocol():
    """write -> update -> clear hand-off on CPU tensors (no compute)"""
    from aniportrait_amd import configs as C
    from aniportrait_amd.mutual_self_attention import ReferenceAttentionControl as RAC
    from aniportrait_amd.unet import UNet2DConditionModel, UNet3DConditionModel
    u3 = UNet3DConditionModel(**C.unet3d_kwargs(True))
    u2 = UNet2DConditionModel(**C.unet2d_kwargs(True))
    w = RAC(u2, mode="write", do_classifier_free_guidance=True, fusion_blocks="full")
    r = RAC(u3, mode="read", do_classifier_free_guidance=True, fusion_blocks="full")
    assert all(rb.state.mode == "write" for rb in u2._ref_blocks.values())
    assert all(rb.state.mode == "read" for rb in u3._ref_blocks.values()) and u3._ref_cfg
    for i, (p, rb) in enumerate(u2._ref_blocks.items()):
        rb.node.bank.append(torch.full((2, 4, 8), float(i) + 0.123456))
    r.update(w)
    for p in u3._ref_blocks:
        got, src = u3._ref_blocks[p].node.bank[0], u2._ref_blocks[p].node.bank[0]
        assert got.dtype == torch.float16 and torch.equal(got, src.half())   # rounded through fp16 (:302,338)
        assert got.data_ptr() != src.data_ptr()
    r.clear()
    w.clear()
    assert all(len(rb.node.bank) == 0 for rb in list(u3._ref_blocks.values()) + list(u2._ref_blocks.values()))
    m = RAC(u3, mode="read", fusion_blocks="midup")
    modes = {p: rb.state.mode for p, rb in u3._ref_blocks.items()}
    assert modes["down_blocks.0.attentions.0"] == "plain" and modes["mid_block.attentions.0"] == "read"
    with pytest.raises(AssertionError):
        RAC(u3, mode="both")


def test_wrapped_dilated_window_accumulates_like_the_reference(monkeypatch):
    """context_stride > 1 can make a window visit a frame twice; the accumulation must equal the reference's index
    assignment (last occurrence wins, frame counted once) — host-side masking + the (emulated) kernel's skip rule"""
    import emu_hipops
    from aniportrait_amd.context import uniform
    from aniportrait_amd.pipeline_pose2vid_long import _last_occurrence_only
    L, S, HWC = 20, 2, 12
    windows = [list(w) for w in uniform(0, 25, L, 16, 2, 4)]
    assert any(len(set(w)) < len(w) for w in windows)           # the case exists with the reference's scheduler
    g = torch.Generator().manual_seed(3)
    acc, cnt = torch.zeros(S, L, HWC), torch.zeros(L)
    ref_acc, ref_cnt = torch.zeros(S, L, HWC), torch.zeros(L)
    for w in windows:
        pred = torch.randn((S, len(w), HWC), generator=g).half()
        ref_acc[:, w] = ref_acc[:, w] + pred.float()
        ref_cnt[w] = ref_cnt[w] + 1
        emu_hipops.window_accumulate(pred, acc, cnt, torch.tensor(_last_occurrence_only(w), dtype=torch.int32), S,
                                     len(w), L, HWC)
    assert torch.equal(acc, ref_acc) and torch.equal(cnt, ref_cnt)
    assert _last_occurrence_only([0, 2, 4, 0, 2]) == [-1, -1, 4, 0, 2]


def test_aux_graph_cache_is_bounded_and_follows_the_packed_weights():
    """the once-per-clip graph cache (ReferenceNet / PoseGuider): at most `max_cached_graphs` entries per kind, and an
    entry dies when its module re-packs its weights (the graph has their addresses baked in)"""
    from aniportrait_amd.pipeline_pose2vid_long import Pose2VideoPipeline

    from aniportrait_amd.engine import PackedNet

    class Mod:
        def __init__(self):
            self.p = PackedNet({}, "cpu")

        def packed(self):
            return self.p

    pipe = Pose2VideoPipeline.__new__(Pose2VideoPipeline)
    pipe.max_cached_graphs = 2
    m, other = Mod(), Mod()
    made = []

    def make(tag):
        def f():
            made.append(tag)
            return tag
        return f

    assert pipe._aux_graph("refnet", m, "a", make("A")) == "A"
    assert pipe._aux_graph("refnet", m, "a", make("A2")) == "A"          # hit
    assert pipe._aux_graph("pose", other, "a", make("P")) == "P"         # another kind: its own population
    assert pipe._aux_graph("refnet", m, "b", make("B")) == "B"
    assert pipe._aux_graph("refnet", m, "c", make("C")) == "C"           # evicts the oldest refnet entry ("a")
    assert made == ["A", "P", "B", "C"]
    assert pipe._aux_graph("refnet", m, "a", make("A3")) == "A3"
    assert pipe._aux_graph("pose", other, "a", make("P2")) == "P"        # untouched by the refnet evictions
    # weights re-packed in the hostile order (HipModel._invalidate drops the old PackedNet BEFORE packed() builds the new
    # one, so CPython may hand the new object the old one's address): the tag is the never-reused serial, not id()
    old_serial, m.p = m.p.serial, None
    m.p = PackedNet({}, "cpu")
    assert m.p.serial != old_serial
    assert pipe._aux_graph("refnet", m, "a", make("A4")) == "A4"
    assert sum(1 for k in pipe.__dict__["_aux_graphs"] if k[0] == "refnet") == 1
    pipe.drop_cached_graphs()
    assert pipe.__dict__["_aux_graphs"] == {}


def test_host_cpu_budget_follows_the_cgroup_quota(tmp_path, monkeypatch):
    """aniportrait_amd/hostcfg.py (round 6): the MI355X box's container grants 16 CPUs (`cpu.max = 1600000 100000`) under 256
    visible cores; a torch pool sized by the latter gets the process throttled for 50-90 ms at a time"""
    import torch

    from aniportrait_amd import hostcfg
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert hostcfg.cpu_quota(str(tmp_path)) == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert hostcfg.cpu_quota(str(tmp_path)) is None
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("400000")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000")
    assert hostcfg.cpu_quota(str(v1)) == 4.0
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1")
    assert hostcfg.cpu_quota(str(v1)) is None
    assert hostcfg.cpu_quota(str(tmp_path / "nowhere")) is None
    # the cap: only downwards, never past an explicit choice of the application
    before = torch.get_num_threads()
    try:
        monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
        monkeypatch.setenv("ANIP_HOST_THREADS", "0")
        assert hostcfg.bound_host_threads(force=True) == before
        monkeypatch.setenv("ANIP_HOST_THREADS", "2")
        assert hostcfg.bound_host_threads(force=True) == 2
        monkeypatch.delenv("ANIP_HOST_THREADS")
        monkeypatch.setattr(hostcfg, "usable_cpus", lambda: 4)
        torch.set_num_threads(before)
        assert hostcfg.bound_host_threads(limit=8, force=True) == min(before, 2)
    finally:
        torch.set_num_threads(before)
