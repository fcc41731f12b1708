"""End-to-end parity of the public pipeline at the REAL SD-1.5 / sd-vae-ft-mse widths (head dims 40 / 80 / 160) on
BASELINE.json's own configurations, against the CPU oracle (oracle/ref_torch.py, fp32) on the GPU box's host cores —
the north star's criterion: PSNR >= 40 dB on the decoded frames for identical weights, latents and inputs
(reference loop: src/pipelines/pipeline_pose2vid_long.py:458-559).

  C1 in full          256x256, L=4, 10 DDIM steps, CFG 3.5                       (BASELINE configs[0])
  windowed            192x192, L=24 -> two 16-frame windows, 2 steps              (the C4 mechanism at real width)
  C2, reduced steps   512x512, L=16, 1 step + ReferenceNet + VAE                  (BASELINE configs[1]; `slow`: the CPU
                      oracle needs ~2.5 min per UNet3D call on the 32-frame CFG batch at this size, so the oracle decodes
                      2 of the 16 frames; round 2's first GPU run did 2 steps x 16 frames: 46.4 dB, oracle 425 s)
  reference fixtures  tests/golden/real_pipeline_<case>.pt — outputs of the REFERENCE's own pipeline run on PyTorch-CPU fp32 in
                      the build container (oracle/make_golden_real_pipeline.py): C2 geometry at 4 DDIM steps, all 16 frames;
                      C5 geometry (768x768, 96x96 latents) at 1, 4 and all 25 steps; C2 in full (25 steps); L=40 at real width (4 windows
                      per step incl. the wrap-around one — the same windows at every step: the reference passes step 0 to its
                      context scheduler); C4 at its own geometry (512x512, L=150: 13 windows) at 1 step.  Only compared here —
                      no CPU oracle run on the GPU box.
  graph reuse         clip A, a different clip B (other image / poses / latents / resolution), clip A again through the
                      SAME pipeline object with the captured hipGraph active: every clip matches the oracle and A is
                      bit-identical before and after B (in-place bank / attn2 refresh, runner cache)
"""
import time

import pytest
import torch

from util import build_hip_models_cached, clip_encoder_for, oracle_threads, psnr

pytestmark = pytest.mark.gpu
PSNR_BAR = 40.0


def _pipe(small, long=True):
    from aniportrait_amd import configs as C
    from aniportrait_amd.scheduling_ddim import DDIMScheduler
    from src.pipelines.pipeline_pose2vid import Pose2VideoPipeline as ShortPipe
    from src.pipelines.pipeline_pose2vid_long import Pose2VideoPipeline as LongPipe
    m, sds = build_hip_models_cached(small)
    pipe = (LongPipe if long else ShortPipe)(
        vae=m["vae"], image_encoder=clip_encoder_for(small), reference_unet=m["reference_unet"],
        denoising_unet=m["denoising_unet"], pose_guider=m["pose_guider"], scheduler=DDIMScheduler(**C.DDIM_V2))
    pipe.set_progress_bar_config(disable=True)
    return pipe, sds


@pytest.fixture(scope="module")
def real_pipe():
    return _pipe(False)


def _clip(H, W, L, seed):
    from aniportrait_amd.synthetic import synth_latents, synth_pose_frames, synth_ref_image
    return dict(H=H, W=W, L=L, ref_image=synth_ref_image(H, W, 1 + seed), poses=list(synth_pose_frames(L, H, W, 1234 + 100 * seed)),
                ref_pose=synth_pose_frames(1, H, W, 999)[0], latents=synth_latents(L, H // 8, W // 8, 42 + seed))


def _run(pipe, c, steps, cfg=3.5, **kw):
    return pipe(c["ref_image"], list(c["poses"]), c["ref_pose"], c["W"], c["H"], c["L"], steps, cfg,
                latents=c["latents"], **kw).videos


def _oracle(pipe, sds, small, c, steps, cfg=3.5, decode_frames=None, **kw):
    """decode_frames: the oracle decodes only these frame indices (its VAE costs 2.5 TFLOP per 512x512 frame)"""
    from aniportrait_amd import configs as C
    from oracle import ref_torch as O
    n = oracle_threads()
    clip = pipe._clip_embeds(c["ref_image"], torch.device("cpu")).float()
    cfgs = {"unet": C.unet3d_kwargs(small), "vae": C.SD_VAE_SMALL if small else C.SD_VAE_FT_MSE}
    t0 = time.time()
    ref = O.pose2vid(sds, cfgs, clip, c["ref_image"], list(c["poses"]), c["ref_pose"], c["W"], c["H"], c["L"], steps, cfg,
                     c["latents"], long=True, return_latents=decode_frames is not None, **kw)
    if decode_frames is not None:
        ref = O.decode_latents(sds["vae"], cfgs["vae"], ref[:, :, list(decode_frames)])
    return ref, time.time() - t0, n


@torch.no_grad()
def test_c1_in_full_and_graph_reuse_at_real_width(real_pipe):
    """BASELINE configs[0] in full; then another clip shape through the same pipeline; then C1 again: bit-identical"""
    pipe, sds = real_pipe
    a = _clip(256, 256, 4, 0)
    vid = _run(pipe, a, 10)
    ref, secs, n = _oracle(pipe, sds, False, a, 10)
    p = psnr(vid, ref)
    print(f"C1 256x256 L=4 10 steps CFG 3.5, real width: PSNR vs CPU oracle = {p:.2f} dB (oracle {secs:.0f} s on {n} threads)")
    assert vid.shape == ref.shape == (1, 3, 4, 256, 256) and p >= PSNR_BAR
    # a different clip in between: other image / poses / latents AND another latent size (new bank shapes, new runner)
    b = _clip(128, 192, 4, 7)
    vb = _run(pipe, b, 2)
    rb, _, _ = _oracle(pipe, sds, False, b, 2)
    pb = psnr(vb, rb)
    print(f"clip B 128x192 L=4 2 steps: PSNR = {pb:.2f} dB")
    assert pb >= PSNR_BAR
    again = _run(pipe, a, 10)
    assert torch.equal(again, vid), f"clip A differs after clip B went through the cached graphs: {psnr(again, vid):.1f} dB"


@torch.no_grad()
def test_windowed_long_clip_at_real_width(real_pipe):
    """two overlapping 16-frame context windows per step (the C4 mechanism) at real width"""
    pipe, sds = real_pipe
    c = _clip(192, 192, 24, 1)
    vid = _run(pipe, c, 2)
    ref, secs, n = _oracle(pipe, sds, False, c, 2)
    p = psnr(vid, ref)
    print(f"windowed 192x192 L=24 (2 windows) 2 steps: PSNR = {p:.2f} dB (oracle {secs:.0f} s on {n} threads)")
    assert p >= PSNR_BAR


def _fixture_case(pipe, name):
    """run the public pipeline on the seeded inputs of a REAL_PIPE_CASES entry; returns (video, per-step latents,
    fixture dict)"""
    import os

    from golden_inputs import real_pipe_inputs
    from util import GOLD
    path = os.path.join(GOLD, f"real_pipeline_{name}.pt")
    if not os.path.isfile(path):
        pytest.skip(f"{path} not generated (oracle/make_golden_real_pipeline.py {name})")
    gold = torch.load(path, map_location="cpu")
    i = real_pipe_inputs(name)
    lats = []
    vid = pipe(i["ref_image"], list(i["poses"]), i["ref_pose"], i["W"], i["H"], i["L"], i["steps"], i["cfg"],
               generator=torch.manual_seed(i["gen_seed"]), callback=lambda k, t, lat: lats.append(lat.float().cpu())).videos
    return vid, lats, gold, i


def _report(name, vid, lats, gold, i):
    fr = [int(k) for k in gold["frames"]]
    want = gold["frames_u8"].float().permute(3, 0, 1, 2)[None] / 255.0           # (1, 3, n, H, W)
    per_frame = [psnr(vid[:, :, k], want[:, :, j]) for j, k in enumerate(fr)]
    p = psnr(vid[:, :, fr], want)
    lat_db = []
    for s_, lat in enumerate(lats):
        ref = gold["latents_f16"][s_].float()
        mse = float(((lat.double() - ref.double()) ** 2).mean())
        lat_db.append(10 * __import__("math").log10(float(ref.double().pow(2).mean()) / max(mse, 1e-30)))
    print(f"{name}: {i['H']}x{i['W']} L={i['L']} {i['steps']} steps vs the REFERENCE's own CPU pipeline: PSNR over "
          f"{len(fr)} decoded frames = {p:.2f} dB (worst frame {min(per_frame):.2f} dB); latent SNR per step "
          f"{[round(x, 1) for x in lat_db]} dB")
    return p, min(per_frame), lat_db


@torch.no_grad()
def test_c2_four_steps_all_frames_vs_reference_fixture(real_pipe):
    """BASELINE configs[1] geometry (512x512, L=16, CFG 3.5; 32-frame CFG batch, 64x64 latents, reference attention over
    8192 keys) at 4 DDIM steps: ALL 16 decoded frames against the frames the reference's own pipeline produced on
    PyTorch-CPU fp32 (display bytes; quantisation floor 58.9 dB), bar 40 dB on the whole and on every single frame"""
    pipe, _ = real_pipe
    vid, lats, gold, i = _fixture_case(pipe, "c2_4step")
    p, worst, lat_db = _report("C2", vid, lats, gold, i)
    assert vid.shape == (1, 3, 16, 512, 512) and len(gold["frames"]) == 16
    assert p >= PSNR_BAR and worst >= PSNR_BAR
    assert abs(float(vid.double().mean()) - float(gold["video_mean"])) < 2e-3


@torch.no_grad()
def test_c5_768_one_step_vs_reference_fixture(real_pipe):
    """BASELINE configs[4] geometry (768x768, L=16: 96x96 latents, T = 9216 tokens, 18 432 keys per conditional frame)
    at 1 DDIM step against the reference's own CPU pipeline (4 stored frames)"""
    pipe, _ = real_pipe
    vid, lats, gold, i = _fixture_case(pipe, "c5_1step")
    p, worst, _ = _report("C5", vid, lats, gold, i)
    assert vid.shape == (1, 3, 16, 768, 768)
    assert p >= PSNR_BAR and worst >= PSNR_BAR


@torch.no_grad()
def test_c2_full_25_step_schedule_vs_reference_fixture(real_pipe):
    """BASELINE configs[1] IN FULL — 512x512, L=16, CFG 3.5, the whole 25-step trailing / zero-SNR / v-prediction DDIM schedule
    that bench.py times — against the reference's own pipeline on PyTorch-CPU fp32 (72 CPU-minutes in the build container:
    oracle/make_golden_real_pipeline.py c2_25step): all 16 decoded frames, whole and worst frame >= 40 dB, and the latent
    SNR after every one of the 25 steps (printed: the fp16 error accumulates over the schedule)"""
    pipe, _ = real_pipe
    vid, lats, gold, i = _fixture_case(pipe, "c2_25step")
    p, worst, lat_db = _report("C2 25 steps", vid, lats, gold, i)
    assert vid.shape == (1, 3, 16, 512, 512) and len(gold["frames"]) == 16 and len(lats) == 25
    assert p >= PSNR_BAR and worst >= PSNR_BAR and min(lat_db) >= 40.0
    assert abs(float(vid.double().mean()) - float(gold["video_mean"])) < 2e-3


@torch.no_grad()
def test_c5_768_four_steps_vs_reference_fixture(real_pipe):
    """BASELINE configs[4] geometry (768x768, L=16) at 4 DDIM steps against the reference's own CPU pipeline (4 stored frames)"""
    pipe, _ = real_pipe
    vid, lats, gold, i = _fixture_case(pipe, "c5_4step")
    p, worst, lat_db = _report("C5 4 steps", vid, lats, gold, i)
    assert vid.shape == (1, 3, 16, 768, 768) and len(lats) == 4
    assert p >= PSNR_BAR and worst >= PSNR_BAR and min(lat_db) >= 40.0


@torch.no_grad()
def test_c5_768_full_25_step_schedule_vs_reference_fixture(real_pipe):
    """BASELINE configs[4] geometry (768x768, L=16, CFG 3.5: 96x96 latents, 18 432 keys per conditional frame) with its WHOLE
    25-step schedule against the reference's own pipeline on PyTorch-CPU fp32 (oracle/make_golden_real_pipeline.py c5_25step,
    hours of CPU in the build container): 4 stored frames, the latent SNR after every one of the 25 steps"""
    pipe, _ = real_pipe
    vid, lats, gold, i = _fixture_case(pipe, "c5_25step")
    p, worst, lat_db = _report("C5 25 steps", vid, lats, gold, i)
    assert vid.shape == (1, 3, 16, 768, 768) and len(lats) == 25
    assert p >= PSNR_BAR and worst >= PSNR_BAR and min(lat_db) >= 40.0
    assert abs(float(vid.double().mean()) - float(gold["video_mean"])) < 2e-3


@torch.no_grad()
def test_l40_wraparound_windows_vs_reference_fixture(real_pipe):
    """real width, L=40, 3 steps: 4 overlapping 16-frame windows per step, the last one wrapping around the clip end
    ([36..39, 0..11]); the reference passes step 0 to the context scheduler at every DDIM step
    (pipeline_pose2vid_long.py:488-501), so the windows are the same at every step — the C4 mechanism"""
    pipe, _ = real_pipe
    vid, lats, gold, i = _fixture_case(pipe, "l40_windows")
    p, worst, lat_db = _report("L40", vid, lats, gold, i)
    assert vid.shape == (1, 3, 40, 128, 128)
    assert p >= PSNR_BAR and worst >= PSNR_BAR and min(lat_db) >= 40.0


@torch.no_grad()
def test_c4_long_clip_150_frames_one_step_vs_reference_fixture(real_pipe):
    """BASELINE configs[3] at its OWN geometry — 512x512, L=150: 13 overlapping 16-frame context windows per step, the last one
    wrapping around the clip end ([144..149, 0..9]), window sums and counters merged over 150 frames
    (src/pipelines/pipeline_pose2vid_long.py:487-555) — at 1 DDIM step against the reference's own pipeline on PyTorch-CPU fp32
    (oracle/make_golden_real_pipeline.py c4_1step): the latents of ALL 150 frames after the step, and six decoded frames out of
    plain, overlapping and wrap-around windows"""
    pipe, _ = real_pipe
    vid, lats, gold, i = _fixture_case(pipe, "c4_1step")
    p, worst, lat_db = _report("C4 1 step", vid, lats, gold, i)
    assert vid.shape == (1, 3, 150, 512, 512) and len(lats) == 1 and tuple(lats[0].shape) == (1, 4, 150, 64, 64)
    assert p >= PSNR_BAR and worst >= PSNR_BAR and min(lat_db) >= 40.0
    # per-frame latent SNR: every frame of the clip (each lies in one or two windows), not only the decoded ones
    ref = gold["latents_f16"][0].float()
    err = ((lats[0].double() - ref.double()) ** 2).mean(dim=(0, 1, 3, 4))
    sig = ref.double().pow(2).mean(dim=(0, 1, 3, 4))
    snr = 10 * torch.log10(sig / err.clamp_min(1e-30))
    print(f"C4 latent SNR per frame: min {float(snr.min()):.1f} dB (frame {int(snr.argmin())}), median {float(snr.median()):.1f} dB")
    assert float(snr.min()) >= 40.0


@torch.no_grad()
def test_c4_long_clip_150_frames_four_steps_vs_reference_fixture(real_pipe):
    """BASELINE configs[3] at its own geometry over a MULTI-step schedule (round 6): 512x512, L = 150, 13
    windows per step incl. the wrap-around one, a 4-step DDIM schedule — the merged window sums of step k are the latents every
    window of step k + 1 starts from, so an error in the overlap / wrap-around accumulate compounds instead of showing once.
    Against the reference's own pipeline on PyTorch-CPU fp32 (oracle/make_golden_real_pipeline.py c4_4step, ~40 CPU-minutes per
    step): the latents of all 150 frames after every step the fixture holds (overall and per frame >= 40 dB), and — when the
    recipe ran to its end — six decoded frames (PSNR >= 40 dB).  (A fixture may hold fewer steps: the build container's sessions
    were interrupted repeatedly, the recipe checkpoints every step and `--finish`es from there; the committed one is complete.)"""
    pipe, _ = real_pipe
    vid, lats, gold, i = _fixture_case(pipe, "c4_4step")
    n_ref = int(gold["latents_f16"].shape[0])
    assert vid.shape == (1, 3, 150, 512, 512) and len(lats) == 4 and tuple(lats[0].shape) == (1, 4, 150, 64, 64) and n_ref >= 3
    if "frames_u8" in gold:
        p, worst, lat_db = _report("C4 4 steps", vid, lats, gold, i)
        assert p >= PSNR_BAR and worst >= PSNR_BAR and min(lat_db) >= 40.0
    for s_ in range(n_ref):
        lat, ref = lats[s_], gold["latents_f16"][s_].float()
        err = ((lat.double() - ref.double()) ** 2).mean(dim=(0, 1, 3, 4))
        sig = ref.double().pow(2).mean(dim=(0, 1, 3, 4))
        snr = 10 * torch.log10(sig / err.clamp_min(1e-30))
        tot = 10 * torch.log10(sig.sum() / err.sum().clamp_min(1e-30))
        print(f"C4 step {s_ + 1} of 4 ({n_ref} in the fixture): latent SNR all frames {float(tot):.1f} dB, per frame min {float(snr.min()):.1f} dB "
              f"(frame {int(snr.argmin())}), median {float(snr.median()):.1f} dB")
        assert float(tot) >= 40.0 and float(snr.min()) >= 40.0, (s_, float(tot), float(snr.min()))


@pytest.mark.slow
@pytest.mark.skipif(not __import__("os").environ.get("ANIP_INLINE_ORACLE"),
                    reason="superseded by test_c2_four_steps_all_frames_vs_reference_fixture (no 3-minute CPU oracle run on "
                           "the GPU box); ANIP_INLINE_ORACLE=1 runs it")
@torch.no_grad()
def test_c2_reduced_steps_at_real_width(real_pipe):
    """BASELINE configs[1] geometry (512x512, L=16, CFG 3.5) at 1 DDIM step: VAE encode + ReferenceNet + PoseGuider on 16
    frames + one UNet3D call on the 32-frame CFG batch (64x64 latents: T = 4096 tokens, head dims 40 / 80 / 160, reference
    attention over 8192 keys) + VAE decode, everything at the sizes the bench runs; frames 0 and 9 against the oracle"""
    pipe, sds = real_pipe
    c = _clip(512, 512, 16, 2)
    vid = _run(pipe, c, 1)
    frames = (0, 9)
    ref, secs, n = _oracle(pipe, sds, False, c, 1, decode_frames=frames)
    p = psnr(vid[:, :, list(frames)], ref)
    print(f"C2 512x512 L=16 1 step, frames {frames}: PSNR = {p:.2f} dB (oracle {secs:.0f} s on {n} threads)")
    assert vid.shape == (1, 3, 16, 512, 512) and p >= PSNR_BAR


@torch.no_grad()
def test_two_different_clips_through_one_graph_small_width():
    """small width (cheap oracle): A -> B (same shapes, different content) -> A' (other resolution) -> A, all through
    one pipeline object with the hipGraph active; every clip vs the oracle, A bit-identical each time"""
    pipe, sds = _pipe(True)
    a, b, c = _clip(128, 128, 4, 0), _clip(128, 128, 4, 5), _clip(192, 128, 4, 6)
    outs = []
    for clip_ in (a, b, c, a, b):
        outs.append(_run(pipe, clip_, 3))
    for clip_, got in zip((a, b, c), outs[:3]):
        ref, _, _ = _oracle(pipe, sds, True, clip_, 3)
        p = psnr(got, ref)
        print(f"small-width clip {clip_['H']}x{clip_['W']}: PSNR = {p:.2f} dB")
        assert p >= PSNR_BAR
    assert torch.equal(outs[3], outs[0]) and torch.equal(outs[4], outs[1])
    assert not torch.equal(outs[0], outs[1])
    # no-CFG clip in between (batch 1: other attn2 / bank buffers), then A again
    _run(pipe, a, 2, cfg=1.0)
    assert torch.equal(_run(pipe, a, 3), outs[0])
    # output options: display bytes made on the device, and the frames draining asynchronously through pinned memory
    u8 = pipe(a["ref_image"], list(a["poses"]), a["ref_pose"], a["W"], a["H"], a["L"], 3, 3.5, latents=a["latents"],
              output_type="uint8").videos
    assert (u8.numpy() == (outs[0][0].permute(1, 2, 3, 0) * 255).numpy().astype("uint8")).all()
    pend = [pipe(c_["ref_image"], list(c_["poses"]), c_["ref_pose"], c_["W"], c_["H"], c_["L"], 3, 3.5,
                 latents=c_["latents"], async_output=True).videos for c_ in (a, b, a)]   # three clips in flight
    assert torch.equal(pend[0].result(), outs[0]) and torch.equal(pend[1].result(), outs[1])
    assert torch.equal(pend[2].result(), outs[0])
    # the runner cache is bounded
    pipe.max_cached_graphs = 1
    _run(pipe, c, 2)
    assert len(pipe._get_runners()) == 1
    pipe.drop_cached_graphs()
    assert len(pipe._get_runners()) == 0
