"""`Pose2VideoPipeline` — the reference's pipeline call surface
(src/pipelines/pipeline_pose2vid_long.py:39-80,338-363,584; short variant in pipeline_pose2vid.py) driving the
MI355X hot path: ReferenceNet single pass -> T DDIM steps x context windows of the denoising UNet3D ->
batched VAE decode, all on HIP kernels behind `aniportrait_amd.engine`.

Differences from the reference's op sequence that leave results unchanged:
PoseGuider features are computed once per window (they do not depend on t); latents / per-frame
accumulators stay channels-last on the device for the whole loop; window accumulation, the
`noise_pred / counter` division, CFG and the DDIM v-prediction update are two fused kernels
(`anip_window_accumulate`, `anip_cfg_ddim_step`) in fp32; frames are decoded in batches.

Extra keyword arguments (all optional, reference callers never pass them):
  latents=         (1, 4, L, h, w) initial noise, injected instead of `prepare_latents` (parity tests);
  dp_group=        torch.distributed process group: shard the context windows of this clip over its ranks
                   (aniportrait_amd.distributed); every rank must make the same call;
  decode_chunk=    frames per VAE decode batch (default 16);
  output_type=     "tensor" / "numpy" as in the reference (fp32 (1,3,L,H,W) in [0,1] on the host), or "uint8":
                   display bytes (L,H,W,3) converted on the device (what save_videos_grid computes per frame on the host,
                   src/utils/util.py:97-98), a quarter of the D2H bytes;
  async_output=    True: `.videos` is a `PendingVideo`; the D2H copy runs on a side stream into pinned memory and
                   `.result()` waits for it — lets a caller start the next clip while the frames drain.
"""
import gc
import inspect
import math
import os
import time
from collections import OrderedDict

import numpy as np
import torch

from . import distributed as D
from . import engine
from . import hipops as ops
from . import hostcfg
from .autoencoder_kl import AutoencoderKL
from .context import get_context_scheduler
from .image_processor import VaeImageProcessor, randn_tensor
from .modeling import BaseOutput
from .mutual_self_attention import ReferenceAttentionControl


_GC_CONTROL = os.environ.get("ANIP_GC_CONTROL", "1") == "1"


class _StageTimer:
    """ANIP_PIPE_TIMING=1: synchronised wall time per pipeline stage, printed at the end of the call."""

    def __init__(self):
        # ANIP_PIPE_TIMING=host: the same marks WITHOUT the synchronize — host time per stage as the un-instrumented call sees it
        mode = os.environ.get("ANIP_PIPE_TIMING")
        self.on = bool(mode)
        self.sync = mode != "host"
        self.t0 = self.last = time.perf_counter()
        self.rows = []

    def mark(self, name):
        if self.on:
            if self.sync:
                torch.cuda.synchronize()
            now = time.perf_counter()
            self.rows.append((name, (now - self.last) * 1e3))
            self.last = now

    def report(self):
        if self.on:
            tot = (time.perf_counter() - self.t0) * 1e3
            print("[pipe timing] " + "  ".join(f"{n}={ms:.1f}ms" for n, ms in self.rows) + f"  total={tot:.1f}ms",
                  flush=True)


class _DenoiseRunner:
    """Static-buffer front of `UNet3DConditionModel.forward_nhwc` for one window shape (CFG batch S, f frames,
    latent h x w), kept on the pipeline across clips.  All per-call inputs live in persistent device buffers
    (x, timestep sinusoid, CLIP token, the 5 pose feature maps); the reference-bank projections and the
    collapsed-attn2 vectors are refreshed IN PLACE once per clip (`set_clip` -> engine.prepare_reference: 32 small
    GEMMs + 32 tiny matrix-vector products).  That makes the forward — about 640 kernel launches — a hipGraph
    captured once and replayed for every (step, window, clip), each clip's first step included."""

    def __init__(self, unet, S, f, x, ehs, pose):
        dev = x.device
        self.unet, self.S, self.f = unet, S, f
        self.x = torch.empty((S * f,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
        self.temb = torch.zeros((S, unet.config["block_out_channels"][0]), dtype=torch.float32, device=dev)
        self.ehs = torch.empty_like(ehs)
        self.pose = None if pose is None else [torch.empty_like(p) for p in pose]
        self.graph = None
        self.pred = None
        # (the two CFG halves of a step as two forwards on two streams inside the graph measured 6 % slower, round 3: half-size
        #  launches lose more than the overlap of HBM-bound and MFMA-bound kernels returns)

    def matches(self, x, ehs, pose):
        return (tuple(self.x.shape[1:]) == tuple(x.shape[1:]) and self.ehs.shape == ehs.shape and
                self.ehs.dtype == ehs.dtype and (self.pose is None) == (pose is None) and
                (pose is None or all(a.shape == b.shape for a, b in zip(self.pose, pose))))

    def set_clip(self, ehs):
        """new clip: CLIP token into the static buffer, reference-bank projections / attn2 vectors refreshed in place"""
        self.ehs.copy_(ehs)
        self.unet.prepare_reference(self.ehs)

    def set_pose(self, pose):
        if pose is not None:
            for dst, src in zip(self.pose, pose):
                dst.copy_(src)

    def _fill_x(self, x):
        for s_ in range(self.S):
            self.x[s_ * self.f:(s_ + 1) * self.f].copy_(x)

    def _forward(self):
        # (S == 2: `_fill_x` writes the same latents into both halves, the pose features are repeated, the sinusoid rows are equal)
        return self.unet.forward_nhwc(self.x, self.S, self.f, None, self.ehs, self.pose, temb_in=self.temb,
                                      attn2_refresh=False, cfg_shared_input=self.S == 2)

    def eager(self, x, temb):
        """un-captured forward (profilers / ANIP_NO_GRAPH): same static buffers, same in-place reference state"""
        self._fill_x(x)
        self.temb.copy_(temb)
        return self._forward()

    def replay(self, x, temb):
        self._fill_x(x)
        self.temb.copy_(temb)
        if self.graph is None:
            # one un-captured forward first: it fills every lazy cache of the path (packed weights, reference-index
            # tensors, LDS-size attributes of the kernels) — host-to-device copies and allocations that are not allowed
            # while the stream is capturing
            self._forward()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.pred = self._forward()
            self.graph = g
        self.graph.replay()
        return self.pred


class _GraphedFn:
    """`fn(*static_inputs) -> list of tensors` behind static buffers: one un-captured call first (it fills the lazy
    caches of the path: packed weights, kernel attributes — host work that is illegal while a stream captures), then a
    hipGraph captured once and replayed for every later call with same-shaped inputs.  The once-per-clip networks
    (ReferenceNet: ~350 launches for 1.6 TFLOP; PoseGuider) are launch-bound when issued eagerly.  The outputs are the
    graph's own buffers: valid until the next call."""

    def __init__(self, fn, inputs):
        self.fn = fn
        self.static = [torch.empty_like(t) for t in inputs]
        self.graph = None
        self.out = None

    def matches(self, inputs):
        return len(inputs) == len(self.static) and all(a.shape == b.shape and a.dtype == b.dtype and a.device == b.device
                                                       for a, b in zip(self.static, inputs))

    def __call__(self, inputs, use_graph=True):
        for dst, src in zip(self.static, inputs):
            dst.copy_(src)
        if not use_graph:
            return self.fn(*self.static)
        if self.graph is None:
            self.fn(*self.static)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.out = self.fn(*self.static)
            self.graph = g
        self.graph.replay()
        return self.out


class PendingVideo:
    """Frames on their way to the host: device tensor -> pinned buffer on the pipeline's copy stream.  `.result()`
    waits for the copy and returns what the synchronous path returns (the reference's `.cpu().float()` order,
    src/pipelines/pipeline_pose2vid_long.py:125: fp16 over the bus, fp32 on the host)."""

    def __init__(self, host, event, finish, keep):
        self._host, self._event, self._finish, self._keep = host, event, finish, keep
        self._out = None

    def done(self):
        return self._event is None or self._event.query()

    def result(self):
        if self._out is None:
            if self._event is not None:
                self._event.synchronize()
            self._out = self._finish(self._host)
            self._keep = None
        return self._out


class Pose2VideoPipelineOutput(BaseOutput):
    """videos: torch.Tensor (1, 3, L, H, W) fp32 in [0, 1] on the CPU (pipeline_pose2vid_long.py:31-33)"""


class _Progress:
    def __init__(self, total, disable=False):
        self.bar = None
        if not disable:
            try:
                from tqdm.auto import tqdm
                self.bar = tqdm(total=total)
            except Exception:  # pragma: no cover
                self.bar = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        if self.bar is not None:
            self.bar.close()

    def update(self, n=1):
        if self.bar is not None:
            self.bar.update(n)


class _PipelineBase:
    """The slice of `diffusers.DiffusionPipeline` the reference pipeline relies on
    (pipeline_pose2vid_long.py:36,58-70,368,458): used as the base class only where diffusers is not importable."""

    def register_modules(self, **kw):
        names = self.__dict__.setdefault("_names", [])
        for k, v in kw.items():
            setattr(self, k, v)
            if k not in names:
                names.append(k)

    @property
    def components(self):
        return {k: getattr(self, k) for k in self.__dict__.get("_names", [])}

    def to(self, *args, **kwargs):
        """`.to(device)`, `.to(device, dtype)`, `.to(dtype=...)` on every nn.Module component"""
        device = kwargs.pop("device", None)
        dtype = kwargs.pop("dtype", None) or kwargs.pop("torch_dtype", None)
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            else:
                device = a
        for m in self.components.values():
            if isinstance(m, torch.nn.Module):
                m.to(device=device, dtype=dtype)
        return self

    @property
    def device(self):
        for m in self.components.values():
            if isinstance(m, torch.nn.Module):
                return next(m.parameters()).device
        return torch.device("cpu")

    def progress_bar(self, iterable=None, total=None):
        return _Progress(total, bool(self.__dict__.get("_progress_bar_config", {}).get("disable", False)))

    def set_progress_bar_config(self, **kw):
        self.__dict__["_progress_bar_config"] = kw


try:  # the reference's class is a diffusers.DiffusionPipeline (pipeline_pose2vid_long.py:36): keep that where it exists
    from diffusers import DiffusionPipeline as _Base
except Exception:  # diffusers is not part of the bare ROCm image
    _Base = _PipelineBase


class Pose2VideoPipeline(_Base):
    _optional_components = []
    _long = True

    def __init__(self, vae, image_encoder, reference_unet, denoising_unet, pose_guider, scheduler,
                 image_proj_model=None, tokenizer=None, text_encoder=None):
        super().__init__()
        self.register_modules(vae=vae, image_encoder=image_encoder, reference_unet=reference_unet,
                              denoising_unet=denoising_unet, pose_guider=pose_guider, scheduler=scheduler,
                              image_proj_model=image_proj_model, tokenizer=tokenizer, text_encoder=text_encoder)
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)
        from transformers import CLIPImageProcessor
        self.clip_image_processor = CLIPImageProcessor()
        self.ref_image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_convert_rgb=True)
        self.cond_image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_convert_rgb=True,
                                                      do_normalize=True)
        self._hip_vae = None
        hostcfg.bound_host_threads()    # a 128-thread intra-op pool under a 16-CPU cgroup quota freezes the process (hostcfg.py)

    @property
    def _execution_device(self):
        return self.device

    def enable_vae_slicing(self):
        self.vae.enable_slicing()

    def disable_vae_slicing(self):
        self.vae.disable_slicing()

    # -- helpers kept from the reference ----------------------------------------------------------------
    def prepare_extra_step_kwargs(self, generator, eta):
        """pipeline_pose2vid_long.py:128-147"""
        kw = {}
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        if "eta" in params:
            kw["eta"] = eta
        if "generator" in params:
            kw["generator"] = generator
        return kw

    def prepare_latents(self, batch_size, num_channels_latents, width, height, video_length, dtype, device, generator,
                        latents=None):
        """pipeline_pose2vid_long.py:149-183"""
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                             f"effective batch size of {batch_size}. Make sure the batch size matches the length of "
                             "the generators.")
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"latents have shape {tuple(latents.shape)}, expected {shape}")
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def _vae(self):
        """the HIP VAE; a foreign (diffusers) AutoencoderKL is adopted once via its state-dict"""
        if isinstance(self.vae, AutoencoderKL):
            return self.vae
        src = self.vae
        p = next(src.parameters())
        tag = (id(src), str(p.device), p.dtype)
        if self._hip_vae is None or self._hip_vae[0] != tag:
            self._hip_vae = (tag, AutoencoderKL.from_module(src))
        return self._hip_vae[1]

    def decode_latents(self, latents, decode_chunk=16):
        """latents (1, 4, L, h, w) -> numpy (1, 3, L, H, W) fp32 in [0, 1] (pipeline_pose2vid_long.py:113-126)"""
        b, c, f, h, w = latents.shape
        z = ops.ncfhw_to_nhwc((latents.to(self.device).float() * (1 / 0.18215)).contiguous())
        return self._decode_nhwc(z, b, decode_chunk).cpu().float().numpy()

    def _decode_nhwc(self, z, b, decode_chunk=16, as_u8=False):
        """z (b*L, h, w, 4) fp16 (already divided by the scaling factor) -> (b, 3, L, H, W) fp16 in [0,1], or, with
        `as_u8`, display bytes (b*L, H, W, 3) uint8 — the decoder's channels-last output IS the image layout"""
        vae = self._vae()
        outs = []
        for s in range(0, z.shape[0], decode_chunk):
            x = vae.decode_nhwc(z[s:s + decode_chunk].contiguous())
            outs.append(ops.f16_to_u8(x, 0.5, 0.5) if as_u8 else x)
        x = torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]
        if as_u8:
            return x
        return ops.nhwc_to_ncfhw(x, b, out_f32=False, scale=0.5, shift=0.5, clamp01=True)

    def interpolate_latents(self, latents, interpolation_factor, device):
        """pipeline_pose2vid_long.py:293-336: a no-op at the default factor (the only value the scripts use)"""
        if interpolation_factor < 2:
            return latents
        raise NotImplementedError("interpolation_factor >= 2 is dead at the reference's defaults (SURVEY.md §2.1 #14)")

    # -- stages ---------------------------------------------------------------------------------------
    def _hip_clip(self):
        """the HIP form of `self.image_encoder` (a transformers CLIPVisionModelWithProjection, adopted once via its config +
        state-dict like a foreign AutoencoderKL; re-adopted when the module is replaced, moved or re-typed)"""
        src = self.image_encoder
        p = next(src.parameters())
        # every parameter's storage address and in-place version (keying on the first parameter alone would keep
        # serving stale packed weights after a partial module swap / LoRA merge); ~400 tensors, tens of microseconds
        sig = 0
        for q in src.parameters():
            sig = (sig * 1000003 + q.data_ptr() + 7919 * q._version) & 0xFFFFFFFFFFFF
        tag = (id(src), str(p.device), p.dtype, sig)
        cur = self.__dict__.get("_hip_clip_state")
        if cur is None or cur[0] != tag:
            from .clip_vision import CLIPVisionHip
            cur = self.__dict__["_hip_clip_state"] = (tag, CLIPVisionHip.from_module(src))
        return cur[1]

    def _clip_embeds(self, ref_image, device):
        """CLIP image embedding of the reference image (pipeline_pose2vid_long.py:379-385).  Round 6: the tower runs on the HIP
        kernels (`engine.clip_vision_forward`; `clip_vision.CLIPVisionHip` adopts the transformers module's weights) — the last
        torch / rocBLAS / AOTriton kernels inside `pipe(...)` are gone.  The 224 x 224 resize and the CLIPImageProcessor stay on
        the host (PIL / numpy, as in the reference), where the im2col of the 14 x 14 patches is done as well, so ONE (257, 640)
        fp16 matrix is uploaded; the ~220 launches of the forward are a hipGraph like the other once-per-clip networks.
        ANIP_CLIP_TORCH=1: the module as given, eagerly (A/B measurements, foreign towers)."""
        tm = getattr(self, "_tm", None)
        clip_image = self.clip_image_processor.preprocess(ref_image.resize((224, 224)), return_tensors="pt").pixel_values
        if tm is not None:
            tm.mark("clip.preprocess")
        enc = self.image_encoder
        p = next(enc.parameters())
        if os.environ.get("ANIP_CLIP_TORCH") == "1":
            return enc(clip_image.to(p.device, dtype=p.dtype)).image_embeds.to(device)
        hip = self._hip_clip()
        rows = hip.patch_rows(clip_image.float()).to(hip.device)
        if tm is not None:
            tm.mark("clip.h2d")
        if (hip.device.type == "cuda" and ops._WORK is None and not os.environ.get("ANIP_NO_GRAPH")
                and not torch.cuda.is_current_stream_capturing()):
            g = self._aux_graph("clip", hip, (tuple(rows.shape), str(hip.device)),
                                lambda: _GraphedFn(lambda r: [hip.image_embeds_from_rows(r)], [rows]))
            out = g([rows])[0].clone()          # the graph's output buffer is reused by the next clip
        else:
            out = hip.image_embeds_from_rows(rows)
        return out.to(dtype=p.dtype).to(device)

    def _fused_step_coefficients(self, t):
        sch = self.scheduler
        if hasattr(sch, "coefficients"):
            return sch.coefficients(t)
        # a foreign (diffusers) DDIMScheduler: the same arithmetic from its tables
        from .scheduling_ddim import linear_step_form
        cfg = sch.config
        if type(sch).__name__ != "DDIMScheduler" or cfg.clip_sample:
            raise NotImplementedError("the fused CFG+DDIM kernel implements DDIMScheduler with clip_sample=False "
                                      "(configs/inference/inference_v1.yaml:18-23, inference_v2.yaml:24-33)")
        prev = int(t) - cfg.num_train_timesteps // sch.num_inference_steps
        a_t = float(sch.alphas_cumprod[int(t)])
        a_p = float(sch.alphas_cumprod[prev]) if prev >= 0 else float(sch.final_alpha_cumprod)
        return linear_step_form(cfg.prediction_type, math.sqrt(a_t), math.sqrt(max(1 - a_t, 0.0)), math.sqrt(a_p),
                                math.sqrt(max(1 - a_p, 0.0)))

    max_cached_graphs = 4   # window shapes whose captured graph (+ private pool, static buffers) stay resident

    def _get_runners(self):
        """{(S, f, h, w, device): _DenoiseRunner}, least recently used first — dropped whenever the denoising UNet
        re-packs its weights (a captured graph has the packed tensors' addresses baked in) and bounded to
        `max_cached_graphs` entries (`drop_cached_graphs()` empties it)."""
        unet = self.denoising_unet
        tag = (id(unet), unet.packed().serial, getattr(unet, "pool_epoch", 0))
        if self.__dict__.get("_runner_tag") != tag:
            self.__dict__["_runner_tag"] = tag
            self.__dict__["_runners"] = OrderedDict()
        return self.__dict__["_runners"]

    def drop_cached_graphs(self):
        """release every captured graph (denoising step, ReferenceNet, PoseGuider), its static buffers, and the per-shape
        reference-projection / attn2 pools the graphs had baked in (safe once no graph references them: a process serving
        many resolutions would otherwise grow without bound)"""
        self.__dict__["_runners"] = OrderedDict()
        self.__dict__.pop("_runner_tag", None)
        self.__dict__["_aux_graphs"] = {}
        for name in ("denoising_unet", "reference_unet"):
            drop = getattr(getattr(self, name, None), "drop_reference_pools", None)
            if drop is not None:
                drop()

    def _aux_graph(self, kind, module, key, make):
        """{(kind, key): _GraphedFn} for the once-per-clip networks; an entry dies with the module's packed weights
        (their addresses are baked into the graph) and the per-kind population is bounded like the step graphs"""
        cache = self.__dict__.setdefault("_aux_graphs", {})
        tag = (id(module), module.packed().serial, getattr(module, "pool_epoch", 0))
        live = {k: v for k, v in cache.items() if k[0] != kind or v[0] == tag}
        mine = [k for k in live if k[0] == kind]
        if (kind, key) not in live:
            while len(mine) >= max(1, int(self.max_cached_graphs)):
                live.pop(mine.pop(0))
            live[(kind, key)] = (tag, make())
        self.__dict__["_aux_graphs"] = live
        return live[(kind, key)][1]

    def _reference_banks(self, ref_lat, S, ehs, use_graph):
        """ReferenceNet pass at t = 0 in write mode (pipeline_pose2vid_long.py:474-485): fills `module.bank` of the 16
        hooked blocks.  Only the banks matter, so the pass stops after the last bank write."""
        unet = self.reference_unet
        blocks = [rb for rb in unet._ref_blocks.values() if rb.state.mode == "write"]
        x = ref_lat.repeat(S, 1, 1, 1).contiguous()
        if not use_graph:
            unet.forward_nhwc(x, S, 1, 0, ehs, None, final=False, stop_after_last_bank=True)
            return
        dev = x.device
        temb0 = engine.timestep_sinusoid(0, S, unet.config["block_out_channels"][0], dev,
                                         unet.config.get("flip_sin_to_cos", True), unet.config.get("freq_shift", 0))

        def fn(xs, es):
            for rb in blocks:
                rb.node.bank = []
            unet.forward_nhwc(xs, S, 1, None, es, None, final=False, stop_after_last_bank=True, temb_in=temb0)
            banks = [rb.node.bank[-1] for rb in blocks]
            for rb in blocks:
                rb.node.bank = []
            return banks

        g = self._aux_graph("refnet", unet, (S, tuple(x.shape), str(dev), len(blocks)), lambda: _GraphedFn(fn, [x, ehs]))
        for rb, bank in zip(blocks, g([x, ehs])):
            rb.node.bank = [bank]

    @staticmethod
    def _require_gpu(device):
        if device.type != "cuda":
            raise RuntimeError("Pose2VideoPipeline: the denoising path only runs on an MI355X (HIP kernels); "
                               "call pipe.to('cuda') — there is no CPU fallback")

    def _run(self, *args, **kwargs):
        """`_run_clip` with the interpreter's cyclic garbage collector under control (ANIP_GC_CONTROL=0: left alone).  A full
        (generation-2) collection of a process that holds five networks, their packed weights and the captured graphs takes
        45-85 ms, and the interpreter starts one wherever the allocation counters happen to cross their thresholds — round 6
        measured it as ONE host-side stall in roughly every second clip, inside whichever host-bound stage was running (CLIP
        preprocessing, pose upload, VAE encode, the first graph replay), with the GPU idle behind it: per-clip wall times were
        bimodal, 1 237 ms / 1 300 ms (profiles/r06/b_*).  So: automatic collection is switched off while the host is on the
        critical path, and ONE full collection per clip runs where the host has nothing to do — behind the last launch of
        the clip, with the GPU's queue holding the work of the remaining steps and the VAE decode."""
        managed = _GC_CONTROL and gc.isenabled()
        if managed:
            gc.disable()
        try:
            return self._run_clip(*args, **kwargs)
        finally:
            if managed:
                gc.enable()

    @torch.no_grad()
    def _run_clip(self, ref_image, pose_images, ref_pose_image, width, height, video_length, num_inference_steps,
                  guidance_scale, num_images_per_prompt, eta, generator, output_type, return_dict, callback, callback_steps,
                  windows_fn, latents=None, dp_group=None, decode_chunk=16, return_latents=False, use_graph=True,
                  async_output=False):
        device = self._execution_device
        self._require_gpu(device)
        tm = self.__dict__["_tm"] = _StageTimer()
        if num_images_per_prompt != 1:
            raise NotImplementedError("num_images_per_prompt != 1")
        if eta != 0.0:
            raise NotImplementedError("eta != 0")
        do_cfg = guidance_scale > 1.0
        S = 2 if do_cfg else 1
        rank, ws = D.world(dp_group) if dp_group is not None else (0, 1)

        sch = self.scheduler
        try:
            sch.set_timesteps(num_inference_steps, device="cpu")
        except TypeError:  # pragma: no cover
            sch.set_timesteps(num_inference_steps)
        timesteps = [int(t) for t in sch.timesteps.tolist()]

        # (ANIP_NO_GRAPH=1: eager launches, for profilers whose counter collection cannot follow graph replays)
        use_graph = (bool(use_graph) and device.type == "cuda" and ops._WORK is None and
                     not os.environ.get("ANIP_NO_GRAPH"))
        clip_dtype = next(self.image_encoder.parameters()).dtype      # = clip_image_embeds.dtype (prepare_latents draws in it)

        writer = ReferenceAttentionControl(self.reference_unet, do_classifier_free_guidance=do_cfg, mode="write",
                                           batch_size=1, fusion_blocks="full")
        reader = ReferenceAttentionControl(self.denoising_unet, do_classifier_free_guidance=do_cfg, mode="read",
                                           batch_size=1, fusion_blocks="full")

        C = self.denoising_unet.in_channels
        lat = self.prepare_latents(1, C, width, height, video_length, clip_dtype, device, generator,
                                   latents)
        self.prepare_extra_step_kwargs(generator, eta)
        L = lat.shape[2]
        h, w = lat.shape[3], lat.shape[4]
        HWC = h * w * C
        # channels-last fp32 master copy [L][h*w*C] + fp16 copy fed to the UNet
        lat16 = ops.ncfhw_to_nhwc(lat.float().contiguous())          # (L, h, w, C) fp16
        lat32 = lat.float().permute(0, 2, 3, 4, 1).reshape(L, HWC).contiguous()

        # reference image -> VAE latent mean * 0.18215
        vae = self._vae()
        tm.mark("latents")
        ref_t = self.ref_image_processor.preprocess(ref_image, height=height, width=width)
        tm.mark("ref.preprocess")
        ref_t = ref_t.to(device)
        tm.mark("ref.h2d")
        ref_x = ops.ncfhw_to_nhwc(ref_t.float().unsqueeze(2).contiguous())
        if use_graph:     # ~70 launches on ONE image: host-bound when issued eagerly (5 ms per clip), a graph like the other once-per-clip networks
            enc = self._aux_graph("vae_enc", vae, (tuple(ref_x.shape), str(device)),
                                  lambda: _GraphedFn(lambda t: [vae.encode_mean_nhwc(t)], [ref_x]))
            ref_lat = enc([ref_x])[0]
        else:
            ref_lat = vae.encode_mean_nhwc(ref_x)
        ref_lat = (ref_lat.float() * 0.18215).half()                 # (1, h, w, 4): a new tensor (the graph's output buffer is reused)
        tm.mark("latents+vae_encode")

        # pose condition images (numpy path of VaeImageProcessor: values in [-1, 509], see image_processor.py)
        pg = self.pose_guider
        hp, wp = height - height % self.vae_scale_factor, width - width % self.vae_scale_factor
        if all(isinstance(p, np.ndarray) and p.dtype == np.uint8 and p.shape == (hp, wp, 3) for p in pose_images):
            # renderings already at the target size (scripts/pose2vid.py:158 resizes them): upload the bytes once;
            # (L, H, W, 3) uint8 is the channels-last frame batch, 2 v - 1 is applied on the device
            # ... through ONE recycled pinned staging buffer (round 6): the frames are copied into it (the np.stack the pageable path
            # needed anyway) and cross asynchronously — a pageable 12.6-MB `.to(device)` blocks the host for 2.5 ms while the GPU is
            # working off the VAE encode, and the CLIP pre-processing behind it is host work, too.  The device copy is enqueued on
            # the current stream before this function can touch the buffer again (next clip), so one buffer per shape suffices.
            n_p = len(pose_images)
            if device.type == "cuda":
                stage = self.__dict__.setdefault("_pose_pinned", {})
                key = (n_p, hp, wp)
                stacked = stage.get(key)
                if stacked is None:
                    stage.clear()
                    stacked = stage[key] = torch.empty((n_p, hp, wp, 3), dtype=torch.uint8, pin_memory=True)
                else:
                    ev = self.__dict__.get("_pose_pinned_done")
                    if ev is not None:
                        ev.synchronize()        # the previous clip's upload out of this buffer has finished (long ago)
                dst = stacked.numpy()
                for j, p_ in enumerate(pose_images):
                    dst[j] = p_
                tm.mark("pose.stack")
                pose_dev = stacked.to(device, non_blocking=True)
                ev = self.__dict__["_pose_pinned_done"] = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(device))
            else:
                stacked = torch.from_numpy(np.stack(pose_images))
                tm.mark("pose.stack")
                pose_dev = stacked.to(device)
            pose_nhwc = ops.u8_to_f16(pose_dev, 2.0, -1.0)
        else:
            pose = torch.cat([self.cond_image_processor.preprocess(p, height=height, width=width).unsqueeze(2)
                              for p in pose_images], dim=2).to(device=device, dtype=pg.dtype)
            pose_nhwc = ops.ncfhw_to_nhwc(pose)
        # ref_pose_image only feeds PoseGuider's second argument, which never reaches the arithmetic
        # (pose_guider.py: cross_attention_dim=None => no attn2): it is not preprocessed
        tm.mark("pose_preprocess")

        # CLIP image embedding -> one token per sample (uncond = zeros).  Issued HERE (round 5), behind the VAE-encode graph and the
        # pose upload: its host part (PIL resize + CLIPImageProcessor, ~4 ms) runs while the GPU works those off; nothing in front
        # of this point needs it (prepare_latents only takes its dtype; the CLIP tower draws no random numbers)
        clip_image_embeds = self._clip_embeds(ref_image, device)
        ehs = clip_image_embeds.unsqueeze(1)
        if do_cfg:
            ehs = torch.cat([torch.zeros_like(ehs), ehs], dim=0)
        ehs = ehs.contiguous()
        tm.mark("clip")

        # ReferenceNet: one pass at t = 0 (pipeline_pose2vid_long.py:474-485); only the banks matter, so
        # the pass stops after the last bank write.  With a dp_group, rank 0 computes and broadcasts.
        windows = [list(c) for c in windows_fn(L, num_inference_steps)]
        # windows -> ranks by longest-processing-time over their frame counts (equal-length windows: round robin)
        my_windows = (D.shard_balanced([len(c) for c in windows], ws)[rank] if ws > 1 else list(range(len(windows))))
        win_idx = {k: torch.tensor(windows[k], dtype=torch.int32, device=device) for k in my_windows}
        # a dilated window can wrap onto a frame twice (context_stride > 1); the reference's index assignment
        # `noise_pred[:, :, c] = noise_pred[:, :, c] + pred` then keeps the LAST occurrence and counts the frame once
        # (pipeline_pose2vid_long.py:546-549): earlier duplicates are masked out (-1) for the accumulation
        acc_idx = {k: torch.tensor(_last_occurrence_only(windows[k]), dtype=torch.int32, device=device)
                   for k in my_windows}
        pose_cache = {}

        def pose_features(k):
            if k not in pose_cache:
                c = windows[k]
                # batch 1: the CFG duplication does not change train-mode BatchNorm statistics; ref_pose never
                # reaches the arithmetic (pose_guider.py: cross_attention_dim=None => no attn2)
                frames = (pose_nhwc if (pose_nhwc.shape[0] == L and c == list(range(L))) else
                          pose_nhwc[torch.tensor(c, dtype=torch.long, device=device)])
                if use_graph:
                    g = self._aux_graph("pose", pg, (tuple(frames.shape), str(device), bool(pg.training)),
                                        lambda: _GraphedFn(lambda x: list(pg.forward_nhwc(x)), [frames]))
                    fea = g([frames])
                else:
                    fea = pg.forward_nhwc(frames)
                # the graph's outputs are overwritten by the next window's replay: keep copies
                pose_cache[k] = [n.repeat(S, 1, 1, 1).contiguous() if S > 1 else n.clone() for n in fea]
            return pose_cache[k]

        # PoseGuider of the first window on a side stream UNDER the ReferenceNet pass: two independent networks on independent
        # inputs, both far from filling the chip (ReferenceNet: one 64x64 latent, 16 row tiles per GEMM; PoseGuider: 9 ms).
        # Measured A/B inside one call (profiles/r04/y_*): 1398.3 / 1395.9 ms without, 1388.0 / 1378.8 ms with
        side = None
        if ws == 1 and use_graph and my_windows:
            side = self.__dict__.setdefault("_side_streams", {}).get(str(device))
            if side is None:
                side = self.__dict__["_side_streams"][str(device)] = torch.cuda.Stream(device=device)
            cur = torch.cuda.current_stream(device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                pose_features(my_windows[0])
        if rank == 0 or ws == 1:
            self._reference_banks(ref_lat, S, ehs, use_graph)
        if ws > 1:
            self._broadcast_banks(writer, S, h, w, dp_group, device)
        reader.update(writer)
        if side is not None:
            cur.wait_stream(side)
            for n in pose_cache[my_windows[0]]:
                n.record_stream(cur)
        tm.mark("refnet+pose")

        # per-step window sums: acc (S, L, HWC) and counter (L,) are views of one flat buffer (one in-place all-reduce)
        sums_flat, acc, counter = D.window_sum_buffers(S, L, HWC, device)
        single = len(windows) == 1 and windows[0] == list(range(L))
        # Denoising UNet forwards run through a persistent _DenoiseRunner (static input buffers): `set_clip` refreshes the
        # reference-bank projections / attn2 vectors in place, then every step replays the runner's hipGraph, captured
        # once per window shape and reused across clips.
        ucfg = self.denoising_unet.config
        temb_table = torch.stack([engine.timestep_sinusoid(t, S, ucfg["block_out_channels"][0], "cpu",
                                                           ucfg.get("flip_sin_to_cos", True), ucfg.get("freq_shift", 0))
                                  for t in timesteps]).to(device)
        tm.mark("temb_table")
        runners = self._get_runners()
        clip_runners = {}

        def runner_for(k):
            c = windows[k]
            key = (S, len(c), h, w, str(device))
            r = clip_runners.get(key)
            if r is None:
                x0 = lat16[: len(c)]
                r = runners.get(key)
                if r is None or not r.matches(x0, ehs[:S], pose_features(k)):
                    r = runners[key] = _DenoiseRunner(self.denoising_unet, S, len(c), x0, ehs[:S], pose_features(k))
                runners.move_to_end(key)
                while len(runners) > max(1, int(self.max_cached_graphs)):
                    runners.popitem(last=False)
                r.set_clip(ehs[:S])
                if single:
                    r.set_pose(pose_features(k))
                clip_runners[key] = r
            return r

        with self.progress_bar(total=num_inference_steps) as bar:
            for i, t in enumerate(timesteps):
                sums_flat.zero_()
                for k in my_windows:
                    c = windows[k]
                    x = lat16 if single else lat16[win_idx[k].long()]
                    r = runner_for(k)
                    if not single:
                        r.set_pose(pose_features(k))
                    pred = r.replay(x, temb_table[i]) if use_graph else r.eager(x, temb_table[i])
                    if i <= 1:
                        tm.mark(f"unet_step{i}" + ("(graph)" if use_graph else "(eager)"))
                    ops.window_accumulate(pred, acc, counter, acc_idx[k], S, len(c), L, HWC)
                if ws > 1:
                    # also for a single window: the ranks that own no window hold zeros (acc / counter = 0 / 0
                    # otherwise), and every rank needs the step's latents for its share of the VAE decode
                    D.allreduce_flat(sums_flat, dp_group)
                sa, sb, sap, sbp = self._fused_step_coefficients(t)
                ops.cfg_ddim_step(acc, counter, lat32, lat16, S, L, HWC, guidance_scale, sa, sb, sap, sbp)
                bar.update()
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, lat32.reshape(1, L, h, w, C).permute(0, 4, 1, 2, 3).contiguous())
        reader.clear()
        writer.clear()
        tm.mark("unet_steps2+")

        final = lat32.reshape(1, L, h, w, C).permute(0, 4, 1, 2, 3).contiguous()
        if return_latents:
            return final
        z = (lat32 * (1 / 0.18215)).half().reshape(L, h, w, C)
        u8 = output_type == "uint8"           # anything but "tensor" / "uint8" returns numpy, as in the reference (:581-582)
        if ws > 1:
            mine = D.shard_round_robin(L, rank, ws)
            sel = torch.tensor(mine, dtype=torch.long, device=device)
            part = self._decode_nhwc(z[sel].contiguous(), 1, decode_chunk, u8)      # (1, 3, n_local, H, W) | (n_local, H, W, 3)
            local = part if u8 else part[0].permute(1, 0, 2, 3).contiguous()
            frames = D.gather_frames(local, mine, L, 0, dp_group)
            video = None if frames is None else (frames if u8 else frames.permute(1, 0, 2, 3).unsqueeze(0))
        else:
            video = self._decode_nhwc(z, 1, decode_chunk, u8)
        if video is None:
            if _GC_CONTROL and not gc.isenabled():
                gc.collect()
            return None
        tm.mark("vae_decode")
        # (only behind a queue that can hide it: ~0.1 s of host time against >= ~0.2 s of queued GPU work; a small clip leaves
        #  collection to the interpreter, which resumes it when the call returns)
        if _GC_CONTROL and not gc.isenabled() and L * h * w * len(timesteps) * len(windows) >= 16 * 32 * 32 * 10:
            gc.collect()        # the host's idle point: everything of this clip is queued (see _run); ~0.1 s of host time that
            tm.mark("gc(host idle)")   # the un-instrumented call spends under the GPU's queue — only the synchronised stage line shows it

        def finish(host):
            if output_type == "uint8":
                return host.clone() if (async_output or host.is_pinned()) else host
            images = host.float().numpy()       # fp32 up-cast on the host, after the fp16 D2H (reference order)
            return torch.from_numpy(images) if output_type == "tensor" else images

        if async_output:
            images = self._to_host_async(video, finish)
        else:
            # through ONE recycled pinned buffer per shape (round 6): a pageable `.cpu()` of the 25-MB clip is a staged copy of
            # 3-5 ms with the GPU idle behind it; pinned, the same bytes cross in ~0.6 ms.  The buffer never leaves this function:
            # `finish` copies out of it (fp32 up-cast, or a clone of the display bytes)
            if video.is_cuda:
                stage = self.__dict__.setdefault("_sync_pinned", {})
                key = (tuple(video.shape), video.dtype)
                host = stage.get(key)
                if host is None:
                    stage.clear()
                    host = stage[key] = torch.empty(video.shape, dtype=video.dtype, pin_memory=True)
                host.copy_(video)
            else:
                host = video.cpu()
            tm.mark("d2h")
            images = finish(host)
        tm.mark("d2h+float")
        tm.report()
        if not return_dict:
            return images
        return Pose2VideoPipelineOutput(videos=images)

    def _to_host_async(self, video, finish):
        """device -> pinned host buffer on a side stream.  Pinned buffers are recycled per shape once the PendingVideo
        that used them has been consumed (or dropped), so the frames of clip i can still be draining / waiting for their
        consumer while clips i+1, i+2 ... are produced."""
        import weakref
        # one side stream + pinned-slot table PER DEVICE (a pipeline moved to / shared across GPUs must not enqueue the
        # copy on a stream of another device)
        per_dev = self.__dict__.setdefault("_d2h", {})
        st = per_dev.get(str(video.device))
        if st is None:
            st = per_dev[str(video.device)] = {"stream": torch.cuda.Stream(device=video.device), "bufs": {}}
        slots = st["bufs"].setdefault((tuple(video.shape), video.dtype), [])
        slot = None
        for sl in slots:
            owner = sl[1]() if sl[1] is not None else None
            if owner is None or owner._out is not None:
                slot = sl
                break
        if slot is None:
            slot = [torch.empty(video.shape, dtype=video.dtype, pin_memory=True), None]
            slots.append(slot)
        host = slot[0]
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(video.device))
        with torch.cuda.stream(st["stream"]):
            st["stream"].wait_event(ready)
            host.copy_(video, non_blocking=True)
            done = torch.cuda.Event()
            done.record(st["stream"])
        video.record_stream(st["stream"])
        pending = PendingVideo(host, done, finish, video)
        slot[1] = weakref.ref(pending)
        return pending

    def _broadcast_banks(self, writer, S, h, w, group, device):
        """rank 0 holds the 16 banks; other ranks allocate same-shaped tensors; one flat RCCL broadcast."""
        unet = self.reference_unet
        rank, _ = D.world(group)
        shapes = bank_shapes(unet.config, S, h, w)  # derived identically on every rank from the topology
        tensors = []
        for p in writer._paths():
            node = unet._ref_blocks[p].node
            if rank != 0:
                node.bank = [torch.empty(shapes[p], dtype=torch.float16, device=device)]
            assert tuple(node.bank[0].shape) == tuple(shapes[p]), (p, tuple(node.bank[0].shape), shapes[p])
            tensors.append(node.bank[0])
        D.broadcast_tensors(tensors, 0, group)

    # -- public call -----------------------------------------------------------------------------------
    def __call__(self, ref_image, pose_images, ref_pose_image, width, height, video_length, num_inference_steps,
                 guidance_scale, num_images_per_prompt=1, eta=0.0, generator=None, output_type="tensor",
                 return_dict=True, callback=None, callback_steps=1, context_schedule="uniform", context_frames=16,
                 context_stride=1, context_overlap=4, context_batch_size=1, interpolation_factor=1, **kwargs):
        if context_batch_size != 1:
            raise NotImplementedError("context_batch_size must stay 1 (the reference's bank repeat assumes it, "
                                      "pipeline_pose2vid_long.py:541)")
        self.interpolate_latents(None, interpolation_factor, None)
        scheduler_fn = get_context_scheduler(context_schedule)

        def windows_fn(L, steps):
            return scheduler_fn(0, steps, L, context_frames, context_stride, context_overlap)

        return self._run(ref_image, pose_images, ref_pose_image, width, height, video_length, num_inference_steps,
                         guidance_scale, num_images_per_prompt, eta, generator, output_type, return_dict, callback,
                         callback_steps, windows_fn, **kwargs)


def _last_occurrence_only(frames):
    """frame indices with every occurrence but the last of a repeated frame replaced by -1"""
    last = {f: j for j, f in enumerate(frames)}
    return [f if last[f] == j else -1 for j, f in enumerate(frames)]


def bank_shapes(unet_cfg, S, h, w):
    """{hooked block path: (S, tokens, channels)} of the ReferenceNet banks for an (h, w) latent."""
    from .engine import attention_paths
    boc = tuple(unet_cfg["block_out_channels"])
    out = {}
    for p in attention_paths(unet_cfg):
        parts = p.split(".")
        if parts[0] == "down_blocks":
            lvl = int(parts[1])
        elif parts[0] == "mid_block":
            lvl = len(boc) - 1
        else:
            lvl = len(boc) - 1 - int(parts[1])
        hh, ww = h, w
        for _ in range(lvl):
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
        out[p] = (S, hh * ww, boc[lvl])
    return out
