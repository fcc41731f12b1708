"""`UNet3DConditionModel` (denoiser) and `UNet2DConditionModel` (ReferenceNet): the reference's operator
API (src/models/unet_3d.py:27-29,399-410,582-590; src/models/unet_2d_condition.py:64,872-887) over the
HIP engine.  Same constructor config, same state-dict key names, same forward signatures and
return types; the arithmetic runs in `engine.unet_forward` on libaniportrait_hip.so.
"""

import torch

from . import engine, hipops as ops
from .modeling import BaseOutput, HipModel, load_state_file
from .params import transformer_block_paths, unet_shapes

_SD15_DEFAULTS = dict(
    sample_size=None, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=None, mid_block_type=None, up_block_types=None, only_cross_attention=False,
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1,
    act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280, attention_head_dim=8,
    dual_cross_attention=False, use_linear_projection=False, class_embed_type=None, num_class_embeds=None,
    upcast_attention=False, resnet_time_scale_shift="default",
)


class UNet3DConditionOutput(BaseOutput):
    """src/models/unet_3d.py:27-29"""


class UNet2DConditionOutput(BaseOutput):
    """src/models/unet_2d_condition.py:52-61"""


class _RefBlock:
    """What ReferenceAttentionControl manipulates on a transformer block: `.bank` (list of tensors,
    as in src/models/mutual_self_attention.py:286-300) plus the engine-side state."""

    def __init__(self, node):
        self.node = node
        self.state = engine.RefState()
        node.bank = []
        node.attn_weight = 1.0

    def engine_state(self):
        st = self.state
        bank = self.node.bank[0] if (st.mode == "read" and len(self.node.bank) > 0) else None
        if bank is not st.bank:  # bank replaced / cleared: K_ref / V_ref^T are re-projected (in place) on next use
            st.bank, st.stale = bank, True
        return st


class _UNetBase(HipModel):
    three_d = False

    @classmethod
    def _shapes(cls, cfg):
        cls._check_config(cfg)
        return unet_shapes(cfg, cls.three_d)

    @staticmethod
    def _check_config(cfg):
        def bad(what):
            raise NotImplementedError(f"UNet config outside the pose2vid hot path: {what} "
                                      "(only the SD-1.5 topology of configs/inference/inference_v2.yaml is built)")
        if len(cfg["block_out_channels"]) != 4:
            bad("block_out_channels must have 4 entries")
        if cfg["use_linear_projection"] or cfg["dual_cross_attention"] or cfg["only_cross_attention"]:
            bad("use_linear_projection / dual_cross_attention / only_cross_attention")
        if cfg["class_embed_type"] is not None or cfg["num_class_embeds"] is not None:
            bad("class embeddings")
        if cfg["resnet_time_scale_shift"] != "default" or cfg["act_fn"] != "silu":
            bad("resnet_time_scale_shift / act_fn")
        if not isinstance(cfg["attention_head_dim"], int):
            bad("per-block attention_head_dim")
        if cfg["center_input_sample"] or cfg["mid_block_scale_factor"] != 1:
            bad("center_input_sample / mid_block_scale_factor")

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        params, _ = unet_shapes(self.config, self.three_d)
        self._ref_paths = transformer_block_paths(params)  # ReferenceAttentionControl pairing order
        self._ref_blocks = {}
        for p in self._ref_paths:
            self._ref_blocks[p[: -len(".transformer_blocks.0")]] = _RefBlock(self.get_submodule(p))
        self._attn2_cache = engine.Attn2Cache()
        self._ref_cfg = False  # do_classifier_free_guidance of the attached reader control

    def _invalidate(self):
        super()._invalidate()
        self.drop_reference_pools()

    def drop_reference_pools(self):
        """free the per-shape K_ref / V_ref^T and collapsed-attn2 buffers (only legal while no captured graph reads
        them: the pipeline calls this from `drop_cached_graphs`, `_invalidate` when the weights change)"""
        if "_attn2_cache" in self.__dict__:
            self._attn2_cache.drop()
            for rb in self._ref_blocks.values():
                rb.state.drop()
        # every pipeline object that captured graphs over this module compares this counter in its cache tag: a second
        # pipeline sharing the module drops its graphs instead of replaying them over freed buffers
        self.__dict__["_pool_epoch"] = self.pool_epoch + 1

    @property
    def pool_epoch(self):
        return self.__dict__.get("_pool_epoch", 0)

    # ------------------------------------------------------------------------------------------------
    def _engine_refs(self):
        return {p: rb.engine_state() for p, rb in self._ref_blocks.items()}

    def _ref_index(self, b, f, refs, device):
        """per-frame reference sample (-1: CFG-unconditional frame, self-attention only) —
        src/models/mutual_self_attention.py:77-85,148-186"""
        if not any(r.mode == "read" and r.bank is not None for r in refs.values()):
            return None
        nb = next(r.bank.shape[0] for r in refs.values() if r.mode == "read" and r.bank is not None)
        key = (b, f, bool(self._ref_cfg), nb, str(device))
        cache = self.__dict__.setdefault("_ref_index_cache", {})
        if key in cache:                       # no host->device copy in steady state (hipGraph-capturable)
            return cache[key]
        N = b * f
        idx = torch.arange(N, dtype=torch.int32) // f
        if self._ref_cfg:
            idx[: N // 2] = -1
        if int(idx.max()) >= nb:
            raise ValueError(f"reference bank holds {nb} sample(s) but the batch addresses sample {int(idx.max())}")
        cache[key] = (idx.to(device), int((idx >= 0).sum()))
        return cache[key]

    def _check_unsupported(self, **kw):
        for k, v in kw.items():
            if v is not None:
                raise NotImplementedError(f"{type(self).__name__}.forward: `{k}` is not part of the pose2vid hot path")

    def prepare_reference(self, encoder_hidden_states, attn2_slot=0):
        """refresh (in place) the reference-bank projections and collapsed-attn2 vectors for a new clip — see
        engine.prepare_reference; later `forward_nhwc(..., attn2_refresh=False)` calls / graph replays reuse them"""
        engine.prepare_reference(self.packed(), self.config, self._engine_refs(), encoder_hidden_states, self._attn2_cache,
                                 attn2_slot)

    def forward_nhwc(self, x, b, f, timestep, encoder_hidden_states, pose_nhwc=None, final=True,
                     stop_after_last_bank=False, temb_in=None, attn2_refresh=True, tap=None, ref_index=None, attn2_slot=0,
                     cfg_shared_input=False):
        """channels-last entry used by the pipeline: x (b*f, h, w, C) fp16 on the GPU.  temb_in: device fp32
        (b, C0) timestep sinusoid replacing `timestep` (see engine.unet_forward).  attn2_refresh=False: reuse the
        collapsed-attn2 vectors the previous forward computed for this batch size (engine.Attn2Cache).  ref_index: explicit
        (int32 tensor (b*f,), number of frames with a reference) instead of the CFG layout derived from b / f — one CFG half
        of a step run as its own forward; attn2_slot: that forward's own attn2 buffers.  cfg_shared_input: the b = 2 samples are a
        CFG pair built by duplication (identical x, pose features, timestep) — see engine.unet_forward."""
        net = self.packed()
        refs = self._engine_refs()
        ridx = ref_index if ref_index is not None else self._ref_index(b, f, refs, net.device)
        out = engine.unet_forward(net, self.config, x, b, f, timestep, encoder_hidden_states, self._attn2_cache,
                                  refs, self.three_d, ridx, pose_nhwc, final, stop_after_last_bank, temb_in,
                                  attn2_refresh, tap, attn2_slot, cfg_shared_input)
        for p, rb in self._ref_blocks.items():  # write mode: append to module.bank like the hacked forward
            if rb.state.mode == "write" and rb.state.written is not None:
                rb.node.bank.append(rb.state.written)
                rb.state.written = None
        return out


class UNet3DConditionModel(_UNetBase):
    """Denoising UNet with reference attention and temporal motion modules (src/models/unet_3d.py:33-580)."""
    three_d = True
    config_defaults = dict(
        _SD15_DEFAULTS, use_inflated_groupnorm=False, use_motion_module=False, motion_module_resolutions=(1, 2, 4, 8),
        motion_module_mid_block=False, motion_module_decoder_only=False, motion_module_type=None,
        motion_module_kwargs={}, unet_use_cross_frame_attention=None, unet_use_temporal_attention=None,
    )

    @staticmethod
    def _check_config(cfg):
        _UNetBase._check_config(cfg)
        if cfg["unet_use_cross_frame_attention"] or cfg["unet_use_temporal_attention"]:
            raise NotImplementedError("unet_use_cross_frame_attention / unet_use_temporal_attention")
        if cfg["use_motion_module"]:
            if cfg["motion_module_type"] != "Vanilla" or cfg["motion_module_decoder_only"]:
                raise NotImplementedError("only motion_module_type='Vanilla', decoder_only=False")
            mk = cfg["motion_module_kwargs"]
            if mk.get("temporal_attention_dim_div", 1) != 1 or mk.get("num_transformer_block", 1) != 1:
                raise NotImplementedError("motion_module_kwargs outside inference_v2.yaml")
            if any(t != "Temporal_Self" for t in mk.get("attention_block_types", ())):
                raise NotImplementedError("attention_block_types other than Temporal_Self")
            if not mk.get("temporal_position_encoding", False):
                raise NotImplementedError("temporal_position_encoding=False")

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, pose_cond_fea=None,
                attention_mask=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict=True):
        """sample (b, c, f, h, w); encoder_hidden_states (b, 1, D); pose_cond_fea: 5 tensors (b, C, f, h', w')
        (src/models/unet_3d.py:399-580)."""
        self._check_unsupported(class_labels=class_labels, attention_mask=attention_mask,
                                down_block_additional_residuals=down_block_additional_residuals,
                                mid_block_additional_residual=mid_block_additional_residual)
        assert sample.dim() == 5, f"Expected hidden_states to have ndim=5, but got ndim={sample.dim()}."
        b, c, f, h, w = sample.shape
        x = ops.ncfhw_to_nhwc(sample)
        pose = None if pose_cond_fea is None else [ops.ncfhw_to_nhwc(p) for p in pose_cond_fea]
        out = self.forward_nhwc(x, b, f, timestep, encoder_hidden_states, pose)
        out = ops.nhwc_to_ncfhw(out, b, out_f32=(sample.dtype == torch.float32))
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, motion_module_path, subfolder=None, unet_additional_kwargs=None,
                           mm_zero_proj_out=False):
        """src/models/unet_3d.py:582-673: SD `unet/config.json` + 2-D weights + motion-module weights,
        `load_state_dict(strict=False)`."""
        from pathlib import Path
        pretrained_model_path = Path(pretrained_model_path)
        motion_module_path = Path(motion_module_path)
        if subfolder is not None:
            pretrained_model_path = pretrained_model_path.joinpath(subfolder)
        config_file = pretrained_model_path / "config.json"
        if not (config_file.exists() and config_file.is_file()):
            raise RuntimeError(f"{config_file} does not exist or is not a file")
        unet_config = cls.load_config(str(config_file))
        unet_config["_class_name"] = cls.__name__
        model = cls.from_config(unet_config, **(unet_additional_kwargs or {}))
        state_dict = None
        for fn in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"):
            if (pretrained_model_path / fn).exists():
                state_dict = load_state_file(str(pretrained_model_path / fn))
                break
        if state_dict is None:
            raise FileNotFoundError(f"no 2-D UNet weights under {pretrained_model_path}")
        if motion_module_path.exists() and motion_module_path.is_file():
            if motion_module_path.suffix.lower() in (".pth", ".pt", ".ckpt", ".safetensors"):
                mm = load_state_file(str(motion_module_path))
            else:
                raise RuntimeError(f"unknown file format for motion module weights: {motion_module_path.suffix}")
            if mm_zero_proj_out:
                mm = {k: v for k, v in mm.items() if "proj_out" not in k}
            state_dict.update(mm)
        model.load_state_dict(state_dict, strict=False)
        return model


class UNet2DConditionModel(_UNetBase):
    """ReferenceNet: SD-1.5 UNet2DConditionModel without conv_norm_out / conv_out
    (src/models/unet_2d_condition.py:645-653,1295-1299); its output is discarded by the pipeline, only the
    banks written under ReferenceAttentionControl(mode="write") matter."""
    three_d = False
    config_defaults = dict(_SD15_DEFAULTS)

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                encoder_attention_mask=None, return_dict=True):
        self._check_unsupported(class_labels=class_labels, timestep_cond=timestep_cond, attention_mask=attention_mask,
                                cross_attention_kwargs=cross_attention_kwargs, added_cond_kwargs=added_cond_kwargs,
                                down_block_additional_residuals=down_block_additional_residuals,
                                mid_block_additional_residual=mid_block_additional_residual,
                                encoder_attention_mask=encoder_attention_mask)
        b = sample.shape[0]
        x = ops.ncfhw_to_nhwc(sample.unsqueeze(2))
        out = self.forward_nhwc(x, b, 1, timestep, encoder_hidden_states, None, final=False)
        out = ops.nhwc_to_ncfhw(out, b, out_f32=(sample.dtype == torch.float32)).squeeze(2)
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)
