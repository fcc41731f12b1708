"""The CLIP image encoder of the pose2vid call (`src/pipelines/pipeline_pose2vid_long.py:379-385`:
`self.image_encoder(clip_image.to(device, dtype=...)).image_embeds`) on the HIP kernels.

The pipeline keeps taking the `transformers.CLIPVisionModelWithProjection` module the scripts build
(`scripts/pose2vid.py:83-85`); like a foreign AutoencoderKL it is adopted once through its config + state-dict
(`CLIPVisionHip.from_module`), its weights packed by `engine.PackedNet`, and `engine.clip_vision_forward` replaces the
module's forward — no rocBLAS / AOTriton / torch kernels inside `pipe(...)` any more (round 6).

Built: the architecture `CLIPVisionModelWithProjection` has — class token, learned position table at the native image size,
pre- / post-LayerNorm, quick-GELU MLP (`hidden_act="quick_gelu"`: ViT-L/14 of sd-image-variations, every OpenAI CLIP
checkpoint).  Another activation or `interpolate_pos_encoding` raises NotImplementedError.
"""
import torch

from . import _lib, engine
from .engine import PackedNet

F16 = torch.float16


def patch_rows(pixel_values, patch, k_multiple=64):
    """CLIPImageProcessor output (B, 3, S, S) on the HOST -> (B * (1 + P), Kp) fp16: per image one zero row (class token)
    + its P = (S / patch)^2 patches, each flattened (c, ky, kx) like the Conv2d weight, K zero-padded to a multiple of
    `k_multiple` (the GEMM wants K % 8 == 0; ViT-L/14 has 3 * 14 * 14 = 588)."""
    B, Cc, S, S2 = pixel_values.shape
    assert S == S2 and S % patch == 0 and Cc == 3, f"pixel_values {tuple(pixel_values.shape)} / patch {patch}"
    g = S // patch
    K = Cc * patch * patch
    Kp = (K + k_multiple - 1) // k_multiple * k_multiple
    x = pixel_values.reshape(B, Cc, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, K)
    rows = torch.zeros((B, 1 + g * g, Kp), dtype=F16)
    rows[:, 1:, :K] = x.to(F16)
    return rows.reshape(B * (1 + g * g), Kp)


class CLIPVisionHip:
    """Packed weights + forward of one adopted CLIP vision tower on one device."""

    def __init__(self, config, state_dict, device, dtype):
        cfg = dict(config)
        if cfg.get("hidden_act", "quick_gelu") != "quick_gelu":
            raise NotImplementedError(f"CLIP vision tower with hidden_act={cfg.get('hidden_act')!r}: only quick_gelu is built "
                                      "(ViT-L/14 of sd-image-variations and the OpenAI checkpoints)")
        need = ("hidden_size", "num_attention_heads", "num_hidden_layers", "patch_size", "image_size")
        missing = [k for k in need if k not in cfg]
        if missing:
            raise ValueError(f"CLIP vision config lacks {missing}")
        cfg.setdefault("layer_norm_eps", 1e-5)
        if cfg["hidden_size"] % (8 * cfg["num_attention_heads"]) != 0:
            raise NotImplementedError("CLIP vision tower: the head dim must be a multiple of 8")
        self.config = cfg
        self.device = torch.device(device)
        self.dtype = dtype
        self._require_gpu(self.device)
        self._net = PackedNet({k: v.detach() for k, v in state_dict.items()}, self.device)
        self.pool_epoch = 0

    @staticmethod
    def _require_gpu(device):
        if device.type != "cuda":
            raise _lib.HipLibraryError(f"the CLIP image encoder is on {device}: the pose2vid path only runs on an MI355X "
                                       "through libaniportrait_hip.so (no CPU / PyTorch fallback); move it with .to('cuda')")
        _lib.load()

    @classmethod
    def from_module(cls, module):
        """adopt a transformers CLIPVisionModelWithProjection (config + state-dict key names)"""
        p = next(module.parameters())
        cfg = module.config.to_dict() if hasattr(module.config, "to_dict") else dict(module.config)
        if "vision_config" in cfg and "hidden_size" not in cfg:
            cfg = dict(cfg["vision_config"])
        sd = module.state_dict()
        if "visual_projection.weight" not in sd or "vision_model.embeddings.patch_embedding.weight" not in sd:
            raise NotImplementedError(f"{type(module).__name__}: not a CLIPVisionModelWithProjection state-dict "
                                      "(vision_model.* + visual_projection.weight expected)")
        return cls(cfg, sd, p.device, p.dtype)

    def packed(self):
        return self._net

    def patch_rows(self, pixel_values):
        return patch_rows(pixel_values, self.config["patch_size"])

    def image_embeds_from_rows(self, rows):
        """rows: `patch_rows` output on this device -> (B, projection_dim) fp16"""
        return engine.clip_vision_forward(self._net, self.config, rows)

    def image_embeds(self, pixel_values):
        """pixel_values (B, 3, S, S) (host or device, any float dtype) -> image_embeds (B, projection_dim) in the module's dtype"""
        if pixel_values.shape[-1] != self.config["image_size"] or pixel_values.shape[-2] != self.config["image_size"]:
            raise ValueError(f"Input image size ({pixel_values.shape[-2]}*{pixel_values.shape[-1]}) doesn't match model "
                             f"({self.config['image_size']}*{self.config['image_size']}).")
        rows = self.patch_rows(pixel_values.detach().float().cpu()).to(self.device)
        return self.image_embeds_from_rows(rows).to(self.dtype)
