"""`ReferenceAttentionControl` — the reference's operator API for reference attention
(src/models/mutual_self_attention.py:19-363) over the HIP engine.

The reference monkey-patches `forward` of every (Temporal)BasicTransformerBlock.  Here the fused block
lives in `engine.transformer_block`; the control flips its mode ("write": store norm1(x) in
`module.bank`; "read": keys/values = [own tokens ++ bank], CFG-unconditional frames self-only) and keeps
the same hand-off protocol: `module.bank` lists on the transformer-block modules, `update()` copying
writer banks to the reader (rounded through fp16 — :302,338), `clear()`.
"""
import torch

from .unet import _UNetBase


class ReferenceAttentionControl:
    def __init__(self, unet, mode="write", do_classifier_free_guidance=False,
                 attention_auto_machine_weight=float("inf"), gn_auto_machine_weight=1.0, style_fidelity=1.0,
                 reference_attn=True, reference_adain=False, fusion_blocks="midup", batch_size=1):
        if not isinstance(unet, _UNetBase):
            raise TypeError("ReferenceAttentionControl needs an aniportrait_amd UNet2DConditionModel / "
                            f"UNet3DConditionModel, got {type(unet).__name__}")
        self.unet = unet
        assert mode in ["read", "write"]
        assert fusion_blocks in ["midup", "full"]
        if reference_adain:
            raise NotImplementedError("reference_adain is dead code in the reference (never enabled)")
        self.reference_attn = reference_attn
        self.reference_adain = reference_adain
        self.fusion_blocks = fusion_blocks
        self.mode = mode
        self.register_reference_hooks(mode, do_classifier_free_guidance, attention_auto_machine_weight,
                                      gn_auto_machine_weight, style_fidelity, reference_attn, reference_adain,
                                      fusion_blocks=fusion_blocks, batch_size=batch_size)

    def _paths(self, unet=None):
        """hooked transformer blocks in the reference's pairing order (:321-337)"""
        unet = unet or self.unet
        paths = [p[: -len(".transformer_blocks.0")] for p in unet._ref_paths]
        if self.fusion_blocks == "midup":
            paths = [p for p in paths if p.startswith(("mid_block", "up_blocks"))]
        return paths

    def register_reference_hooks(self, mode, do_classifier_free_guidance, attention_auto_machine_weight,
                                 gn_auto_machine_weight, style_fidelity, reference_attn, reference_adain,
                                 dtype=torch.float16, batch_size=1, num_images_per_prompt=1,
                                 device=torch.device("cpu"), fusion_blocks="midup"):
        if batch_size != 1 or num_images_per_prompt != 1:
            raise NotImplementedError("batch_size / num_images_per_prompt != 1: the reference's bank repeat "
                                      "assumes one clip per call (pipeline_pose2vid_long.py:377)")
        if not reference_attn:
            return
        paths = self._paths()
        hooked = set(paths)
        for p, rb in self.unet._ref_blocks.items():
            if p in hooked:
                rb.state.mode = mode
                rb.node.bank = []
                rb.node.attn_weight = float(paths.index(p)) / float(len(paths))
            else:
                rb.state.mode = "plain"
        if mode == "read":
            self.unet._ref_cfg = bool(do_classifier_free_guidance)

    def update(self, writer, dtype=torch.float16):
        if not self.reference_attn:
            return
        rp, wp = self._paths(), self._paths(writer.unet)
        for r, w in zip(rp, wp):
            src = writer.unet._ref_blocks[w].node.bank
            # values are rounded through `dtype` (fp16) exactly as the reference does, also in fp32 runs;
            # the engine consumes fp16 banks
            self.unet._ref_blocks[r].node.bank = [v.clone().to(dtype) for v in src]

    def clear(self):
        if not self.reference_attn:
            return
        for p in self._paths():
            self.unet._ref_blocks[p].node.bank.clear()
