"""Test-only stub: `UNet2DConditionLoadersMixin` (LoRA loaders) is inherited but never used."""


class UNet2DConditionLoadersMixin:
    pass
