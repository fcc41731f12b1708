"""Test-only stub: LoRA-compatible layers without a LoRA layer are plain Conv2d/Linear whose
forward ignores `scale` (diffusers 0.24.0 `models/lora.py`)."""
import torch.nn as nn


class LoRACompatibleConv(nn.Conv2d):
    def forward(self, hidden_states, scale=1.0):
        return super().forward(hidden_states)


class LoRACompatibleLinear(nn.Linear):
    def forward(self, hidden_states, scale=1.0):
        return super().forward(hidden_states)
