"""Test-only stub of `diffusers.schedulers`: `DDIMScheduler` restated from diffusers 0.24.0
(`scheduling_ddim.py`) for the configuration in `configs/inference/inference_v2.yaml:24-33`
(linear betas, rescale_betas_zero_snr, trailing spacing, v-prediction, eta = 0).  The other
scheduler classes the pipeline imports for type hints are placeholders."""
from dataclasses import dataclass

import numpy as np
import torch

from .._placeholder import make_placeholder
from ..configuration_utils import ConfigMixin, register_to_config
from ..utils import BaseOutput


def rescale_zero_terminal_snr(betas):
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    alphas_bar_sqrt = alphas_cumprod.sqrt()
    a0 = alphas_bar_sqrt[0].clone()
    aT = alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt -= aT
    alphas_bar_sqrt *= a0 / (a0 - aT)
    alphas_bar = alphas_bar_sqrt**2
    alphas = alphas_bar[1:] / alphas_bar[:-1]
    alphas = torch.cat([alphas_bar[0:1], alphas])
    return 1 - alphas


@dataclass
class DDIMSchedulerOutput(BaseOutput):
    prev_sample: torch.FloatTensor = None
    pred_original_sample: torch.FloatTensor = None


class DDIMScheduler(ConfigMixin):
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            self.betas = rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "leading":
            ratio = n // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif sp == "trailing":
            ratio = n / num_inference_steps
            ts = np.round(np.arange(n, 0, -ratio)).astype(np.int64)
            ts -= 1
        else:
            raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts).to(device)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif pt == "sample":
            x0 = model_output
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        elif pt == "v_prediction":
            x0 = (a_t**0.5) * sample - (b_t**0.5) * model_output
            eps = (a_t**0.5) * model_output + (b_t**0.5) * sample
        else:
            raise ValueError(pt)
        if self.config.thresholding:
            raise NotImplementedError
        elif self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        variance = self._get_variance(timestep, prev_timestep)
        std_dev_t = eta * variance ** 0.5
        if use_clipped_model_output:
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        direction = (1 - a_prev - std_dev_t**2) ** 0.5 * eps
        prev_sample = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            raise NotImplementedError("eta > 0 is never used on the hot path")
        if not return_dict:
            return (prev_sample,)
        return DDIMSchedulerOutput(prev_sample=prev_sample, pred_original_sample=x0)


for _n in ["DPMSolverMultistepScheduler", "EulerAncestralDiscreteScheduler", "EulerDiscreteScheduler",
           "LMSDiscreteScheduler", "PNDMScheduler"]:
    globals()[_n] = make_placeholder(_n)
del _n
