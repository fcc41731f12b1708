"""`DDIMScheduler` with the diffusers 0.24.0 call surface the pipeline uses
(`scripts/pose2vid.py:83-84`, `src/pipelines/pipeline_pose2vid_long.py:373,526,557`): `set_timesteps`,
`timesteps`, `scale_model_input`, `step(...).prev_sample`, `init_noise_sigma`, `order`,
`alphas_cumprod`, `config`.  Configuration in use: configs/inference/inference_v2.yaml:24-33 (linear betas
rescaled to zero terminal SNR, trailing spacing, v-prediction, eta 0).

`coefficients(t)` exposes the four scalars of one deterministic step; the pipeline feeds them to the
fused CFG + DDIM kernel (`anip_cfg_ddim_step`) instead of calling `step` on tensors.
"""
import math

import numpy as np
import torch

from .modeling import BaseOutput, FrozenConfig


class DDIMSchedulerOutput(BaseOutput):
    pass


def _zero_terminal_snr(betas):
    """Rescale so that alpha_bar_T == 0 (Lin et al. 2023, alg. 1)."""
    abar_sqrt = torch.cumprod(1.0 - betas, dim=0).sqrt()
    first, last = abar_sqrt[0].clone(), abar_sqrt[-1].clone()
    abar_sqrt = (abar_sqrt - last) * (first / (first - last))
    abar = abar_sqrt ** 2
    alphas = torch.cat([abar[:1], abar[1:] / abar[:-1]])
    return 1.0 - alphas


def linear_step_form(prediction_type, sa, sb, sap, sbp):
    """(sa, sb, sap, sbp) as anip_cfg_ddim_step takes them, for any prediction type — see DDIMScheduler.coefficients"""
    if prediction_type == "v_prediction":
        return sa, sb, sap, sbp
    if prediction_type == "epsilon":
        if sa <= 0.0:
            raise ValueError("epsilon prediction at alpha_bar = 0 (zero terminal SNR needs v-prediction)")
        return 1.0, 0.0, sap / sa, sbp - sap * sb / sa
    if prediction_type == "sample":
        if sb <= 0.0:
            raise ValueError("sample prediction at alpha_bar = 1")
        return 1.0, 0.0, sbp / sb, sap - sbp * sa / sb
    raise ValueError(f"prediction_type {prediction_type}")


class DDIMScheduler:
    order = 1
    _defaults = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                     trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                     prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                     clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                     rescale_betas_zero_snr=False)

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(self._defaults)
        if unknown:
            raise TypeError(f"DDIMScheduler: unexpected arguments {sorted(unknown)}")
        cfg = dict(self._defaults)
        cfg.update(kwargs)
        self.config = FrozenConfig(cfg)
        n = cfg["num_train_timesteps"]
        if cfg["trained_betas"] is not None:
            betas = torch.tensor(cfg["trained_betas"], dtype=torch.float32)
        elif cfg["beta_schedule"] == "linear":
            betas = torch.linspace(cfg["beta_start"], cfg["beta_end"], n, dtype=torch.float32)
        elif cfg["beta_schedule"] == "scaled_linear":
            betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"beta_schedule {cfg['beta_schedule']}")
        if cfg["thresholding"]:
            raise NotImplementedError("thresholding")
        if cfg["rescale_betas_zero_snr"]:
            betas = _zero_terminal_snr(betas)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.arange(n - 1, -1, -1, dtype=torch.int64)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError(f"num_inference_steps {num_inference_steps} > num_train_timesteps {n}")
        self.num_inference_steps = num_inference_steps
        mode = self.config.timestep_spacing
        if mode == "trailing":
            ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
        elif mode == "leading":
            ts = (np.arange(num_inference_steps) * (n // num_inference_steps)).round()[::-1].astype(np.int64)
            ts = ts + self.config.steps_offset
        elif mode == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].astype(np.int64)
        else:
            raise ValueError(f"timestep_spacing {mode}")
        self.timesteps = torch.from_numpy(np.ascontiguousarray(ts)).to(device)

    def _abar(self, t):
        t = int(t)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_p

    def coefficients(self, t):
        """Four scalars (sa, sb, sap, sbp) of one deterministic (eta = 0, unclipped) step in the form the fused CFG + DDIM kernel
        evaluates:  x0 = sa x - sb m ; eps = sa m + sb x ; x_prev = sap x0 + sbp eps   (m = the guided model output).
        v-prediction (configs/inference/inference_v2.yaml:24-33): the four square roots themselves.  Epsilon / sample prediction
        (configs/inference/inference_v1.yaml:18-23 is epsilon, leading spacing, no zero-SNR): every such step is linear,
        x_prev = cx x + cm m, passed as (1, 0, cx, cm):
            epsilon:  x0 = (x - sb m) / sa          ->  cx = sap / sa,  cm = sbp - sap sb / sa
            sample:   eps = (x - sa m) / sb         ->  cx = sbp / sb,  cm = sap - sbp sa / sb"""
        if self.config.clip_sample:
            raise NotImplementedError("fused step: clip_sample=False only (both shipped inference configs; clipping x0 is not "
                                      "a linear step)")
        a_t, a_p = self._abar(t)
        sa, sb, sap, sbp = math.sqrt(a_t), math.sqrt(max(1.0 - a_t, 0.0)), math.sqrt(a_p), math.sqrt(max(1.0 - a_p, 0.0))
        return linear_step_form(self.config.prediction_type, sa, sb, sap, sbp)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        """tensor form (any device), kept for API parity; the pipeline uses the fused kernel."""
        if eta != 0.0:
            raise NotImplementedError("eta > 0 is never used on the pose2vid path")
        a_t, a_p = self._abar(timestep)
        sa, sb = a_t ** 0.5, (1 - a_t) ** 0.5
        kind = self.config.prediction_type
        if kind == "v_prediction":
            x0 = sa * sample - sb * model_output
            eps = sa * model_output + sb * sample
        elif kind == "epsilon":
            x0 = (sample - sb * model_output) / sa
            eps = model_output
        elif kind == "sample":
            x0 = model_output
            eps = (sample - sa * x0) / sb
        else:
            raise ValueError(kind)
        if self.config.clip_sample:
            r = self.config.clip_sample_range
            x0 = x0.clamp(-r, r)
            if use_clipped_model_output:
                eps = (sample - sa * x0) / sb
        prev = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)
