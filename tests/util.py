"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
MODEL_KEYS = ["denoising_unet", "reference_unet", "vae", "pose_guider"]


def load_manifest(small):
    with open(os.path.join(GOLD, "shapes_small.json" if small else "shapes_real.json")) as f:
        return json.load(f)


def oracle_state_dicts(small, keys=MODEL_KEYS, seed=0):
    """name-hash synthetic fp32 state-dicts with the reference's key names (params only; the
    oracle recomputes the positional-encoding buffers)."""
    from aniportrait_amd.synthetic import synth_state_dict

    man = load_manifest(small)
    return {k: synth_state_dict(man[k]["params"], seed, prefix=k + ".") for k in keys}


def load_golden(name):
    return torch.load(os.path.join(GOLD, name), map_location="cpu")


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
