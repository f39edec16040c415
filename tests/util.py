"""Shared helpers for the test-suite (fixtures loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def load_keys(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return {k: tuple(v) for k, v in json.load(fh).items()}


def rel_err(a, b):
    """max |a-b| / max |b|  (the '1e-3 relative fp32' metric of BASELINE.json)."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rms_rel_err(a, b):
    a = a.double()
    b = b.double()
    return float(((a - b).pow(2).mean().sqrt()) / b.pow(2).mean().sqrt().clamp_min(1e-30))
