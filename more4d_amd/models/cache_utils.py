"""TeaCache bookkeeping (mirror of MoRe4D/models/cache_utils.py:19-74): skip the DiT blocks of a step when
the accumulated, polynomially rescaled relative-L1 change of the modulated timestep embedding stays under a
threshold, re-using the previous residual.  Host-side control logic; off by default (infer.py:1044)."""
import numpy as np
import torch


def get_teacache_coefficients(model_name):
    table = {
        "Wan2.1-Fun-V1.1-14B": [-5784.54975374, 5449.50911966, -1811.16591783, 256.27178429, -13.02252404],
        "Wan2.1-Fun-14B": [-5784.54975374, 5449.50911966, -1811.16591783, 256.27178429, -13.02252404],
    }
    for k, v in table.items():
        if k in model_name:
            return v
    return None


class TeaCache:
    def __init__(self, coefficients, num_steps, rel_l1_thresh=0.0, num_skip_start_steps=0, offload=True):
        if num_steps < 1:
            raise ValueError(f"`num_steps` must be greater than 0 but is {num_steps}.")
        if rel_l1_thresh < 0:
            raise ValueError(f"`rel_l1_thresh` must be greater than or equal to 0 but is {rel_l1_thresh}.")
        if num_skip_start_steps < 0 or num_skip_start_steps > num_steps:
            raise ValueError("`num_skip_start_steps` must be in [0, num_steps].")
        self.coefficients = coefficients
        self.num_steps = num_steps
        self.rel_l1_thresh = rel_l1_thresh
        self.num_skip_start_steps = num_skip_start_steps
        self.offload = offload
        self.rescale_func = np.poly1d(self.coefficients)
        self.reset()

    @staticmethod
    def compute_rel_l1_distance(prev, cur):
        return ((torch.abs(cur - prev).mean()) / torch.abs(prev).mean()).cpu().item()

    def reset(self):
        self.cnt = 0
        self.should_calc = True
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = None
        self.previous_residual = None
        self.previous_residual_cond = None
        self.previous_residual_uncond = None
