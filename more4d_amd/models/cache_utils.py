"""TeaCache bookkeeping (mirror of MoRe4D/models/cache_utils.py:19-74): skip the DiT blocks of a step when
the accumulated, polynomially rescaled relative-L1 change of the modulated timestep embedding stays under a
threshold, re-using the previous residual.  Host-side control logic; off by default (infer.py:1044)."""
import numpy as np
import torch

# Rescaling polynomials published with TeaCache, keyed by the substrings of the (lower-cased) model name the reference
# matches, in its order (cache_utils.py:4-16); numeric data, highest power first (np.poly1d).
_COEFFICIENTS = (
    (("wan2.1-t2v-1.3b", "wan2.1-fun-1.3b", "wan2.1-fun-v1.1-1.3b"),
     [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02]),
    (("wan2.1-t2v-14b",),
     [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01]),
    (("wan2.1-i2v-14b-480p",),
     [2.57151496e+05, -3.54229917e+04, 1.40286849e+03, -1.35890334e+01, 1.32517977e-01]),
    (("wan2.1-i2v-14b-720p", "wan2.1-fun-14b", "wan2.2-fun", "wan2.2-i2v-a14b", "wan2.2-t2v-a14b", "wan2.2-ti2v-5b"),
     [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02]),
)

# per-run state: step counter, decision of the conditional pass (re-used by the unconditional one), accumulated distance,
# the previous modulated input and the cached residuals (joint / conditional / unconditional forward)
_RUN_STATE = dict(cnt=0, should_calc=True, accumulated_rel_l1_distance=0, previous_modulated_input=None,
                  previous_residual=None, previous_residual_cond=None, previous_residual_uncond=None)


def get_teacache_coefficients(model_name):
    name = model_name.lower()
    for keys, coeff in _COEFFICIENTS:
        if any(k in name for k in keys):
            return list(coeff)
    print(f"The model {model_name} is not supported by TeaCache.")
    return None


class TeaCache:
    def __init__(self, coefficients, num_steps, rel_l1_thresh=0.0, num_skip_start_steps=0, offload=True):
        for label, bad in (("num_steps` must be greater than 0", num_steps < 1),
                           ("rel_l1_thresh` must be greater than or equal to 0", rel_l1_thresh < 0),
                           (f"num_skip_start_steps` must be in [0, num_steps={num_steps}]",
                            not 0 <= num_skip_start_steps <= num_steps)):
            if bad:
                raise ValueError("`" + label)
        self.coefficients, self.num_steps = coefficients, num_steps
        self.rel_l1_thresh, self.num_skip_start_steps, self.offload = rel_l1_thresh, num_skip_start_steps, offload
        self.rescale_func = np.poly1d(coefficients)
        self.reset()

    @staticmethod
    def compute_rel_l1_distance(prev, cur):
        return float((cur - prev).abs().mean() / prev.abs().mean())

    def reset(self):
        self.__dict__.update(_RUN_STATE)
