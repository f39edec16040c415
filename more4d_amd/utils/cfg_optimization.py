"""`cfg_skip`: in the last `cfg_skip_ratio` fraction of the schedule run only the conditional half of the
CFG batch and duplicate its output (mirror of MoRe4D/utils/cfg_optimization.py:5-37 — host logic only)."""
import numpy as np
import torch


def _second_half(v, half, members=False):
    if members and isinstance(v, (list, tuple)):
        # a tuple of batched tensors (first_frame_features = (patch [2B,..], cls [2B,..])): slice every member's batch axis;
        # the reference passes one batched tensor `first_frame` here, which slices the same way (:25-30)
        return type(v)(u[half:] for u in v)
    if hasattr(v, "slice_batch"):          # ContextCache
        return v.slice_batch(half)
    if isinstance(v, (torch.Tensor, list, tuple, np.ndarray)):
        return v[half:]
    return v


def cfg_skip():
    def decorator(func):
        def wrapper(self, x, *args, **kwargs):
            bs = len(x)
            skip = (bs >= 2 and self.cfg_skip_ratio is not None
                    and self.current_steps >= self.num_inference_steps * (1 - self.cfg_skip_ratio))
            if skip:
                half = int(bs // 2)
                x = x[half:]
                args = [_second_half(a, half) for a in args]
                kwargs = {k: _second_half(v, half, members=(k == "first_frame_features")) for k, v in kwargs.items()}
            result = func(self, x, *args, **kwargs)
            if skip:
                result = torch.cat([result, result], dim=0)
            return result
        return wrapper
    return decorator
