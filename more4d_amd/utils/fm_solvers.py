"""Flow-matching sigma schedule and the order-1 solver of the 4D-STraG sampler.

Mirror of the configured path of MoRe4D/utils/fm_solvers.py: `get_sampling_sigmas` (:22-26),
`FlowDPMSolverMultistepScheduler.set_timesteps` (:226-289) and the first-order update (:415-483, documented
there as "equivalent to DDIM"), which for flow prediction reduces to the Euler step
x <- x + (sigma_next - sigma) * v (checked against the reference in tests/golden/sched.npz).
Orders 2/3 and the SDE variants are not the configured path and are not built.  The device-side update is
fused with classifier-free guidance in one kernel (`ops.cfg_euler_`).
"""
from typing import List, Optional, Union

import numpy as np
import torch

from .. import ops


def get_sampling_sigmas(sampling_steps, shift):
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    if timesteps is not None and sigmas is not None:
        raise ValueError("Only one of `timesteps` or `sigmas` can be passed.")
    if timesteps is not None:
        raise ValueError("custom timesteps are not supported by this scheduler; pass sigmas")
    if sigmas is not None:
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, len(scheduler.timesteps)


class _SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class FlowDPMSolverMultistepScheduler:
    """Order-1 flow solver (DDIM-equivalent Euler).  Same ctor/`set_timesteps`/`step` contract as the
    reference class for the arguments the 4D-STraG pipeline uses."""
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 1, prediction_type: str = "flow_prediction",
                 shift: Optional[float] = 1.0, use_dynamic_shifting=False, final_sigmas_type: str = "zero", **unused):
        if solver_order != 1:
            raise NotImplementedError("only solver_order=1 (the configured DDIM-equivalent path) is built")
        if prediction_type != "flow_prediction":
            raise NotImplementedError("prediction_type must be flow_prediction")
        if use_dynamic_shifting:
            raise NotImplementedError("use_dynamic_shifting is not part of the configured path")
        if final_sigmas_type != "zero":
            raise NotImplementedError("final_sigmas_type must be 'zero'")
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sig = 1.0 - alphas
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_min, self.sigma_max = float(sig[-1]), float(sig[0])
        self.num_inference_steps = None
        self.timesteps = None
        self.sigmas = None
        self._step_index = None

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: Union[int, None] = None, device=None,
                      sigmas: Optional[List[float]] = None, mu=None, shift: Optional[float] = None):
        if sigmas is None:
            sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        sigmas = np.asarray(sigmas, dtype=np.float64)
        if shift is None:
            shift = self.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.num_train_timesteps
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(timesteps).to(device=device, dtype=torch.int64)  # truncation, :276-277
        self.num_inference_steps = len(timesteps)
        self._step_index = None

    def _init_step_index(self, timestep):
        t = int(timestep)
        idx = (self.timesteps.cpu() == t).nonzero()
        self._step_index = int(idx[1 if len(idx) > 1 else 0]) if len(idx) else len(self.timesteps) - 1

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def dsigma(self, i):
        return float(self.sigmas[i + 1]) - float(self.sigmas[i])

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict=True):
        """prev = sample + (sigma_next - sigma) * model_output, fp32 then cast back (:760, :789)."""
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        if self._step_index is None:
            self._init_step_index(timestep)
        ds = self.dsigma(self._step_index)
        x = sample.float().contiguous().clone()
        # guidance 1 with both halves = model_output gives x + ds * model_output through the fused kernel
        v2 = torch.stack([model_output, model_output]).contiguous()
        ops.cfg_euler_(x, v2, 1.0, ds)
        prev = x.to(model_output.dtype)
        self._step_index += 1
        return _SchedulerOutput(prev) if return_dict else (prev,)

    def step_cfg_(self, latents_f32, v_pair, guidance_scale, i, round_dtype=torch.float32):
        """Fused CFG + Euler on the fp32 latent state, in place (pipeline :820-825 in one kernel)."""
        return ops.cfg_euler_(latents_f32, v_pair, guidance_scale, self.dsigma(i), round_dtype)
