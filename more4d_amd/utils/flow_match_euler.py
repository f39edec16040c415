"""`FlowMatchEulerDiscreteScheduler` — the reference's DEFAULT sampler object ("Flow", infer.py:667-682; train_wan.py:768-770
uses its `.timesteps` / `.sigmas` tables for the flow-matching noising :1909-1924).  It is a diffusers class
(requirements.txt: `diffusers>=0.30.1`, unpinned, not under /root/reference), so this is a restatement of the published
algorithm — **parity unpinned** (DESIGN.md §2):
    ctor:          sigma_i = t_i / N for t = N..1, shifted  s -> shift*s / (1 + (shift-1)*s)
    set_timesteps: N' sigmas on linspace(sigma_max*N, sigma_min*N, N')/N (or the caller's), shifted, a final 0 appended;
                   timesteps = sigmas * N (float32, NOT truncated)
    step:          prev = sample + (sigma_next - sigma) * model_output   in float32, cast back
The update is the same Euler step as the in-tree order-1 solver, which IS pinned to the reference (sched.npz); the
device side is the fused CFG + Euler kernel."""
import math

import numpy as np
import torch

from .. import ops
from .fm_solvers import _SchedulerOutput


class _Config(dict):
    __getattr__ = dict.__getitem__


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, use_dynamic_shifting: bool = False,
                 base_shift: float = 0.5, max_shift: float = 1.15, base_image_seq_len: int = 256,
                 max_image_seq_len: int = 4096, **unused):
        self.config = _Config(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=use_dynamic_shifting,
                              base_shift=base_shift, max_shift=max_shift, base_image_seq_len=base_image_seq_len,
                              max_image_seq_len=max_image_seq_len)
        t = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(t).to(torch.float32) / num_train_timesteps
        if not use_dynamic_shifting:
            sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self.sigma_min, self.sigma_max = sig[-1].item(), sig[0].item()
        self.num_inference_steps = None
        self._step_index = None
        self._begin_index = None

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    @staticmethod
    def time_shift(mu, sigma, t):
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
        n = self.config.num_train_timesteps
        if sigmas is None:
            ts = np.linspace(self.sigma_max * n, self.sigma_min * n, num_inference_steps)
            sigmas = ts / n
        sigmas = np.asarray(sigmas, dtype=np.float32)
        if self.config.use_dynamic_shifting:
            if mu is None:
                raise ValueError("you have to pass a value for `mu` when `use_dynamic_shifting` is set to be `True`")
            sigmas = self.time_shift(mu, 1.0, sigmas)
        else:
            s = self.config.shift
            sigmas = s * sigmas / (1 + (s - 1) * sigmas)
        sig = torch.from_numpy(np.asarray(sigmas, dtype=np.float32))
        self.timesteps = (sig * n).to(device=device)
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self.num_inference_steps = len(sig)
        self._step_index = None
        self._begin_index = None

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def scale_noise(self, sample, timestep, noise):
        """Forward process of flow matching: sigma * noise + (1 - sigma) * sample (the noising of train_wan.py:1926)."""
        idx = [int((self.timesteps.cpu() - float(t)).abs().argmin()) for t in torch.as_tensor(timestep).reshape(-1)]
        sigma = self.sigmas[idx].to(sample.device, sample.dtype).reshape(-1, *([1] * (sample.dim() - 1)))
        return sigma * noise + (1.0 - sigma) * sample

    def _init_step_index(self, timestep):
        if self._begin_index is not None:
            self._step_index = self._begin_index
            return
        d = (self.timesteps.cpu().float() - float(timestep)).abs()
        self._step_index = int(d.argmin())

    def dsigma(self, i):
        return float(self.sigmas[i + 1]) - float(self.sigmas[i])

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, **unused):
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        if self._step_index is None:
            self._init_step_index(timestep)
        x = sample.float().contiguous().clone()
        ops.cfg_euler_(x, torch.stack([model_output, model_output]).contiguous(), 1.0, self.dsigma(self._step_index))
        prev = x.to(model_output.dtype)
        self._step_index += 1
        return _SchedulerOutput(prev) if return_dict else (prev,)

    def step_cfg_(self, latents_f32, v_pair, guidance_scale, i, round_dtype=torch.float32):
        """Fused CFG + Euler on the fp32 latent state, in place (pipeline :820-825 in one kernel)."""
        return ops.cfg_euler_(latents_f32, v_pair, guidance_scale, self.dsigma(i), round_dtype)

    def __len__(self):
        return self.config.num_train_timesteps
