"""Flow-matching sigma schedule and the DPM-Solver++ multistep solver of the 4D-STraG sampler.

Mirror of MoRe4D/utils/fm_solvers.py: `get_sampling_sigmas` (:22-26), `FlowDPMSolverMultistepScheduler.set_timesteps`
(:226-289), the first-order update (:415-483, documented there as "equivalent to DDIM", which for flow prediction
reduces to the Euler step x <- x + (sigma_next - sigma) * v) and the multistep second / third order
updates (:486-677) with the order selection of `step` (:741-779; algorithm dpmsolver++, solver_type midpoint,
lower_order_final, final_sigmas_type zero).  Checked against the reference in tests/golden/sched.npz and
sched_multistep.npz.  The SDE variants are not built.  On the device every update is a linear combination of the
sample and the stored data predictions (`ops.lincomb`); order 1 is fused with classifier-free guidance in one kernel
(`ops.cfg_euler_`).
"""
import math
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
    """Flow DPM-Solver++ multistep solver, orders 1 (DDIM-equivalent Euler, the configured path) to 3.  Same
    ctor / `set_timesteps` / `step` contract as the reference class for the arguments the 4D-STraG pipeline uses."""
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 shift: Optional[float] = 1.0, use_dynamic_shifting=False, final_sigmas_type: str = "zero",
                 algorithm_type: str = "dpmsolver++", solver_type: str = "midpoint", lower_order_final: bool = True,
                 euler_at_final: bool = False, thresholding: bool = False, **unused):
        if solver_order not in (1, 2, 3):
            raise NotImplementedError("solver_order must be 1, 2 or 3")
        if algorithm_type not in ("dpmsolver++", "deis") or solver_type not in ("midpoint", "logrho", "bh1", "bh2") or thresholding:
            raise NotImplementedError("only algorithm_type dpmsolver++ / solver_type midpoint without thresholding is built")
        self.solver_order, self.lower_order_final, self.euler_at_final = solver_order, lower_order_final, euler_at_final
        if prediction_type != "flow_prediction":
            raise NotImplementedError("prediction_type must be flow_prediction")
        if use_dynamic_shifting:
            raise NotImplementedError("use_dynamic_shifting is not part of the configured path")
        if final_sigmas_type != "zero":
            raise NotImplementedError("final_sigmas_type must be 'zero'")
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sig = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)          # float32 like the reference (:178-184)
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.sigma_min, self.sigma_max = sig[-1].item(), sig[0].item()
        self.num_inference_steps = None
        self.timesteps = None
        self.sigmas = None
        self._step_index = None
        self._begin_index = None
        self.order = 1                  # the pipeline's warm-up arithmetic reads scheduler.order (:742)
        self.init_noise_sigma = 1.0
        # diffusers' ConfigMixin surface the callers touch (`scheduler.config.num_train_timesteps`, `.shift`)
        self.config = type("Config", (dict,), {"__getattr__": dict.__getitem__})(
            num_train_timesteps=num_train_timesteps, solver_order=solver_order, prediction_type=prediction_type, shift=shift,
            use_dynamic_shifting=use_dynamic_shifting, final_sigmas_type=final_sigmas_type, algorithm_type=algorithm_type,
            solver_type=solver_type, lower_order_final=lower_order_final, euler_at_final=euler_at_final, thresholding=thresholding)

    def __len__(self):
        return self.num_train_timesteps

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        """Start the step counter at `begin_index` instead of looking the first timestep up (reference :216-224)."""
        self._begin_index = begin_index

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        """Index of `timestep` in the schedule; with duplicates the SECOND match, so that a loop entering mid-schedule does not
        skip a sigma (reference :679-691)."""
        sched = self.timesteps if schedule_timesteps is None else schedule_timesteps
        idx = (sched.cpu() == int(timestep)).nonzero()
        return int(idx[1 if len(idx) > 1 else 0])

    def add_noise(self, original_samples, noise, timesteps):
        """alpha_t x + sigma_t noise with (alpha, sigma) = (1 - s, s) of the schedule entry of each timestep (reference :815-854,
        _sigma_to_alpha_sigma_t :333-335): img2img / in-painting entry points."""
        sig = self.sigmas.to(device=original_samples.device, dtype=original_samples.dtype)
        if self._begin_index is None:
            steps = [self.index_for_timestep(t) for t in timesteps]
        elif self._step_index is not None:
            steps = [self._step_index] * timesteps.shape[0]
        else:
            steps = [self._begin_index] * timesteps.shape[0]
        s_ = sig[steps].flatten()
        while s_.dim() < original_samples.dim():
            s_ = s_.unsqueeze(-1)
        return (1 - s_) * original_samples + s_ * noise

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
        self._begin_index = None
        self._sig64 = np.concatenate([sigmas, [0.0]]).astype(np.float32).astype(np.float64)   # the float32 table, in float64
        self.model_outputs = []          # data predictions x0 = x - sigma v of the last <= solver_order steps (float32)
        self.lower_order_nums = 0

    def _init_step_index(self, timestep):
        if self._begin_index is not None:
            self._step_index = self._begin_index
            return
        t = int(timestep)
        idx = (self.timesteps.cpu() == t).nonzero()
        self._step_index = int(idx[1 if len(idx) > 1 else 0]) if len(idx) else len(self.timesteps) - 1

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def dsigma(self, i):
        return float(self.sigmas[i + 1]) - float(self.sigmas[i])

    # ---- multistep machinery: every update is  x_next = c_x * x + sum_k c_k * m_k  with host-side float64 coefficients
    @staticmethod
    def _lam(s):
        if s <= 0.0:
            return math.inf
        return -math.inf if s >= 1.0 else math.log(1.0 - s) - math.log(s)

    def _coefficients(self, i):
        """(c_x, [c_m0, c_m1, c_m2]) of step i given how many data predictions are usable (:741-779)."""
        n = self.num_inference_steps
        sig = self._sig64
        final = i == n - 1          # final_sigmas_type == "zero"
        second = i == n - 2 and self.lower_order_final and n < 15
        st, s0 = sig[i + 1], sig[i]
        at = 1.0 - st
        h = self._lam(st) - self._lam(s0)
        e = math.expm1(-h) if math.isfinite(h) else -1.0
        cx = st / s0
        if self.solver_order == 1 or self.lower_order_nums < 1 or final:
            return cx, [-at * e]
        h0 = self._lam(s0) - self._lam(sig[i - 1])
        r0 = h0 / h
        if self.solver_order == 2 or self.lower_order_nums < 2 or second:
            # x = cx x - at e m0 - 0.5 at e (m0 - m1) / r0
            return cx, [-at * e * (1.0 + 0.5 / r0), 0.5 * at * e / r0]
        h1 = self._lam(sig[i - 1]) - self._lam(sig[i - 2])
        r1 = h1 / h
        a1 = at * (e / h + 1.0)                  # coefficient of D1
        a2 = -at * ((e + h) / h ** 2 - 0.5)      # coefficient of D2
        # D1_0 = (m0 - m1)/r0, D1_1 = (m1 - m2)/r1, D1 = D1_0 + r0/(r0+r1) (D1_0 - D1_1), D2 = (D1_0 - D1_1)/(r0+r1)
        w10 = a1 * (1.0 + r0 / (r0 + r1)) + a2 / (r0 + r1)      # weight of D1_0
        w11 = -a1 * r0 / (r0 + r1) - a2 / (r0 + r1)             # weight of D1_1
        return cx, [-at * e + w10 / r0, -w10 / r0 + w11 / r1, -w11 / r1]

    def _advance(self, x32, v32, i):
        """x32, v32: float32 sample and (guided) model output of step i -> next sample (float32)."""
        m0 = ops.lincomb([(1.0, x32), (-float(self._sig64[i]), v32)])           # x0 = x - sigma v (:385-388)
        self.model_outputs = (self.model_outputs + [m0])[-self.solver_order:]
        cx, cm = self._coefficients(i)
        terms = [(cx, x32)] + [(c, m) for c, m in zip(cm, reversed(self.model_outputs))]
        nxt = ops.lincomb(terms)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return nxt

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict=True):
        """One solver step; fp32 arithmetic, result cast back to the model-output dtype (:760, :789)."""
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        if self._step_index is None:
            self._init_step_index(timestep)
        if self.solver_order == 1:
            ds = self.dsigma(self._step_index)
            x = sample.float().contiguous().clone()
            # guidance 1 with both halves = model_output gives x + ds * model_output through the fused kernel
            v2 = torch.stack([model_output, model_output]).contiguous()
            ops.cfg_euler_(x, v2, 1.0, ds)
        else:
            x = self._advance(sample.float().contiguous(), model_output.float().contiguous(), self._step_index)
        prev = x.to(model_output.dtype)
        self._step_index += 1
        return _SchedulerOutput(prev) if return_dict else (prev,)

    def step_cfg_(self, latents_f32, v_pair, guidance_scale, i, round_dtype=torch.float32):
        """CFG + solver step on the fp32 latent state, in place (pipeline :820-825).  Order 1: one fused kernel."""
        if self.solver_order == 1:
            return ops.cfg_euler_(latents_f32, v_pair, guidance_scale, self.dsigma(i), round_dtype)
        vu, vc = (h.float().contiguous() for h in v_pair.chunk(2))      # [2B,...]: unconditional half, conditional half
        v = ops.lincomb([(1.0 - guidance_scale, vu), (guidance_scale, vc)])      # v_u + g (v_c - v_u)
        if round_dtype != torch.float32:
            v = v.to(round_dtype).float()
        nxt = self._advance(latents_f32.contiguous().view(v.shape), v, i)
        latents_f32.copy_(nxt.view(latents_f32.shape))
        return latents_f32
