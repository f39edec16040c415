"""UniPC multistep solver for flow prediction (mirror of MoRe4D/utils/fm_solvers_unipc.py: `FlowUniPCMultistepScheduler`,
predict_x0, solver_type bh2, lower_order_final; step :655-739, predictor :350-484, corrector :486-626).

Every UniP / UniC update is a linear combination of the last sample, the stored data predictions x0 = x - sigma v and
the new data prediction; the coefficients (incl. the small R rho = b systems) are computed on the host in float64 and
the tensors are combined by `ops.lincomb`.  Pinned to trajectories of the reference scheduler on its own linspace sigma
tables (tests/golden/sched_unipc.npz).  With `get_sampling_sigmas` tables the first sigma is exactly 1 (lambda = -inf) and
the reference's order >= 2 corrector returns NaN; this implementation raises instead."""
import math

import numpy as np
import torch

from .. import ops
from .fm_solvers import FlowDPMSolverMultistepScheduler, _SchedulerOutput


class FlowUniPCMultistepScheduler(FlowDPMSolverMultistepScheduler):
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 shift=1.0, use_dynamic_shifting=False, thresholding: bool = False, predict_x0: bool = True,
                 solver_type: str = "bh2", lower_order_final: bool = True, disable_corrector=(), solver_p=None,
                 final_sigmas_type: str = "zero", **unused):
        if not predict_x0 or solver_type not in ("bh2", "midpoint", "heun", "logrho") or solver_p is not None or thresholding:
            raise NotImplementedError("only predict_x0 / solver_type bh2 without solver_p and thresholding is built")
        super().__init__(num_train_timesteps=num_train_timesteps, solver_order=solver_order, prediction_type=prediction_type,
                         shift=shift, use_dynamic_shifting=use_dynamic_shifting, final_sigmas_type=final_sigmas_type,
                         lower_order_final=lower_order_final)
        self.disable_corrector = list(disable_corrector)
        self.last_sample = None
        self.this_order = 1

    def set_timesteps(self, *args, **kwargs):
        super().set_timesteps(*args, **kwargs)
        self.last_sample = None
        self.this_order = 1

    @staticmethod
    def _bh2(h, rks, n_rhos, half):
        """(h_phi_1, B_h, rhos) of the bh2 update for `rks` (last entry 1.0); reference :431-466 / :566-596."""
        hh = -h
        fin = math.isfinite(hh)
        h_phi_1 = math.expm1(hh) if fin else -1.0
        B_h = h_phi_1
        h_phi_k = (h_phi_1 / hh - 1.0) if fin else -1.0
        fact, R, b = 1, [], []
        for i in range(1, len(rks) + 1):
            R.append([rk ** (i - 1) for rk in rks])
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = (h_phi_k / hh - 1.0 / fact) if fin else -1.0 / fact
        if half:
            return h_phi_1, B_h, [0.5]
        if n_rhos == 0:
            return h_phi_1, B_h, []
        R = np.array(R, dtype=np.float64)[:n_rhos, :n_rhos]
        if not np.isfinite(R).all():
            raise FloatingPointError("UniPC: a sigma of exactly 1 in the table makes the order >= 2 update singular "
                                     "(the reference returns NaN here); use the scheduler's own sigma table")
        return h_phi_1, B_h, [float(r) for r in np.linalg.solve(R, np.array(b, dtype=np.float64)[:n_rhos])]

    @staticmethod
    def _combine(terms):
        """sum of (coefficient, tensor) terms, four at a time."""
        terms = [(c, t) for c, t in terms if c != 0.0]
        out = ops.lincomb(terms[:4])
        rest = terms[4:]
        while rest:
            out = ops.lincomb([(1.0, out)] + rest[:3], out=out)
            rest = rest[3:]
        return out

    def _unipc_advance(self, x32, v32, i):
        sig = self._sig64
        n = self.num_inference_steps
        m_t = ops.lincomb([(1.0, x32), (-float(sig[i]), v32)])                    # x0 prediction of this step
        ms = self.model_outputs
        if i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None:
            # ---- UniC: correct the sample this step started from, with the order of the step that predicted it
            st, s0 = sig[i], sig[i - 1]
            h = self._lam(st) - self._lam(s0)
            rks = [(self._lam(sig[i - (j + 1)]) - self._lam(s0)) / h for j in range(1, self.this_order)] + [1.0]
            h_phi_1, B_h, rhos = self._bh2(h, rks, len(rks), self.this_order == 1)
            at = 1.0 - st
            c_m0 = -at * h_phi_1 + at * B_h * rhos[-1]
            terms = [(st / s0, self.last_sample), (-at * B_h * rhos[-1], m_t)]
            for j in range(1, self.this_order):                                    # rho_j (m_j - m0) / rk_j
                w = -at * B_h * rhos[j - 1] / rks[j - 1]
                terms.append((w, ms[-(j + 1)]))
                c_m0 -= w
            terms.insert(1, (c_m0, ms[-1]))
            x32 = self._combine(terms)
        ms = (ms + [m_t])[-self.solver_order:]
        self.model_outputs = ms
        this_order = min(self.solver_order, n - i) if self.lower_order_final else self.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = x32
        # ---- UniP
        st, s0 = sig[i + 1], sig[i]
        h = self._lam(st) - self._lam(s0)
        rks = [(self._lam(sig[i - j]) - self._lam(s0)) / h for j in range(1, self.this_order)] + [1.0]
        h_phi_1, B_h, rhos = self._bh2(h, rks, len(rks) - 1, self.this_order == 2)
        at = 1.0 - st
        c_m0 = -at * h_phi_1
        terms = [(st / s0, x32)]
        for j in range(1, self.this_order):
            w = -at * B_h * rhos[j - 1] / rks[j - 1]
            terms.append((w, ms[-(j + 1)]))
            c_m0 -= w
        terms.insert(1, (c_m0, ms[-1]))
        nxt = self._combine(terms)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return nxt

    def step(self, model_output, timestep, sample, return_dict=True, generator=None):
        if self.num_inference_steps is None:
            raise ValueError("run set_timesteps first")
        if self._step_index is None:
            self._init_step_index(timestep)
        x = self._unipc_advance(sample.float().contiguous(), model_output.float().contiguous(), self._step_index)
        prev = x.to(model_output.dtype)
        self._step_index += 1
        return _SchedulerOutput(prev) if return_dict else (prev,)

    def step_cfg_(self, latents_f32, v_pair, guidance_scale, i, round_dtype=torch.float32):
        vu, vc = (h.float().contiguous() for h in v_pair.chunk(2))      # [2B,...]: unconditional half, conditional half
        v = ops.lincomb([(1.0 - guidance_scale, vu), (guidance_scale, vc)])
        if round_dtype != torch.float32:
            v = v.to(round_dtype).float()
        nxt = self._unipc_advance(latents_f32.contiguous().view(v.shape), v, i)
        latents_f32.copy_(nxt.view(latents_f32.shape))
        return latents_f32
