"""Oracle: flow-matching sigma schedule, order-1 (Euler == "DDIM-equivalent") update, CFG and
the 4D-STraG denoise loop.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Reference: MoRe4D/utils/fm_solvers.py (get_sampling_sigmas :22-26, set_timesteps :226-289,
dpm_solver_first_order_update :415-483, step :706-797) and the loop in
MoRe4D/pipeline/pipeline_wan_fun_control.py:741-840.  For flow prediction the order-1
DPM-Solver++ update  x_t = (s_t/s_s) x - a_t (exp(-h)-1) x0,  x0 = x - s_s v,  a = 1-s,
h = log(a_t/s_t) - log(a_s/s_s)  reduces algebraically to  x + (s_t - s_s) v.
"""
import numpy as np
import torch


def sampling_sigmas(steps, shift):
    """fm_solvers.py:22-26."""
    s = np.linspace(1, 0, steps + 1)[:steps]
    return shift * s / (1 + (shift - 1) * s)


def set_timesteps(sigmas, num_train_timesteps=1000):
    """fm_solvers.py:226-289 with config.shift=1, final_sigmas_type='zero':
    returns (timesteps int64 [N], sigmas float32 [N+1])."""
    sig = np.asarray(sigmas, dtype=np.float64)
    timesteps = torch.from_numpy(sig * num_train_timesteps).to(torch.int64)  # truncation (:276-277)
    sig = np.concatenate([sig, [0.0]]).astype(np.float32)
    return timesteps, torch.from_numpy(sig)


def euler_step(x, v, sigma, sigma_next):
    """One scheduler.step: fp32 upcast (:760), x + (sigma_next - sigma) * v."""
    return x.float() + (float(sigma_next) - float(sigma)) * v.float()


def cfg_combine(v_uncond, v_cond, scale):
    """pipeline_wan_fun_control.py:820-822."""
    return v_uncond + scale * (v_cond - v_uncond)


def denoise_loop(model_fn, latents, timesteps, sigmas, guidance_scale):
    """pipeline_wan_fun_control.py:745-825: model_fn(x2 [2B,...], t [2B]) -> v [2B,...] with the
    uncond half first (negative prompt + prompt, :571)."""
    x = latents
    for i, t in enumerate(timesteps):
        v = model_fn(torch.cat([x, x]), t.expand(2 * x.shape[0]))
        vu, vc = v.chunk(2)
        x = euler_step(x, cfg_combine(vu, vc, guidance_scale), sigmas[i], sigmas[i + 1]).to(v.dtype)
    return x


def dpmpp_multistep_loop(model_fn, x, sigmas, timesteps, order):
    """FlowDPMSolverMultistepScheduler (algorithm dpmsolver++, solver_type midpoint, lower_order_final=True,
    final_sigmas_type zero) for solver_order 1..3, float64 coefficient arithmetic restated from
    MoRe4D/utils/fm_solvers.py: convert_model_output :385-388 (x0 = x - sigma v), first order :415-483,
    second :486-594, third :596-677, order selection / lower-order rules :741-779.
    sigmas has len(timesteps) + 1 entries (last 0).  Returns the list of samples after every step."""
    import math
    n = len(timesteps)
    sig = [float(s) for s in sigmas]

    def lam(s):
        if s <= 0.0:
            return math.inf
        return -math.inf if s >= 1.0 else math.log(1.0 - s) - math.log(s)

    ms, lower, traj = [], 0, []
    x = x.float()
    for i in range(n):
        v = model_fn(x, timesteps[i])
        ms.append(x - sig[i] * v)                       # data prediction
        ms = ms[-3:]
        final = i == n - 1                              # final_sigmas_type == "zero" => first order on the last step
        second = i == n - 2 and n < 15
        st, s0 = sig[i + 1], sig[i]
        at = 1.0 - st
        h = lam(st) - lam(s0)
        e = math.expm1(-h) if math.isfinite(h) else -1.0        # exp(-h) - 1
        if order == 1 or lower < 1 or final:
            x = (st / s0) * x - (at * e) * ms[-1]
        elif order == 2 or lower < 2 or second:
            h0 = lam(s0) - lam(sig[i - 1])
            r0 = h0 / h
            d1 = (1.0 / r0) * (ms[-1] - ms[-2])
            x = (st / s0) * x - (at * e) * ms[-1] - 0.5 * (at * e) * d1
        else:
            h0, h1 = lam(s0) - lam(sig[i - 1]), lam(sig[i - 1]) - lam(sig[i - 2])
            r0, r1 = h0 / h, h1 / h
            d10, d11 = (1.0 / r0) * (ms[-1] - ms[-2]), (1.0 / r1) * (ms[-2] - ms[-3])
            d1 = d10 + (r0 / (r0 + r1)) * (d10 - d11)
            d2 = (1.0 / (r0 + r1)) * (d10 - d11)
            x = (st / s0) * x - (at * e) * ms[-1] + (at * (e / h + 1.0)) * d1 - (at * ((e + h) / h ** 2 - 0.5)) * d2
        if lower < order:
            lower += 1
        traj.append(x)
    return traj
