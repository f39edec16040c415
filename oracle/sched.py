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
