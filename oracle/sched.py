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


def unipc_loop(model_fn, x, sigmas, timesteps, order):
    """FlowUniPCMultistepScheduler (predict_x0, solver_type bh2, lower_order_final, no disabled correctors), restated from
    MoRe4D/utils/fm_solvers_unipc.py: step :655-739 (corrector first, then order selection, then predictor),
    multistep_uni_p_bh_update :350-484, multistep_uni_c_bh_update :486-626.  float64 coefficients.
    Returns the list of samples after every step."""
    import math
    n = len(timesteps)
    sig = [float(s) for s in sigmas]

    def lam(s):
        if s <= 0.0:
            return math.inf
        return -math.inf if s >= 1.0 else math.log(1.0 - s) - math.log(s)

    def coeffs(h, rks, n_rhos, simplified_half):
        """(h_phi_1, B_h, rhos) for the bh2 update with `rks` (last one 1.0)."""
        hh = -h
        h_phi_1 = math.expm1(hh) if math.isfinite(hh) else -1.0
        B_h = h_phi_1
        K = len(rks)
        h_phi_k = (h_phi_1 / hh - 1.0) if math.isfinite(hh) else -1.0
        fact, R, b = 1, [], []
        for i in range(1, K + 1):
            R.append([rk ** (i - 1) for rk in rks])
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = (h_phi_k / hh - 1.0 / fact) if math.isfinite(hh) else -1.0 / fact
        if simplified_half:
            return h_phi_1, B_h, [0.5]
        R, b = np.array(R, dtype=np.float64)[:n_rhos, :n_rhos], np.array(b, dtype=np.float64)[:n_rhos]
        return h_phi_1, B_h, list(np.linalg.solve(R, b)) if n_rhos else []

    ms, lower, last_sample, this_order, traj = [], 0, None, 1, []
    x = x.float()
    for i in range(n):
        m_t = x - sig[i] * model_fn(x, timesteps[i])                       # convert_model_output (x0 prediction)
        if i > 0 and last_sample is not None:                              # ---- corrector (UniC), order = this_order
            st, s0 = sig[i], sig[i - 1]
            h = lam(st) - lam(s0)
            m0 = ms[-1]
            rks, d1s = [], []
            for j in range(1, this_order):
                rk = (lam(sig[i - (j + 1)]) - lam(s0)) / h
                rks.append(rk)
                d1s.append((ms[-(j + 1)] - m0) / rk)
            rks.append(1.0)
            h_phi_1, B_h, rhos = coeffs(h, rks, len(rks), this_order == 1)
            corr = sum(r * d for r, d in zip(rhos[:-1], d1s)) if d1s else 0
            x = (st / s0) * last_sample - ((1.0 - st) * h_phi_1) * m0 - ((1.0 - st) * B_h) * (corr + rhos[-1] * (m_t - m0))
        ms = (ms + [m_t])[-order:]
        this_order = min(order, n - i, lower + 1)                          # lower_order_final + warm-up
        last_sample = x
        st, s0 = sig[i + 1], sig[i]                                        # ---- predictor (UniP)
        h = lam(st) - lam(s0)
        m0 = ms[-1]
        rks, d1s = [], []
        for j in range(1, this_order):
            rk = (lam(sig[i - j]) - lam(s0)) / h
            rks.append(rk)
            d1s.append((ms[-(j + 1)] - m0) / rk)
        rks.append(1.0)
        h_phi_1, B_h, rhos = coeffs(h, rks, len(rks) - 1, this_order == 2)
        pred = sum(r * d for r, d in zip(rhos, d1s)) if d1s else 0
        x = (st / s0) * x - ((1.0 - st) * h_phi_1) * m0 - ((1.0 - st) * B_h) * pred
        if lower < order:
            lower += 1
        traj.append(x)
    return traj
