"""The body of one 4D-STraG training step (scripts/4D_STraG_training/train_wan.py:1891-2015) as functions over the
HIP-backed model / optimizer: flow-matching noising (:1924-1929), thresholded MSE (:1953-1966), the adaptive
gradient-norm clip (:1991-2009) and the AdamW update (:2014).  Host logic only; the arithmetic on model-sized tensors
runs in the kernels (autograd.py, optim.py); the [B,16,13,60,104]-sized loss tensors stay in torch like in the reference."""
import torch

from .optim import AdamW, clip_grad_norm_, grad_norm


def linear_decay(initial_value, final_value, total_steps, current_step):
    """train_wan.py:76-82."""
    if current_step >= total_steps:
        return final_value
    current_step = max(0, current_step)
    return initial_value + (final_value - initial_value) / total_steps * current_step


def add_noise(latents, noise, sigmas):
    """zt = (1 - sigma) x + sigma z1 and the velocity target z1 - x (:1924-1929); sigmas broadcast over [B,1,1,1,1]."""
    sigmas = sigmas.reshape(-1, *([1] * (latents.dim() - 1))).to(latents.dtype)
    return (1.0 - sigmas) * latents + sigmas * noise, noise - latents


def custom_mse_loss(noise_pred, target, weighting=None, threshold=50):
    """:1953-1963 — elements whose error exceeds `threshold` are masked out of the mean."""
    noise_pred, target = noise_pred.float(), target.float()
    diff = noise_pred - target
    loss = diff * diff * (diff.abs() <= threshold).float()
    if weighting is not None:
        loss = loss * weighting
    return loss.mean()


def adaptive_max_grad_norm(total_norm, max_grad_norm, initial_grad_norm_ratio, abnormal_norm_clip_start, global_step):
    """:1995-1999 — the clip threshold decays linearly from ratio*max to max; a norm more than 5x over it after the
    warm-up tightens the threshold by min(norm/max, 10)."""
    mx = linear_decay(max_grad_norm * initial_grad_norm_ratio, max_grad_norm, abnormal_norm_clip_start, global_step)
    if total_norm / mx > 5 and global_step > abnormal_norm_clip_start:
        return mx / min(total_norm / mx, 10)
    return mx


def train_step(model, optimizer, *, latents, noise, sigmas, timesteps, forward_kwargs, global_step=0, max_grad_norm=0.05,
               initial_grad_norm_ratio=5.0, abnormal_norm_clip_start=1000, motion_sub_loss_ratio=None, params=None,
               abnormal_loss=0.25, abnormal_loss_start=50):
    """One optimisation step.  `model` may be the bare WanTransformer4DModel or its DDP wrapper; returns
    (loss, total_grad_norm, actual_max_grad_norm).  Host syncs: the abnormal-loss check (:1977-1985: after step 50 an
    update whose process-averaged loss exceeds 0.25 is skipped — returned norms are None) and the adaptive-clip decision
    (:1996).  The loss weighting of :1964 is diffusers' compute_loss_weighting_for_sd3; the released recipe leaves it at
    its default scheme ("none" = 1)."""
    params = list(params) if params is not None else [p for p in model.parameters() if p.requires_grad]
    noisy, target = add_noise(latents, noise, sigmas)
    pred = model(x=noisy.to(next(iter(params)).dtype), t=timesteps, **forward_kwargs)
    loss = custom_mse_loss(pred, target)
    if motion_sub_loss_ratio and pred.size(1) > 2:                       # :1968-1971
        sub = torch.nn.functional.mse_loss(pred[:, 1:].float() - pred[:, :-1].float(),
                                           target[:, 1:].float() - target[:, :-1].float())
        loss = loss * (1 - motion_sub_loss_ratio) + sub * motion_sub_loss_ratio
    if abnormal_loss is not None and global_step > abnormal_loss_start:
        avg = loss.detach().clone()
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(avg)
            avg /= torch.distributed.get_world_size()
        if float(avg) > abnormal_loss:
            optimizer.zero_grad(set_to_none=True)
            return loss.detach(), None, None
    loss.backward()
    total_dev = grad_norm(params)          # ONE sum-of-squares pass over the gradients; reused by the clip below
    total = float(total_dev)
    actual = adaptive_max_grad_norm(total, max_grad_norm, initial_grad_norm_ratio, abnormal_norm_clip_start, global_step)
    clip_grad_norm_(params, actual, optimizer=optimizer if isinstance(optimizer, AdamW) else None, total_norm=total_dev)
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach(), total, actual
