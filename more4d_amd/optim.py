"""Optimizer side of the training step (train_wan.py:1136-1142 `torch.optim.AdamW(bf16 params, betas 0.9/0.999,
weight_decay 3e-2, eps 1e-10)`, :1991-2014 gradient-norm / clip / step).  One fused HIP kernel per parameter does
clip-scale + weight decay + moment update + parameter update (m4d_adamw); the global gradient norm is a chain of
m4d_sumsq launches into one device scalar, so a step never synchronises with the host."""
import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction).  state_dtype=None keeps the moments in
    the parameter dtype like the reference's plain AdamW on bf16 parameters; torch.float32 keeps fp32 moments
    (14B: 112 GB instead of 56 GB — both fit the 288 GB of one MI355X)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, state_dtype=None):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("AdamW: invalid hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.state_dtype = state_dtype
        self._grad_scale = None

    def set_grad_scale(self, scale):
        """Device scalar multiplied into every gradient by the next step() (the clip coefficient), then cleared."""
        self._grad_scale = scale

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    sd = self.state_dtype or p.dtype
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, dtype=sd, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=sd, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                lr = group["lr"]
                ops.adamw_(p.data, g, st["exp_avg"], st["exp_avg_sq"], lr=float(lr), beta1=b1, beta2=b2, eps=group["eps"],
                           weight_decay=group["weight_decay"], step=st["step"], grad_scale=self._grad_scale)
                # the kernel wrote through the raw pointer: tell torch, so that every (data_ptr, _version)-keyed cache of
                # this parameter (fp32 copies, packed conv weights) is rebuilt and autograd sees the in-place update
                torch.autograd.graph.increment_version(p)
        self._grad_scale = None
        return loss


def grad_norm(parameters):
    """Global L2 norm of the gradients as a 0-d device tensor (torch.norm(stack(norm(g))) of :1991-1993)."""
    acc = None
    for p in parameters:
        if p.grad is None:
            continue
        if acc is None:
            acc = torch.zeros((), device=p.grad.device, dtype=torch.float32)
        ops.sumsq(p.grad if p.grad.is_contiguous() else p.grad.contiguous(), acc)
    if acc is None:
        return torch.zeros(())
    return acc.sqrt()


def clip_grad_norm_(parameters, max_norm, optimizer=None, total_norm=None):
    """accelerator.clip_grad_norm_ (:2009).  With a more4d_amd AdamW the coefficient is fused into the next step()
    (no extra pass over 33 GB of gradients); otherwise the gradients are scaled in place.  `total_norm`: the global norm
    if the caller already has it (0-d device tensor from grad_norm) — the sum-of-squares pass is not repeated.
    Returns the total norm."""
    parameters = [p for p in parameters if p.grad is not None]
    total = grad_norm(parameters) if total_norm is None else total_norm
    coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
    if isinstance(optimizer, AdamW):
        optimizer.set_grad_scale(coef.to(torch.float32).contiguous())
    else:
        for p in parameters:
            p.grad.mul_(coef.to(p.grad.dtype))
    return total
