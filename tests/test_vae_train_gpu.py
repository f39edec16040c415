"""VAE / adaptor TRAINING path on the MI355X (SURVEY §8 v7 twins, train_vae.py:434-495): every backward kernel against fp32 torch
autograd of the same op, the conv data / weight gradients against autograd through F.conv3d, and one whole train_vae.py step
against the gradients produced by the REFERENCE itself (tests/golden/vae_train.npz) — fp32 mode at north_star's 1e-3, bf16
(production dtype) within a stated budget."""
import math

import pytest
import torch
import torch.nn.functional as F

import cpu_ops
from util import grad_sample, load_keys, load_npz, rel_err, rms_rel_err
from weights import fill

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_pad_transpose_and_batched_gemm(dtype):
    from more4d_amd import ops
    g = gen(1)
    T_, H, W, C = 3, 5, 7, 12
    ps = 16
    src = torch.randn(T_, H, W, ps, generator=g).to(dtype)
    Hp, Wp = H + 2, 16
    for (pt, pl, ns, cols) in ((1, 1, 3, 3 * Hp * Wp + 40), (0, 0, 1, 2 * Hp * Wp)):
        want = cpu_ops.pad_transpose(src, ps, C, T_, H, W, Hp, Wp, pt, pl, ns, cols)
        got = ops.pad_transpose(src.to(DEV), ps, C, T_, H, W, Hp, Wp, pt, pl, ns, cols)
        assert torch.equal(got.cpu(), want)
    a = torch.randn(20, 512, generator=g).to(dtype)
    w = torch.randn(24, 640, generator=g).to(dtype)
    kw = dict(M=20, N=24, K=128, nb1=3, a_bs1=128, w_bs1=128, nb2=2, a_bs2=0, w_bs2=64)
    want = cpu_ops.gemm_bt_batched(a, w, **kw)
    got = ops.gemm_bt_batched(a.to(DEV), w.to(DEV), **kw)
    assert rel_err(got.cpu(), want) < (1e-5 if dtype == torch.float32 else 1e-5)     # float32 accumulators, unrounded
    part = torch.randn(3, 5, 8, 2 * 16, generator=g)
    dw = torch.randn(8, 2, 3, 2, 16, generator=g)
    want = cpu_ops.wgrad_reduce(part.clone(), dw.clone(), 1, 8)
    got = ops.wgrad_reduce(part.to(DEV), dw.to(DEV), 1, 8)
    assert rel_err(got.cpu(), want) < 1e-6


@pytest.mark.parametrize("dtype,C", [(torch.float32, 96), (BF, 96), (BF, 384), (torch.float32, 16), (BF, 192)])
@pytest.mark.parametrize("silu", [True, False])
def test_rmsnorm_silu_bwd(dtype, C, silu):
    from more4d_amd import ops
    g = gen(2)
    P = 1000
    x = torch.randn(P, C, generator=g).to(dtype)
    dy = torch.randn(P, C, generator=g).to(dtype)
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    wdx, wdg = cpu_ops.rmsnorm_silu_cl_bwd(x.float(), gamma, dy.float(), silu=silu)
    dx, dg = ops.rmsnorm_silu_cl_bwd(x.to(DEV), gamma.to(DEV), dy.to(DEV), silu=silu)
    tol = 1e-5 if dtype == torch.float32 else 8e-3
    assert rel_err(dx.float().cpu(), wdx) < tol and rel_err(dg.cpu(), wdg) < (1e-4 if dtype == torch.float32 else 8e-3)
    # strided rows (a staging-buffer view) give the same result
    xb = torch.zeros(P, C + 8).to(dtype)
    xb[:, :C] = x
    dx2, dg2 = ops.rmsnorm_silu_cl_bwd(xb.to(DEV)[:, :C], gamma.to(DEV), dy.to(DEV), silu=silu)
    assert torch.equal(dx2, dx)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_softmax_bwd_upsample_groupnorm_act(dtype):
    from more4d_amd import ops
    g = gen(3)
    R, C, Cp = 40, 37, 40
    p = torch.zeros(R, Cp)
    p[:, :C] = torch.softmax(torch.randn(R, C, generator=g), -1)
    p = p.to(dtype)
    dp = torch.randn(R, Cp, generator=g)
    want = cpu_ops.softmax_rows_bwd(p, dp, scale=0.3, C=C)
    got = ops.softmax_rows_bwd(p.to(DEV), dp.to(DEV), scale=0.3, C=C)
    assert rel_err(got.float().cpu(), want.float()) < (1e-6 if dtype == torch.float32 else 8e-3)
    for ts in (False, True):
        t, h, w, c = 2, 3, 5, 16
        x = torch.randn(t * h * w, c * (2 if ts else 1), generator=g).to(dtype)
        up = ops.upsample2x_cl(x.to(DEV), t, h, w, c, tsplit=ts)
        assert torch.equal(up.cpu(), cpu_ops.upsample2x_cl(x, t, h, w, c, tsplit=ts))
        du = torch.randn(up.shape, generator=g).to(dtype)
        got = ops.upsample2x_cl_bwd(du.to(DEV), t, h, w, c, tsplit=ts)
        assert rel_err(got.float().cpu(), cpu_ops.upsample2x_cl_bwd(du, t, h, w, c, tsplit=ts).float()) < (1e-6 if dtype == torch.float32 else 8e-3)
    Fr, HW, C = 3, 5000, 128
    x = (torch.randn(Fr, HW, C, generator=g) * 1.5 + 0.3).to(dtype)
    dy = torch.randn(Fr, HW, C, generator=g).to(dtype)
    wt, bs = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    for silu in (True, False):
        wdx, wdw, wdb = cpu_ops.groupnorm_cl_bwd(x.float(), wt, bs, dy.float(), F=Fr, HW=HW, silu=silu)
        dx, dwt, dbs = ops.groupnorm_cl_bwd(x.to(DEV), wt.to(DEV), bs.to(DEV), dy.to(DEV), F=Fr, HW=HW, silu=silu)
        tol = 2e-5 if dtype == torch.float32 else 1e-2
        assert rel_err(dx.float().cpu(), wdx) < tol and rel_err(dwt.cpu(), wdw) < tol and rel_err(dbs.cpu(), wdb) < tol
    n = 4096
    pre = torch.randn(n, generator=g).to(dtype) * 2
    for act in (4, 5, 6):
        src = torch.sigmoid(pre.float()).to(dtype) if act == 5 else pre
        d0 = torch.randn(n, generator=g).to(dtype)
        want = cpu_ops.act_bwd_(d0.clone(), src, act)
        got = ops.act_bwd_(d0.to(DEV).clone(), src.to(DEV), act)
        assert rel_err(got.float().cpu(), want.float()) < (1e-6 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("shape", [(3, 6, 10, 16, 24, (3, 3, 3)), (1, 9, 13, 8, 4, (1, 3, 3)), (2, 4, 6, 32, 64, (3, 1, 1)), (4, 20, 36, 96, 96, (3, 3, 3))])
def test_conv_gradients_vs_autograd(dtype, shape):
    """conv_dgrad / conv_wgrad (flipped-tap conv; pad_transpose + batched split-K GEMM + reduce) against torch autograd through
    F.conv3d on the same causal stride-1 geometry: input buffer [tail + chunk] frames, gradient w.r.t. the chunk frames only."""
    from more4d_amd.vae_autograd import conv_dgrad, conv_wgrad
    t, h, w, ci, co, k = shape
    kt, kh, kw = k
    g = gen(4)
    Tin = t + kt - 1
    x = torch.randn(Tin, h, w, ci, generator=g).to(dtype)
    wt = (torch.randn(co, kt, kh, kw, ci, generator=g) / math.sqrt(kt * kh * kw * ci)).to(dtype)
    dy = torch.randn(t * h * w, co, generator=g).to(dtype)
    xr = x.float().permute(3, 0, 1, 2)[None].requires_grad_(True)
    wr = wt.float().permute(0, 4, 1, 2, 3).requires_grad_(True)
    y = F.conv3d(F.pad(xr, (kw // 2, kw // 2, kh // 2, kh // 2, 0, 0)), wr)
    y.backward(dy.float().view(t, h, w, co).permute(3, 0, 1, 2)[None])
    want_dx = xr.grad[0].permute(1, 2, 3, 0)[kt - 1:].reshape(t * h * w, ci)
    want_dw = wr.grad.permute(0, 2, 3, 4, 1)
    dx = conv_dgrad(dy.to(DEV), wt.view(co, -1).to(DEV), co, ci, k, t, h, w)
    dw = conv_wgrad(x.to(DEV), ci, Tin, h, w, ci, dy.to(DEV), co, k, (kh // 2, kw // 2))
    tol = 1e-4 if dtype == torch.float32 else 1.5e-2
    assert rel_err(dx.float().cpu(), want_dx) < tol
    assert rel_err(dw.cpu(), want_dw) < tol


def _models(dtype):
    from more4d_amd.models.trajectory_module import VAEDecoderadaptor, VAEEncoderadaptor
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    vae = AutoencoderKLWan()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    ea, da = VAEEncoderadaptor(), VAEDecoderadaptor()
    ea.load_state_dict(fill(load_keys("adaptor_enc_keys.json"), 78))
    da.load_state_dict(fill(load_keys("adaptor_dec_keys.json"), 77))
    vae, ea, da = vae.to(DEV, dtype), ea.to(DEV, dtype), da.to(DEV, dtype)
    ea.requires_grad_(True).train()
    da.requires_grad_(True).train()
    vae.model.encoder.requires_grad_(False).eval()
    vae.model.conv1.requires_grad_(False)
    vae.model.decoder.requires_grad_(True).train()
    return vae, ea, da


def _train_step(tag, dtype):
    z = load_npz("vae_train.npz")
    vae, ea, da = _models(dtype)
    targets = z["targets"].to(DEV, dtype)
    pseudo = ea(targets) * 2 - 1
    if tag == "A":
        with torch.no_grad():
            posterior = vae.encode_memory_saver(pseudo).latent_dist
    else:
        posterior = vae.encode_memory_saver(pseudo).latent_dist
    latents = posterior.mean + posterior.std * z[f"{tag}/eps"].to(DEV, dtype)
    recon = vae.decode_memory_saver(latents).sample
    rec2 = da(recon)
    rec_loss = (rec2.float() - targets.float()).abs()
    nll = rec_loss.sum() / rec_loss.shape[0]
    kl = posterior.kl().sum() / posterior.kl().shape[0]
    loss = nll + 1e-6 * kl
    loss.backward()
    named = {}
    for pre, mod in (("encoder_prompt.", ea), ("decoder_prompt.", da), ("vae.", vae)):
        for n, p in mod.named_parameters():
            named[pre + n] = p
    return z, dict(pseudo=pseudo, params=posterior.parameters, recon=recon, reconstructions=rec2, loss=loss), named, ea


@pytest.mark.parametrize("tag", ["A", "B"])
def test_train_vae_step_fp32_vs_reference(tag):
    """One train_vae.py step through the HIP kernels == the reference's step: forward values, loss and every parameter gradient
    (A: as written — decoder + decoder prompt; B: gradient through the frozen encoder — encoder prompt and the KL path too)."""
    z, fwd, named, ea = _train_step(tag, torch.float32)
    for k in ("pseudo", "params", "recon", "reconstructions"):
        assert rel_err(fwd[k].detach().float().cpu(), z[f"{tag}/{k}"]) < 1e-3, k
    assert abs(float(fwd["loss"].detach()) - float(z[f"{tag}/loss"])) < 1e-3 * float(z[f"{tag}/loss"])
    names = [k[len(tag) + 6:] for k in z if k.startswith(f"{tag}/grad/") and not k.startswith(f"{tag}/grad/vae.model.conv1")]
    assert len(names) > 100
    gmax = max(float(z[f"{tag}/grad/{n}"].abs().max()) for n in names)
    worst = ("", 0.0)
    for n in names:
        g = named[n].grad
        assert g is not None, n
        ref = z[f"{tag}/grad/{n}"]
        e = float((grad_sample(g.float().cpu()).double() - ref.double()).abs().max() / max(float(ref.abs().max()), 1e-3 * gmax))
        ne = abs(float(g.float().norm()) - float(z[f"{tag}/norm/{n}"])) / max(float(z[f"{tag}/norm/{n}"]), 1e-3 * gmax)
        worst = max(worst, (n, max(e, ne)), key=lambda u: u[1])
        assert e < 1e-3 and ne < 1e-3, (n, e, ne)
    print("worst gradient error vs the reference", worst)
    if tag == "A":
        assert all(p.grad is None for p in ea.parameters())        # encode runs under no_grad in train_vae.py:444-448


def test_train_vae_step_bf16_budget():
    """Production dtype: same step in bf16.  Gradients are compared as directions (cosine) and norms against the reference's fp32
    gradients: bf16 activations through a 30-conv decoder give percent-level element errors, the update direction must agree."""
    z, fwd, named, _ = _train_step("B", BF)
    assert rms_rel_err(fwd["recon"].detach().float().cpu(), z["B/recon"]) < 8e-2
    names = [k[7:] for k in z if k.startswith("B/grad/") and not k.startswith("B/grad/vae.model.conv1")]
    cos = []
    for n in names:
        g = grad_sample(named[n].grad.float().cpu()).double()
        ref = z[f"B/grad/{n}"].double()
        assert bool(torch.isfinite(g).all()), n
        if float(ref.norm()) > 1e-3 * max(float(z[f"B/norm/{m}"]) for m in names) and ref.numel() >= 64:
            cos.append(float((g * ref).sum() / (g.norm() * ref.norm()).clamp_min(1e-30)))
    assert len(cos) > 50 and min(cos) > 0.9 and sum(cos) / len(cos) > 0.98, (min(cos), sum(cos) / len(cos))


def test_train_vae_optimizer_step_moves_parameters():
    """clip + AdamW over the trainable set of train_vae.py (:469-481) with the product optimizer: the loss of a second step on the
    same batch goes down."""
    from more4d_amd.optim import AdamW, clip_grad_norm_
    z = load_npz("vae_train.npz")
    vae, ea, da = _models(torch.float32)
    params = list(da.parameters()) + list(vae.model.decoder.parameters())
    opt = AdamW(params, lr=2e-4, weight_decay=1e-2)
    targets = z["targets"].to(DEV)
    losses = []
    for _ in range(3):
        with torch.no_grad():
            pseudo = ea(targets) * 2 - 1
            lat = vae.encode_memory_saver(pseudo).latent_dist.mode()
        rec2 = da(vae.decode_memory_saver(lat).sample)
        loss = (rec2.float() - targets).abs().sum()
        loss.backward()
        clip_grad_norm_(params, 1.0, optimizer=opt)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    assert losses[2] < losses[0], losses


def test_train_vae_step_with_trainable_encoder_fp32_vs_reference():
    """The same step with the ENCODER trainable (train_vae.py:355 freezes it; round-4 review "missing" #4): gradients of every encoder
    parameter against the reference's — including the Resample layers' ZeroPad2d + stride-2 Conv2d and the strided (3, 1, 1) time
    conv (wan_vae.py:96-110), whose weight gradients are the stride-1 weight gradients against dy scattered onto a zero map."""
    z, ze = load_npz("vae_train.npz"), load_npz("vae_train_enc.npz")
    vae, ea, da = _models(torch.float32)
    vae.model.encoder.requires_grad_(True)
    targets = z["targets"].to(DEV)
    pseudo = ea(targets) * 2 - 1
    posterior = vae.encode_memory_saver(pseudo).latent_dist
    latents = posterior.mean + posterior.std * z["B/eps"].to(DEV)
    rec2 = da(vae.decode_memory_saver(latents).sample)
    rec_loss = (rec2.float() - targets.float()).abs()
    loss = rec_loss.sum() / rec_loss.shape[0] + 1e-6 * posterior.kl().sum() / posterior.kl().shape[0]
    assert abs(float(loss.detach()) - float(ze["loss"])) < 1e-3 * float(ze["loss"])
    loss.backward()
    names = [k[5:] for k in ze if k.startswith("grad/")]
    assert len(names) > 60 and any("time_conv" in n for n in names) and any("resample.1" in n for n in names)
    gmax = max(float(ze["grad/" + n].abs().max()) for n in names)
    named = {"vae." + n: p for n, p in vae.named_parameters()}
    worst = ("", 0.0)
    for n in names:
        g = named[n].grad
        assert g is not None, n
        ref = ze["grad/" + n]
        e = float((grad_sample(g.float().cpu()).double() - ref.double()).abs().max() / max(float(ref.abs().max()), 1e-3 * gmax))
        ne = abs(float(g.float().norm()) - float(ze["norm/" + n])) / max(float(ze["norm/" + n]), 1e-3 * gmax)
        worst = max(worst, (n, max(e, ne)), key=lambda u: u[1])
        assert e < 1e-3 and ne < 1e-3, (n, e, ne)
    print("worst encoder gradient error vs the reference", worst)
