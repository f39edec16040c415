"""Oracle: one Motion-Sensitive-VAE fine-tuning step (scripts/4D_STraG_training/train_vae.py:434-495, loss :173-187) under torch
autograd over the functional restatement of oracle/vae.py.  TEST INFRASTRUCTURE — see oracle/__init__.py.

The training-time twins of encode / decode (wan_vae.py: encode_full :549-613, decode_full :633-676) compute the same forward
values but CUT THE GRAPH at every chunk boundary: the streaming cache is detached after each slice (`_detach_cache`, :616-621,
called at :596 / :674), so a chunk's gradient reaches its own input slice and the parameters, never the frames of earlier
chunks.  Pinned to gradients produced by the reference itself (tests/golden/vae_train.npz, make_golden.py:make_vae_train).
"""
import torch

from . import vae as ov


def _detach(st):
    for k, v in st.tails.items():
        if isinstance(v, torch.Tensor):
            st.tails[k] = v.detach()


def vae_encode_full(sd, x, z_dim=16):
    """encode_full (:549-613): chunks 1 + 4 + 4 ..., cache detached after each, conv1, mu normalised."""
    st = ov.Stream()
    t = x.shape[2]
    outs = []
    for i in range(1 + (t - 1) // 4):
        chunk = x[:, :, :1] if i == 0 else x[:, :, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(ov.encoder3d(sd, "model.encoder", chunk, st, ov.ENC_MODES))
        _detach(st)
    out = torch.cat(outs, 2)
    mu, logvar = ov.causal_conv3d(sd, "model.conv1", out).chunk(2, dim=1)
    mean = torch.tensor(ov.MEAN[:z_dim], dtype=x.dtype).view(1, z_dim, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(ov.STD[:z_dim], dtype=x.dtype)).view(1, z_dim, 1, 1, 1)
    return torch.cat([(mu - mean) * inv_std, logvar], dim=1)


def vae_decode_full(sd, z, z_dim=16):
    """decode_full (:633-676) + the wrapper's clamp_(-1, 1) (:815-818)."""
    st = ov.Stream()
    mean = torch.tensor(ov.MEAN[:z_dim], dtype=z.dtype).view(1, z_dim, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(ov.STD[:z_dim], dtype=z.dtype)).view(1, z_dim, 1, 1, 1)
    x = ov.causal_conv3d(sd, "model.conv2", z / inv_std + mean)
    outs = []
    for i in range(x.shape[2]):
        outs.append(ov.decoder3d(sd, "model.decoder", x[:, :, i:i + 1], st, ov.DEC_MODES))
        _detach(st)
    return torch.cat(outs, 2).clamp(-1, 1)


def kl_standard_normal(params):
    """diffusers DiagonalGaussianDistribution.kl() (third-party, restated — unpinned): 0.5 sum(mu^2 + var - 1 - logvar) over
    dims [1, 2, 3] with logvar clamped to [-30, 20]."""
    mean, logvar = params.chunk(2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    return 0.5 * torch.sum(mean.pow(2) + logvar.exp() - 1.0 - logvar, dim=[1, 2, 3])


def train_step_loss(sd_vae, sd_enc, sd_dec, targets, eps, grad_through_encoder=False, kl_scale=1e-6):
    """train_vae.py:434-458 + :173-187 (rec_loss 'l1'): returns (loss, nll, kl, dict of forward values)."""
    pseudo = ov.encoder_adaptor(sd_enc, targets) * 2 - 1
    if grad_through_encoder:
        params = vae_encode_full(sd_vae, pseudo)
    else:
        with torch.no_grad():                       # :444-448
            params = vae_encode_full(sd_vae, pseudo)
    latents = ov.gaussian_sample(params, eps)
    recon = vae_decode_full(sd_vae, latents)
    rec2 = ov.decoder_adaptor(sd_dec, recon)
    rec_loss = (rec2.float() - targets.float()).abs()
    nll = rec_loss.sum() / rec_loss.shape[0]
    klv = kl_standard_normal(params)
    kl = klv.sum() / klv.shape[0]
    return nll + kl_scale * kl, nll, kl, dict(pseudo=pseudo, params=params, latents=latents, recon=recon, reconstructions=rec2)
