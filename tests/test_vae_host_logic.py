"""CPU: host logic of the VAE / adaptor runners (streaming tails, weight packing, channel padding, layout
boundaries, first-chunk special cases) against the reference fixtures, with the kernels replaced by tests/cpu_ops.py."""
import pytest
import torch

import cpu_ops
from util import load_keys, load_npz, rel_err
from weights import fill


def sd_of(z):
    return {k[3:]: v for k, v in z.items() if k.startswith("sd.")}


def test_vae_state_dict_contract():
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    keys = load_keys("vae_keys.json")
    sd = AutoencoderKLWan().state_dict()
    assert set(sd) == set(keys), set(sd) ^ set(keys)
    assert all(tuple(sd[k].shape) == keys[k] for k in keys)


def test_vae_roundtrip_host_logic(monkeypatch):
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    cpu_ops.install(monkeypatch)
    z = load_npz("vae_roundtrip.npz")
    vae = AutoencoderKLWan().eval()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    with torch.no_grad():
        enc = vae._encode(z["x"])
        assert rel_err(enc, z["enc"]) < 1e-4
        d = vae.encode(z["x"])[0]
        assert torch.equal(d.mode(), enc[:, :16])
        dec = vae.decode(z["enc"][:, :16]).sample
        assert rel_err(dec, z["dec"]) < 1e-4
        # the inner model's own entry points (reference wan_vae.py:520, :549, :633, :678, :717), called the way the reference's
        # wrapper calls them (:770-776, :825-832): same values; decode / decode_full are NOT clamped
        m = vae.model
        assert rel_err(m.encode(z["x"], vae.scale), enc) < 1e-5 and torch.equal(m.encode_full(z["x"], vae.scale), m.encode(z["x"], vae.scale))
        raw = m.decode(z["enc"][:, :16], vae.scale)
        assert rel_err(raw.clamp(-1, 1), dec) < 1e-5 and torch.equal(m.decode_full(z["enc"][:, :16], vae.scale), raw)     # (std = 1 / (1 / std))
        assert float(raw.abs().max()) > 1.0                      # (this fixture does leave [-1, 1] before the clamp)
        # another normalisation than the wrapper's: two floats (the reference accepts both forms, :539-545)
        mu = m.encode(z["x"], [0.5, 2.0])[:, :16]
        assert rel_err(mu, (enc[:, :16] / vae.scale[1].view(1, -1, 1, 1, 1) + vae.scale[0].view(1, -1, 1, 1, 1) - 0.5) * 2.0) < 1e-5
        m.clear_cache()
        assert m._conv_idx == [0] and len(m._feat_map) == m._conv_num > 0 and len(m._enc_feat_map) == m._enc_conv_num > 0


def test_vae_planar_staging_host_logic(monkeypatch):
    """The planar-16 staging buffers (RMS-norm writes [C/16, frames, h*w, 16], the LDS-halo conv reads it; 4-chunk ring with wrap-around
    and growth, tails handed over in place) forced on at the fixture's tiny maps: same round trip as the reference."""
    from more4d_amd.models import wan_vae
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    cpu_ops.install(monkeypatch)
    monkeypatch.setattr(wan_vae._Runner, "PLANAR_MIN_PIXELS", 0)
    monkeypatch.setattr(wan_vae._Runner, "PLANAR_DTYPES", (torch.float32, torch.bfloat16))
    planar_stages = []
    orig = wan_vae._Stage.__init__

    def spy(self, *a, **kw):
        orig(self, *a, **kw)
        planar_stages.append(self.planar)
    monkeypatch.setattr(wan_vae._Stage, "__init__", spy)
    z = load_npz("vae_roundtrip.npz")
    vae = AutoencoderKLWan().eval()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    with torch.no_grad():
        assert rel_err(vae._encode(z["x"]), z["enc"]) < 1e-4
        assert rel_err(vae.decode(z["enc"][:, :16]).sample, z["dec"]) < 1e-4
    assert sum(planar_stages) > 20 and not all(planar_stages)      # residual-block convs planar, conv1 / time_conv stages not
    # the ring itself: tails stay in place, wrap-around copies them to the front, growth keeps them
    st = wan_vae._Stage(2, 2, 2, 2, 32, torch.float32, "cpu", ring=2, planar=True)
    frames = []
    for i, t in enumerate([1, 2, 2, 1, 4, 2]):
        c = torch.arange(t * 4 * 32, dtype=torch.float32).view(t, 4, 32) + 1000 * i
        st.chunk(t).t.copy_(c.view(t, 4, 2, 16).permute(2, 0, 1, 3))
        frames += list(c)
        win = st.window(t).t.permute(1, 2, 0, 3).reshape(2 + t, 4, 32)
        want = torch.stack(([torch.zeros(4, 32)] * 2 + frames)[-(2 + t):])
        assert torch.equal(win, want), i
        st.roll(t)
    # a planar buffer must stay below the 2 GiB the conv kernel can address: fewer chunks per buffer, then channels-last
    monkeypatch.setattr(wan_vae._Stage, "PLANAR_MAX_BYTES", 2 * (2 + 2 * 3) * 4 * 32)      # room for tail + 2 chunks of 3 frames
    st = wan_vae._Stage(2, 3, 2, 2, 32, torch.float32, "cpu", ring=4, planar=True)
    assert st.planar and st.buf.shape == (2, 2 + 2 * 3, 4, 16)
    tail = torch.arange(2 * 4 * 32, dtype=torch.float32).view(2, 4, 32)
    st.tail().copy_(tail.view(2, 4, 2, 16).permute(2, 0, 1, 3))
    v = st.chunk(8)                                                                        # grows past the limit: channels-last from now on
    assert not st.planar and not isinstance(v, ops_mod().Planar16) and v.shape == (8 * 4, 32)
    assert torch.equal(st.tail(), tail)


def ops_mod():
    import more4d_amd.ops as o
    return o


@pytest.mark.parametrize("planar", [False, True])
def test_adaptors_host_logic(monkeypatch, planar):
    from more4d_amd.models import trajectory_module
    from more4d_amd.models.trajectory_module import VAEDecoderadaptor, VAEEncoderadaptor
    cpu_ops.install(monkeypatch)
    if planar:        # GroupNorm -> planar-16 frame groups -> conv, forced on at the fixture's size with 2-frame groups
        monkeypatch.setattr(trajectory_module._AdaptorBase, "PLANAR_MIN_PIXELS", 0)
        monkeypatch.setattr(trajectory_module._AdaptorBase, "PLANAR_MAX_BYTES", 1)
        monkeypatch.setattr(trajectory_module._AdaptorBase, "PLANAR_DTYPES", (torch.float32, torch.bfloat16))
    for cls, name in ((VAEEncoderadaptor, "adaptor_enc.npz"), (VAEDecoderadaptor, "adaptor_dec.npz")):
        z = load_npz(name)
        m = cls().eval()
        m.load_state_dict(sd_of(z), strict=True)
        with torch.no_grad():
            out = m(z["x"])
        assert rel_err(out, z["out"]) < 2e-5, name


def test_vae_from_pretrained_checkpoint_formats(tmp_path):
    """AutoencoderKLWan.from_pretrained (reference wan_vae.py:849-871): the Wan2.1_VAE.pth layout (keys WITHOUT the `model.`
    prefix the wrapper adds, :864-868), as .pth and as .safetensors."""
    from safetensors.torch import save_file
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    sd = fill(load_keys("vae_keys.json"), 2024)
    raw = {k[len("model."):]: v for k, v in sd.items()}
    assert len(raw) == len(sd) == 194
    torch.save(raw, tmp_path / "Wan2.1_VAE.pth")
    save_file({k: v.contiguous() for k, v in raw.items()}, str(tmp_path / "vae.safetensors"))
    for name in ("Wan2.1_VAE.pth", "vae.safetensors"):
        vae = AutoencoderKLWan.from_pretrained(str(tmp_path / name), additional_kwargs={"latent_channels": 16})
        got = vae.state_dict()
        assert all(torch.equal(got[k], v) for k, v in sd.items())
        assert vae.latent_channels == 16 and vae.config.temporal_compression_ratio == 4 and vae.spatial_compression_ratio == 8


def _named_params(vae, ea, da):
    out = {}
    for pre, mod in (("encoder_prompt.", ea), ("decoder_prompt.", da), ("vae.", vae)):
        for n, p in mod.named_parameters():
            out[pre + n] = p
    return out


@pytest.mark.parametrize("tag", ["A", "B"])
def test_vae_train_step_host_logic_matches_reference_gradients(monkeypatch, tag):
    """more4d_amd.vae_autograd (per-chunk recompute, truncated-cache gradients, conv data / weight gradients through the padded
    pixel-major panels, attention / norm / up-sampling backward) driven through the torch stand-ins: loss terms, forward values and
    every parameter gradient of a train_vae.py step equal the reference's (tests/golden/vae_train.npz)."""
    import cpu_ops
    from util import grad_sample
    from more4d_amd.models.trajectory_module import VAEDecoderadaptor, VAEEncoderadaptor
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    cpu_ops.install(monkeypatch)
    z = load_npz("vae_train.npz")
    vae = AutoencoderKLWan()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    ea, da = VAEEncoderadaptor(), VAEDecoderadaptor()
    ea.load_state_dict(fill(load_keys("adaptor_enc_keys.json"), 78))
    da.load_state_dict(fill(load_keys("adaptor_dec_keys.json"), 77))
    ea.requires_grad_(True).train()
    da.requires_grad_(True).train()
    vae.model.encoder.requires_grad_(False).eval()
    vae.model.conv1.requires_grad_(False)
    vae.model.decoder.requires_grad_(True).train()
    targets = z["targets"]
    pseudo = ea(targets) * 2 - 1
    assert rel_err(pseudo.detach(), z[f"{tag}/pseudo"]) < 1e-4
    if tag == "A":
        with torch.no_grad():
            posterior = vae.encode_memory_saver(pseudo).latent_dist
    else:
        posterior = vae.encode_memory_saver(pseudo).latent_dist
    assert rel_err(posterior.parameters.detach(), z[f"{tag}/params"]) < 1e-4
    latents = posterior.mean + posterior.std * z[f"{tag}/eps"]
    recon = vae.decode_memory_saver(latents).sample
    assert rel_err(recon.detach(), z[f"{tag}/recon"]) < 1e-4
    rec2 = da(recon)
    assert rel_err(rec2.detach(), z[f"{tag}/reconstructions"]) < 1e-4
    rec_loss = (rec2.float() - targets.float()).abs()
    nll = rec_loss.sum() / rec_loss.shape[0]
    kl = posterior.kl().sum() / posterior.kl().shape[0]
    loss = nll + 1e-6 * kl
    assert abs(float(loss.detach()) - float(z[f"{tag}/loss"])) < 1e-4 * float(z[f"{tag}/loss"])
    loss.backward()
    named = _named_params(vae, ea, da)
    names = [k[len(tag) + 6:] for k in z if k.startswith(f"{tag}/grad/")]
    gmax = max(float(z[f"{tag}/grad/{n}"].abs().max()) for n in names)
    worst = ("", 0.0)
    for n in names:
        if n.startswith("vae.model.conv1"):
            continue                      # frozen here with the encoder (the reference leaves it trainable but never steps it)
        g = named[n].grad
        assert g is not None, n
        ref = z[f"{tag}/grad/{n}"]
        e = float((grad_sample(g).double() - ref.double()).abs().max() / max(float(ref.abs().max()), 1e-3 * gmax))
        if e > worst[1]:
            worst = (n, e)
        assert e < 1e-3, (n, e)
    print("worst VAE-train gradient error", worst)
    if tag == "A":
        assert all(p.grad is None for p in ea.parameters())
