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


def test_adaptors_host_logic(monkeypatch):
    from more4d_amd.models.trajectory_module import VAEDecoderadaptor, VAEEncoderadaptor
    cpu_ops.install(monkeypatch)
    for cls, name in ((VAEEncoderadaptor, "adaptor_enc.npz"), (VAEDecoderadaptor, "adaptor_dec.npz")):
        z = load_npz(name)
        m = cls().eval()
        m.load_state_dict(sd_of(z), strict=True)
        with torch.no_grad():
            out = m(z["x"])
        assert rel_err(out, z["out"]) < 2e-5, name
