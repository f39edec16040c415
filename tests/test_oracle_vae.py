"""Pin oracle/vae.py (VAE streaming + adaptors) to fixtures produced by the reference itself."""
import pytest
import torch

from util import load_keys, load_npz, rel_err
from weights import fill

from oracle import vae as ov

TOL = 2e-5


def sd_of(z):
    return {k[3:]: v for k, v in z.items() if k.startswith("sd.")}


def test_causal_conv_and_rmsnorm():
    z = load_npz("vae_ops.npz")
    sd = {"c.weight": z["cc_w"], "c.bias": z["cc_b"], "n.gamma": z["rn_g"]}
    assert rel_err(ov.causal_conv3d(sd, "c", z["cc_x"]), z["cc_out_nocache"]) < TOL
    assert rel_err(ov.causal_conv3d(sd, "c", z["cc_x"], z["cc_cache"]), z["cc_out_cache"]) < TOL
    assert rel_err(ov.causal_conv3d(sd, "c", z["cc_x"], z["cc_cache"][:, :, -1:]), z["cc_out_cache1"]) < TOL
    assert rel_err(ov.rms_norm(sd, "n", z["cc_x"]), z["rn_out"]) < TOL


@pytest.mark.parametrize("mode", ["upsample2d", "upsample3d", "downsample2d", "downsample3d"])
def test_resample_streaming(mode):
    z = load_npz(f"vae_resample_{mode}.npz")
    sd = {"r." + k: v for k, v in sd_of(z).items()}
    st = ov.Stream()
    for i in range(3):
        out = ov.resample(sd, "r", mode, z[f"c{i}"], st)
        assert out.shape == z[f"o{i}"].shape
        assert rel_err(out, z[f"o{i}"]) < TOL, (mode, i)


def test_residual_block_streaming():
    z = load_npz("vae_resblock.npz")
    sd = {"b." + k: v for k, v in sd_of(z).items()}
    st = ov.Stream()
    for i in range(3):
        assert rel_err(ov.residual_block(sd, "b", z[f"c{i}"], st), z[f"o{i}"]) < TOL, i


def test_attention_block():
    z = load_npz("vae_attn.npz")
    sd = {"a." + k: v for k, v in sd_of(z).items()}
    assert rel_err(ov.attention_block(sd, "a", z["x"]), z["out"]) < TOL


def test_vae_encode_decode_roundtrip():
    """AutoencoderKLWan encode/decode on [1,3,9,32,32] (3 chunks), full-size network, recipe weights."""
    z = load_npz("vae_roundtrip.npz")
    sd = fill(load_keys("vae_keys.json"), 2024)
    enc = ov.vae_encode(sd, z["x"])
    assert rel_err(enc, z["enc"]) < 1e-4
    dec = ov.vae_decode(sd, z["enc"][:, :16])
    assert rel_err(dec, z["dec"]) < 1e-4


def test_adaptors():
    z = load_npz("adaptor_enc.npz")
    assert rel_err(ov.encoder_adaptor(sd_of(z), z["x"]), z["out"]) < TOL
    z = load_npz("adaptor_dec.npz")
    assert rel_err(ov.decoder_adaptor(sd_of(z), z["x"]), z["out"]) < TOL


@pytest.mark.parametrize("tag", ["A", "B"])
def test_vae_train_step_oracle_matches_reference_gradients(tag):
    """oracle/vae_train.py under torch autograd == the reference's train_vae.py step (tests/golden/vae_train.npz): loss terms,
    forward values and EVERY parameter gradient the reference produced — A: step as written (encode under no_grad: decoder +
    decoder prompt), B: gradient through the frozen encoder (encoder prompt + KL path as well)."""
    from oracle import vae_train as ovt
    from util import grad_sample
    z = load_npz("vae_train.npz")
    sdv = {k: v.clone().requires_grad_(True) for k, v in fill(load_keys("vae_keys.json"), 2024).items()}
    sde = {k: v.clone().requires_grad_(True) for k, v in fill(load_keys("adaptor_enc_keys.json"), 78).items()}
    sdd = {k: v.clone().requires_grad_(True) for k, v in fill(load_keys("adaptor_dec_keys.json"), 77).items()}
    loss, nll, kl, fwd = ovt.train_step_loss(sdv, sde, sdd, z["targets"], z[f"{tag}/eps"], grad_through_encoder=(tag == "B"))
    for k in ("pseudo", "params", "latents", "recon", "reconstructions"):
        assert rel_err(fwd[k].detach(), z[f"{tag}/{k}"]) < 2e-5, k
    assert abs(float(nll) - float(z[f"{tag}/nll"])) < 1e-5 * float(z[f"{tag}/nll"])
    assert abs(float(kl) - float(z[f"{tag}/kl"])) < 1e-5 * float(z[f"{tag}/kl"])
    loss.backward()
    named = {**{"vae." + k: v for k, v in sdv.items()}, **{"encoder_prompt." + k: v for k, v in sde.items()},
             **{"decoder_prompt." + k: v for k, v in sdd.items()}}
    names = [k[len(tag) + 6:] for k in z if k.startswith(f"{tag}/grad/")]
    assert len(names) > 100
    gmax = max(float(z[f"{tag}/grad/{n}"].abs().max()) for n in names)
    for n in names:
        g = named[n].grad
        assert g is not None, n
        ref = z[f"{tag}/grad/{n}"]
        e = float((grad_sample(g).double() - ref.double()).abs().max() / max(float(ref.abs().max()), 1e-3 * gmax))
        assert e < 2e-4, (n, e)
    # nothing else received a gradient (the frozen encoder, and in A the encoder prompt)
    extra = [n for n, v in named.items() if v.grad is not None and n not in names and not n.startswith("vae.model.encoder")
             and not n.startswith("vae.model.conv1")]
    assert not extra, extra
