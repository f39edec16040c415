"""Pin the oracle (oracle/*.py) to fixtures produced by running the reference itself
(tests/golden/make_golden.py).  CPU only; runs in seconds.  Tolerance: the oracle is the same
fp32 algorithm as the reference, only re-associated, so 2e-5 relative is ample."""
import numpy as np
import pytest
import torch

from util import load_keys, load_npz, rel_err
from weights import fill

from oracle import dit as odit

TOL = 2e-5

TINY = odit.DiTConfig(model_type="i2v", in_dim=64, dim=128, ffn_dim=512, num_heads=4, num_layers=2,
                      text_dim=64, text_len=32, freq_dim=256, out_dim=16, add_ref_conv=True,
                      cross_attn_norm=True)


def test_sinusoid():
    z = load_npz("dit_ops.npz")
    out = odit.sinusoidal_embedding_1d(256, z["sin_pos"])
    assert rel_err(out, z["sin_out"]) < 1e-12


def test_rope_with_padded_tail():
    z = load_npz("dit_ops.npz")
    out = odit.rope_apply(z["rope_x"], (2, 3, 4))
    assert out.dtype == torch.float32
    assert rel_err(out, z["rope_out"]) < 1e-6
    assert torch.equal(out[:, 24:], z["rope_x"][:, 24:])


def test_rmsnorm():
    z = load_npz("dit_ops.npz")
    assert rel_err(odit.rms_norm(z["rms_x"], z["rms_w"], 1e-6), z["rms_out"]) < 1e-6


def test_sdpa():
    z = load_npz("dit_ops.npz")
    assert rel_err(odit.sdpa(z["att_q"], z["att_k"], z["att_v"]), z["att_out"]) < 1e-5


@pytest.mark.parametrize("guid", [False, True])
def test_block(guid):
    z = load_npz("dit_block_guid.npz" if guid else "dit_block.npz")
    sd = fill(load_keys("dit_block_guid_keys.json" if guid else "dit_block_keys.json"), 99)
    cfg = odit.DiTConfig(dim=256, ffn_dim=1024, num_heads=2, use_spatial_guidance=guid)
    g = (z["feats"], z["cls"]) if guid else None
    out = odit.block_forward(sd, 0, cfg, z["x"], z["e0"], tuple(z["grid"].tolist()), z["ctx"], g)
    assert rel_err(out, z["out"]) < TOL


def test_tiny_dit_forward():
    z = load_npz("dit_tiny.npz")
    sd = fill(load_keys("dit_tiny_keys.json"), 1234)
    ctx = [z["ctx0"], z["ctx1"]]
    out = odit.dit_forward(sd, TINY, z["x"], z["t"], ctx, int(z["seq_len_pad"]), z["clip"], z["y"],
                           z["full_ref"])
    assert rel_err(out, z["out_ref"]) < TOL
    out = odit.dit_forward(sd, TINY, z["x"], z["t"], ctx, int(z["seq_len"]), z["clip"], z["y"], None)
    assert rel_err(out, z["out_noref"]) < TOL


def test_sched_tables_and_steps():
    from oracle import sched
    z = load_npz("sched.npz")
    sig = sched.sampling_sigmas(50, 5.0)
    assert np.allclose(sig, z["sampling_sigmas"].numpy(), rtol=0, atol=0)
    ts, sigmas = sched.set_timesteps(sig)
    assert torch.equal(ts, z["timesteps"])
    assert torch.equal(sigmas, z["sigmas"])
    x1 = sched.euler_step(z["x"], z["v"], sigmas[0], sigmas[1])
    assert rel_err(x1, z["x1"]) < 5e-6
    x2 = sched.euler_step(x1, z["v"], sigmas[1], sigmas[2])
    assert rel_err(x2, z["x2"]) < 5e-6


def test_loop_50_steps():
    """Config 1 (BASELINE.json configs[0]): 50-step CFG Euler loop on the tiny DiT."""
    from oracle import sched
    z = load_npz("loop_tiny.npz")
    sd = fill(load_keys("dit_tiny_keys.json"), 1234)
    ts, sigmas = sched.set_timesteps(sched.sampling_sigmas(int(z["steps"]), float(z["shift"])))
    assert torch.equal(ts, z["timesteps"])

    def model_fn(x2, t2):
        return odit.dit_forward(sd, TINY, x2, t2, [z["ctx_u"], z["ctx_c"]], 16 * 16,
                                torch.cat([z["clip"]] * 2), torch.cat([z["y"]] * 2),
                                torch.cat([z["full_ref"]] * 2))

    out = sched.denoise_loop(model_fn, z["lat"], ts, sigmas, float(z["guidance"]))
    assert rel_err(out, z["final"]) < 2e-4  # 50 chained fp32 forwards


def test_omnimae_vit_patch_features():
    """oracle/omnimae.py == the reference's own VisionTransformer.forward_patch_features (tests/golden/omnimae.npz)."""
    from oracle import omnimae as oom
    z = load_npz("omnimae.npz")
    sd = fill(load_keys("omnimae_keys.json"), 555)
    sd["trunk.pos_embed"] = oom.sinusoid_table(8 * 14 * 14, 768)
    assert torch.equal(sd["trunk.pos_embed"][0, :4, :8], z["pos_head"])
    assert rel_err(sd["trunk.pos_embed"][0, 190:196, 760:], z["pos_tail"]) < 1e-6
    feats, cls = oom.forward_patch_features(sd, oom.normalize(z["frame"]))
    assert rel_err(feats, z["feats"]) < 2e-5 and rel_err(cls, z["cls"]) < 2e-5


def test_geometry_oracle_matches_reference_chain():
    """oracle/geometry.py against the reference's own back_project_coords / depth prologue / inverse_flow_norm_transform_no_diff
    (tests/golden/pipeline_chain.npz, made by make_golden.py:make_pipeline_chain)."""
    from oracle import geometry as og
    z = load_npz("pipeline_chain.npz")
    H = W = 32
    ffc = og.back_project_coords(z["depth_pred"], H, W).permute(2, 0, 1)[None, :, None]
    assert rel_err(ffc, z["first_frame_coords"]) < 1e-6
    assert rel_err(og.depth_control_image(ffc), z["depth_pixel_values"]) < 1e-6
    bad = og.back_project_coords(z["depth_bad"], H, W).permute(2, 0, 1)[None, :, None]
    assert rel_err(og.depth_control_image(bad), z["depth_pixel_values_bad"]) < 1e-6
    flow, diff = og.recover_flow(z["recon"], z["first_frame_coords"])
    assert rel_err(flow, z["flow_rel"]) < 1e-6 and rel_err(diff, z["diff"]) < 1e-7
    assert rel_err(og.stage1_coords(z["recon"], z["first_frame_coords"]), z["coords_rel"]) < 1e-6


def test_chain_oracle_matches_reference():
    """The whole stage-1 chain (conditioning encodes -> CFG loop -> decode -> decoder prompt -> coordinates) composed from the
    oracle pieces equals the chain composed from the reference's modules."""
    from oracle import dit as odit, geometry as og, sched as osched, vae as ovae
    from weights import fill
    from util import load_keys
    z = load_npz("pipeline_chain.npz")
    NF = int(z["num_frames"])
    vsd = fill(load_keys("vae_keys.json"), 2024)
    enc = lambda v: ovae.vae_encode(vsd, v.float())[:, :16]
    ctrl = enc(og.preprocess_image(z["image01"].repeat(1, 1, NF, 1, 1)))
    assert rel_err(ctrl, z["control_latents"]) < 2e-5
    depth = enc(z["depth_pixel_values"].repeat(1, 1, NF, 1, 1))
    assert rel_err(depth, z["depth_latents"]) < 2e-5
    refl = enc(og.preprocess_image(z["image01"]))[:, :, 0]
    assert rel_err(refl, z["ref_latents"]) < 2e-5
    y = torch.cat([ctrl, torch.zeros_like(ctrl), depth], dim=1)
    sd = fill(load_keys("dit_tiny_keys.json"), 1234)
    cfg = TINY
    ts, sig = osched.set_timesteps(osched.sampling_sigmas(int(z["steps"]), float(z["shift"])))

    def fn(x2, t2):
        return odit.dit_forward(sd, cfg, x2, t2, [z["ctx_u"], z["ctx_c"]], 2 * 2 * 2, torch.cat([z["clip"]] * 2), torch.cat([y] * 2),
                                torch.cat([refl] * 2))
    final = osched.denoise_loop(fn, z["lat0"], ts, sig, float(z["guidance"]))
    assert rel_err(final, z["final_latents"]) < 1e-4
    video = ovae.vae_decode(vsd, final).clamp(-1, 1)
    assert rel_err(video, z["video"]) < 1e-4
    recon = ovae.decoder_adaptor(fill(load_keys("adaptor_dec_keys.json"), 77), video)
    assert rel_err(recon, z["recon"]) < 1e-4
