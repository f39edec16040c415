"""Round-3 GPU parity: the Motion-Sensitive VAE chain pinned to the REFERENCE at map sizes where the production bf16 tile paths are
selected, BASELINE configs[2] at its full 49 frames (weak #2), and the launch-class diagnostics that prove
which kernels a case ran."""
import pytest
import torch

from util import bf16_budget, load_keys, load_npz, rel_err, rms_rel_err
from weights import fill

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _chain_modules(dtype):
    from more4d_amd.models.trajectory_module import VAEDecoderadaptor, VAEEncoderadaptor
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    vae = AutoencoderKLWan().eval()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    ea, da = VAEEncoderadaptor().eval(), VAEDecoderadaptor().eval()
    ea.load_state_dict(fill(load_keys("adaptor_enc_keys.json"), 31))
    da.load_state_dict(fill(load_keys("adaptor_dec_keys.json"), 32))
    return vae.to(DEV, dtype), ea.to(DEV, dtype), da.to(DEV, dtype)


def _check(name, got, samples, norms, tol, report):
    """got [1, C, T, H, W] against the reference's every-11th element and per-(channel, frame) L2 norms; tol = (max, rms, norm)"""
    g = got.float().cpu()
    e_max = rel_err(g.reshape(-1)[::11], samples)
    e_rms = rms_rel_err(g.reshape(-1)[::11], samples)
    e_nrm = rel_err(g[0].flatten(2).norm(dim=-1), norms)
    report[name] = (e_max, e_rms, e_nrm)
    assert e_max <= tol[0] and e_rms <= tol[1] and e_nrm <= tol[2], (name, (e_max, e_rms, e_nrm), tol)


@pytest.mark.parametrize("size", ["120x208", "96x128"])
@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["fp32", "bf16"])
def test_vae_chain_vs_reference_production_tiles(size, dtype):
    """train_vae.py:434-453's chain (encoder adaptor -> x*2-1 -> encode -> decode -> decoder adaptor; wan_vae.py:190-224, 520-547,
    678-703; trajectory_module.py:125-279) against the reference's own outputs at 9 x 120 x 208 (BASELINE.md section 3's probe size)
    and 5 x 96 x 128.  Every stage is fed the reference's (fp16-rounded, stored) input of that stage, so the stages are judged in
    isolation; the last check runs the chain end to end on our own hand-offs.
    fp32: north_star's 1e-3.  bf16: the production kernels — asserted through the launch-class counters: the three-pixel-tile
    LDS-halo kernels (24 x 16 patches at 120 x 208, 12 x 32 at 96 x 128), the conv epilogues that write the next layer's RMS-norm
    (inside a residual block and across blocks) and the GroupNorm statistics of the adaptors."""
    from more4d_amd import ops
    z = load_npz(f"vae_probe_{size}.npz")
    T, H, W = (int(v) for v in z["shape"])
    traj = torch.rand(1, 3, T, H, W, generator=torch.Generator().manual_seed(int(z["seed"])))
    vae, ea, da = _chain_modules(dtype)
    fp32 = dtype == torch.float32

    def tol(stage):
        """fp32: north_star's 1e-3.  bf16: 1.5 x what the REFERENCE'S OWN bf16-autocast run of this stage loses against its fp32 run
        on the same stored inputs (bf16_calibration.json; 2 x for the max-type metrics, which are extreme values of ~1e6 samples
        and move by tens of percent between two equally good bf16 implementations)."""
        if fp32:
            return (1e-3, 1e-3, 1e-3)
        key = f"vae_probe_{size}"
        nrm = bf16_budget(key, stage, "nrm", factor=2.0) if stage != "encode" else 0.0
        return (bf16_budget(key, stage, "max", factor=2.0), bf16_budget(key, stage, "rms"), max(nrm, 2e-3))
    rep = {}
    ops.launch_counts(reset=True)
    with torch.no_grad():
        pseudo = ea(traj.to(DEV)) * 2 - 1      # fp32 coordinates in (train_vae.py:438): the skip / sigmoid of the adaptor stay fp32 like the reference's autocast run
        _check("enc-adaptor", pseudo, z["pseudo_s"], z["pseudo_n"], tol("enc-adaptor"), rep)
        enc = vae._encode(z["pv16"].float().to(DEV, dtype))
        e_enc = (rel_err(enc.float().cpu(), z["enc"]), rms_rel_err(enc.float().cpu(), z["enc"]))
        rep["encode"] = e_enc
        assert e_enc[0] <= tol("encode")[0] and e_enc[1] <= tol("encode")[1], (e_enc, tol("encode"))
        dec = vae.decode(z["enc"][:, :16].half().float().to(DEV, dtype)).sample
        _check("decode", dec, z["dec_s"], z["dec_n"], tol("decode"), rep)
        rec = da(z["dec16"].float().to(DEV, dtype))
        _check("dec-adaptor", rec, z["rec_s"], z["rec_n"], tol("dec-adaptor"), rep)
        counts = ops.launch_counts()
        # end to end on our own hand-offs (the reference's hand-offs were rounded to fp16: noise of 5e-4 per stage on its side)
        chain = da(vae.decode(vae._encode(pseudo)[:, :16].contiguous()).sample)
        _check("chain", chain, z["rec_s"], z["rec_n"], (5e-3, 3e-3, 3e-3) if fp32 else tol("chain"), rep)
    print(size, dtype, {k: tuple(f"{x:.2e}" for x in v) for k, v in rep.items()}, {k: v for k, v in counts.items() if v})
    if fp32:
        assert counts["conv_generic"] > 0 and counts["conv_halo"] + counts["conv_halo_mt3_12x32"] + counts["conv_halo_mt3_24x16"] + counts["conv_halo64"] == 0
    else:
        mt3 = "conv_halo_mt3_24x16" if size == "120x208" else "conv_halo_mt3_12x32"
        assert counts[mt3] + counts["conv_halo64"] > 0, counts      # three pixel tiles per wave / (round 6) five per wave, one wave per SIMD
        assert counts["conv_halo"] > 0, counts             # the 8 x 32 / 16 x 16 patch kernels of the deeper levels
        assert counts["conv_fused_norm"] > 0 and counts["conv_fused_norm_resid"] > 0, counts
        assert counts["conv_gnstats"] > 0, counts


def test_launch_counters_follow_the_dispatch():
    """m4d_launch_count: a big bf16 GEMM lands in one of the two production structures, a ragged-K one in the generic kernel."""
    from more4d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(0)
    a = torch.randn(1024, 512, device=DEV, dtype=BF, generator=g)
    w = torch.randn(768, 512, device=DEV, dtype=BF, generator=g)
    ops.launch_counts(reset=True)
    ops.gemm_bt(a, w)
    c = ops.launch_counts()
    assert c["gemm_phased"] + c["gemm_wide"] == 1 and c["gemm_generic"] == 0, c
    ops.gemm_bt(a[:, :72].contiguous(), w[:, :72].contiguous())
    c = ops.launch_counts(reset=True)
    assert c["gemm_generic"] == 1, c
    assert sum(ops.launch_counts().values()) == 0


def test_vae_configs2_full_49_frames():
    """BASELINE configs[2] at its FULL length: 49 x 480 x 832 through encode and decode (13 latent frames; the first frame alone,
    then twelve 4-frame encoder chunks / latent-frame decoder steps — the staging ring wraps around several times).  Properties the
    chunked causal network must have (wan_vae.py:520-547, 678-703): the 17-frame prefix of the input encodes to exactly the first
    5 latent frames of the 49-frame encode, 5 latent frames decode to exactly the first 17 frames of the full decode, a second run
    is bit-identical (no state leaks through clear_cache), outputs finite and clamped to [-1, 1]."""
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    vae = AutoencoderKLWan().eval()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    vae = vae.to(DEV, BF)
    g = torch.Generator(device=DEV).manual_seed(49)
    x = (torch.rand(1, 3, 49, 480, 832, device=DEV, generator=g) * 2 - 1).to(BF)
    with torch.no_grad():
        full = vae.encode(x)[0].mode()
        assert full.shape == (1, 16, 13, 60, 104) and bool(torch.isfinite(full.float()).all())
        part = vae.encode(x[:, :, :17].contiguous())[0].mode()
        assert torch.equal(part, full[:, :, :5])
        dec = vae.decode(full).sample
        assert dec.shape == (1, 3, 49, 480, 832)
        assert bool(torch.isfinite(dec.float()).all()) and float(dec.float().abs().max()) <= 1.0
        dec_part = vae.decode(full[:, :, :5].contiguous()).sample
        assert torch.equal(dec_part, dec[:, :, :17])
        del dec_part
        assert torch.equal(vae.encode(x)[0].mode(), full)
        assert torch.equal(vae.decode(full).sample, dec)


def test_inner_model_entry_points_on_device():
    """vae.model.encode / encode_full / decode / decode_full / clear_cache under the reference's names (wan_vae.py:520, 549, 633, 678,
    717), called the way the reference's wrapper calls them (:770-776, :825-832), against the reference's own outputs (fp32, 1e-3):
    decode is NOT clamped there; under autograd the *_full twins give the same values as the plain forward."""
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    z = load_npz("vae_roundtrip.npz")
    vae = AutoencoderKLWan().eval()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    vae = vae.to(DEV)
    m = vae.model
    x, lat = z["x"].to(DEV), z["enc"][:, :16].to(DEV)
    with torch.no_grad():
        enc = m.encode(x, vae.scale)
        assert rel_err(enc.cpu(), z["enc"]) < 1e-3 and torch.equal(m.encode_full(x, vae.scale), enc)
        raw = m.decode(lat, vae.scale)
        assert float(raw.abs().max()) > 1.0 and rel_err(raw.clamp(-1, 1).cpu(), z["dec"]) < 1e-3
        assert torch.equal(m.decode_full(lat, vae.scale), raw)
    m.clear_cache()
    assert m._conv_idx == [0] and len(m._feat_map) == m._conv_num > 0
    # with gradients: same forward values, a gradient for the decoder parameters and for the latent
    lat_g = lat.clone().requires_grad_(True)
    out = m.decode_full(lat_g, vae.scale)
    assert rel_err(out.detach().cpu(), raw.cpu()) < 1e-5
    out.square().mean().backward()
    assert lat_g.grad is not None and bool(torch.isfinite(lat_g.grad).all()) and float(lat_g.grad.abs().max()) > 0
    assert m.decoder.head[2].weight.grad is not None


@pytest.mark.parametrize("M,N,K,what", [
    (70000, 512, 256, "548 tiles of 4 K-tiles: the shortest K loop the persistent form takes, ragged last tile row"),
    (4100, 4104, 512, "17 x 17 tiles: both edges shifted inwards, the last round has 33 of 256 workgroups"),
    (33000, 776, 1088, "K/64 odd: stays on the one-tile form"),
    (8200, 2056, 768, "33 x 9 tiles, 12 K-tiles"),
])
def test_gemm_persistent_form_shapes(M, N, K, what):
    """The persistent wide GEMM (default for bf16 stores when there are more tiles than CUs) against fp32 torch on sampled rows, for
    plain / no-bias / tanh-GELU stores; repeated launches must agree bit for bit (a hand-over race between tiles would not).  The
    bit-equality with the one-tile and phased forms is test_round2_gpu.py::test_gemm_switch_paths_in_a_subprocess."""
    from more4d_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().to(DEV)
    b = torch.randn(N, generator=g).bfloat16().to(DEV)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 300), torch.arange(M - 300, M)]).to(DEV)
    ref = a[rows].float() @ w.float().t()
    ops.launch_counts(reset=True)
    for bias, epi, f in ((b, ops.EPI_STORE, lambda r: r + b.float()), (None, ops.EPI_STORE, lambda r: r),
                         (b, ops.EPI_GELU_TANH, lambda r: torch.nn.functional.gelu(r + b.float(), approximate="tanh"))):
        out = ops.gemm_bt(a, w, bias, epilogue=epi)
        want = f(ref)
        assert rel_err(out[rows].float().cpu(), want.cpu()) < 8e-3, what
        for _ in range(2):
            assert torch.equal(ops.gemm_bt(a, w, bias, epilogue=epi), out), what
    assert ops.launch_counts()["gemm_wide"] == 9


@pytest.mark.parametrize("C", [5120, 3072])
@pytest.mark.parametrize("affine", [False, True], ids=["modulate", "affine"])
def test_ln_modulate_rows_form_is_the_one_shot_kernel_bit_for_bit(C, affine):
    """ln_modulate takes its LDS-staged persistent form from 4096 rows on (the DiT's shapes); the same rows fed in chunks of < 4096 go
    through the one-shot kernel.  Same arithmetic in the same order: the outputs must agree bit for bit, and both match fp32 torch."""
    from more4d_amd import ops
    B, L = 2, 4100
    g = torch.Generator().manual_seed(C + affine)
    x = (torch.randn(B * L, C, generator=g) * 2 + 0.3).to(DEV)
    sc = (torch.randn(B, 2, C, generator=g) * 0.3).to(DEV)
    w, b = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    if affine:
        full = ops.ln_modulate(x, BF, ln_w=w, ln_b=b)
    else:
        full = ops.ln_modulate(x, BF, shift=sc[:, 0], scale=sc[:, 1], mod_stride=2 * C, rows_per_sample=L)
    parts = []
    for s_ in range(B):
        for lo, hi in ((0, 2048), (2048, L)):
            xs = x[s_ * L + lo:s_ * L + hi].contiguous()
            if affine:
                parts.append(ops.ln_modulate(xs, BF, ln_w=w, ln_b=b))
            else:
                parts.append(ops.ln_modulate(xs, BF, shift=sc[s_:s_ + 1, 0], scale=sc[s_:s_ + 1, 1], mod_stride=2 * C, rows_per_sample=hi - lo))
    assert torch.equal(full, torch.cat(parts))
    ref = torch.nn.functional.layer_norm(x.view(B, L, C), (C,), eps=1e-6)
    ref = ref * w + b if affine else ref * (1 + sc[:, 1].reshape(B, 1, C)) + sc[:, 0].reshape(B, 1, C)
    assert rel_err(full.float().cpu().view(B, L, C), ref.cpu()) < 8e-3
