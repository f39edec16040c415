"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE ITSELF (CPU fp32).

Run only in the build container (needs /root/reference):
    python tests/golden/make_golden.py [dit|vae|loop|ops|all]
Outputs are small .npz files: seeded inputs, the reference's randomly initialised weights
(state-dict key names unchanged) and the reference's outputs.  They are data only; no
reference source is copied.  Same container + same torch => reproducible.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def npz_save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path)/1e6:.2f} MB")


def sd_arrays(prefix, module):
    return {prefix + k: v for k, v in module.state_dict().items()}


TINY_DIT = dict(model_type="i2v", in_dim=64, dim=128, ffn_dim=512, num_heads=4, num_layers=2,
                text_dim=64, text_len=32, freq_dim=256, out_dim=16, add_ref_conv=True,
                use_dino_guidance=False, use_omnimae_guidance=False, cross_attn_norm=True)


def load_recipe(module, keys_json, seed, prefix=""):
    """Fill `module` from tests/golden/weights.py's recipe and record {key: shape} as a fixture
    (the state-dict contract our own modules must reproduce)."""
    import json
    from weights import fill
    shapes = {prefix + k: list(t.shape) for k, t in module.state_dict().items()}
    if keys_json is not None:
        with open(os.path.join(HERE, keys_json), "w") as fh:
            json.dump(shapes, fh, indent=0, sort_keys=True)
    sd = fill(shapes, seed)
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    return sd


def randomize(model, gen_seed=4321, std=0.05):
    """The reference zero-inits head.head.weight, biases, and guidance gates (:1378,1390,750-755):
    give every zero-initialised tensor N(0, std) values so the fixture exercises them."""
    g = torch.Generator().manual_seed(gen_seed)
    for name, p in model.named_parameters():
        if float(p.abs().sum()) == 0.0:
            p.copy_(torch.randn(p.shape, generator=g) * std)
        elif name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or \
                name.endswith("norm_k_img.weight") or name.endswith("norm3.weight") or \
                ".proj.0.weight" in name or ".proj.4.weight" in name:
            p.add_(torch.randn(p.shape, generator=g) * 0.1)


def make_dit_tiny(ref):
    """Config 1 of SURVEY 8(d): tiny DiT, latent 1x32x32, with full_ref, seq_len padding."""
    m = ref.dit.WanTransformer4DModel(**TINY_DIT).eval()
    load_recipe(m, "dit_tiny_keys.json", seed=1234)
    g = torch.Generator().manual_seed(7)
    B = 2
    x = torch.randn(B, 16, 2, 16, 16, generator=g)
    y = torch.randn(B, 48, 2, 16, 16, generator=g)
    full_ref = torch.randn(B, 16, 16, 16, generator=g)
    ctx = [torch.randn(9, 64, generator=g), torch.randn(1, 64, generator=g)]
    clip = torch.randn(B, 257, 1280, generator=g)
    t = torch.tensor([500.0, 37.0])
    L = 2 * 8 * 8
    out_ref = m(x=x, t=t, context=ctx, seq_len=L + 5, clip_fea=clip, y=y, full_ref=full_ref)
    out_noref = m(x=x, t=t, context=ctx, seq_len=L, clip_fea=clip, y=y, full_ref=None)
    npz_save("dit_tiny.npz", x=x, y=y, full_ref=full_ref, ctx0=ctx[0], ctx1=ctx[1], clip=clip, t=t,
             seq_len_pad=np.int64(L + 5), seq_len=np.int64(L), out_ref=out_ref, out_noref=out_noref)


def grad_sample(g, n=4096):
    """Deterministic subsample of a gradient tensor (keeps the fixture small): every k-th element."""
    flat = g.detach().reshape(-1)
    k = max(1, flat.numel() // n)
    return flat[::k][:n].clone()


def make_dit_grads(ref):
    """Training-step contract (train_wan.py:1939-1988): reference forward with autograd, thresholded-MSE loss
    (custom_mse_loss :1953-1963, threshold 50, unit weighting), loss.backward(); per-parameter gradient norms and
    subsampled gradient values of every parameter of the tiny DiT, on the dit_tiny.npz inputs (with ref row)."""
    z = dict(np.load(os.path.join(HERE, "dit_tiny.npz")))
    z = {k: torch.from_numpy(v) for k, v in z.items()}
    with torch.enable_grad():
        m = ref.dit.WanTransformer4DModel(**TINY_DIT).train()
        load_recipe(m, None, seed=1234)
        target = torch.randn(z["out_ref"].shape, generator=torch.Generator().manual_seed(21))
        pred = m(x=z["x"], t=z["t"], context=[z["ctx0"], z["ctx1"]], seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"],
                 y=z["y"], full_ref=z["full_ref"])
        diff = pred.float() - target
        loss = (torch.nn.functional.mse_loss(pred.float(), target, reduction="none") * (diff.abs() <= 50).float()).mean()
        loss.backward()
    out = {"target": target, "loss": loss.detach(), "pred": pred.detach()}
    for name, p_ in m.named_parameters():
        if p_.grad is None:
            continue
        out["norm/" + name] = p_.grad.norm()
        out["grad/" + name] = grad_sample(p_.grad)
    npz_save("dit_tiny_grads.npz", **out)


def make_dit_guid_grads(ref, t_override=None, out_name="dit_tiny_guid_grads.npz"):
    """(t_override / out_name: tests/golden/make_golden_r6.py re-runs this with PER-TOKEN timesteps)
    Guided training step (train_wan.sh: --use_omnimae_guidance; train_wan.py:1939-1951 passes first_frame): the reference
    forward + backward with spatial guidance on.  The OmniMAE ViT-B (needs timm/hydra, frozen, train_wan.py:952) and
    torchvision's Normalize are absent here: a stand-in extractor returns the seeded synthetic patch / cls features stored in
    the fixture (the product takes exactly those through first_frame_features), Normalize is restated ((x-mean)/std).
    Everything downstream of the features — feature_adapter convs, bilinear resize, T-repeat, every SpatialGuidanceModule,
    the blocks — is the reference's own code and autograd."""
    import types
    z = dict(np.load(os.path.join(HERE, "dit_tiny.npz")))
    z = {k: torch.from_numpy(v) for k, v in z.items()}
    B = z["x"].shape[0]
    g = torch.Generator().manual_seed(77)
    patch = torch.randn(B, 196, 768, generator=g)
    cls = torch.randn(B, 768, generator=g)

    class _Trunk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.calls = 0

        def forward_patch_features(self, img, mask):
            i = self.calls % B
            self.calls += 1
            return patch[i:i + 1].clone(), cls[i:i + 1].clone()

    class _Extractor(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.trunk = _Trunk()

    om = types.ModuleType("MoRe4D.models.omnimae")
    om.vit_base_mae_pretraining = lambda: _Extractor()
    sys.modules["MoRe4D.models.omnimae"] = om
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    tvt.Normalize = Normalize
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt

    cfg = dict(TINY_DIT, use_omnimae_guidance=True)
    with torch.enable_grad():
        m = ref.dit.WanTransformer4DModel(**cfg).train()
        sd = load_recipe(m, "dit_tiny_guid_keys.json", seed=4321)
        for n_, p_ in m.named_parameters():
            p_.requires_grad_("omnimae_extractor" not in n_)
        target = torch.randn(z["out_ref"].shape, generator=torch.Generator().manual_seed(22))
        first_frame = torch.rand(B, 3, 224, 224, generator=g)
        pred = m(x=z["x"], t=z["t"] if t_override is None else t_override, context=[z["ctx0"], z["ctx1"]], seq_len=int(z["seq_len_pad"]),
                 clip_fea=z["clip"], y=z["y"], full_ref=z["full_ref"], first_frame=first_frame)
        diff = pred.float() - target
        loss = (torch.nn.functional.mse_loss(pred.float(), target, reduction="none") * (diff.abs() <= 50).float()).mean()
        loss.backward()
    assert m.omnimae_extractor.trunk.calls == B
    out = {"target": target, "loss": loss.detach(), "pred": pred.detach(), "patch": patch, "cls": cls}
    for name, p_ in m.named_parameters():
        if p_.grad is None:
            continue
        out["norm/" + name] = p_.grad.norm()
        out["grad/" + name] = grad_sample(p_.grad)
    print("guided params with grad:", sum(1 for k in out if k.startswith("norm/")),
          "gate grad norm", float(out["norm/blocks.0.spatial_guidance_self.gate"]),
          "adapter grad norm", float(out["norm/feature_adapter.0.weight"]))
    npz_save(out_name, **out)


def _load_reference_omnimae():
    """The reference's vendored OmniMAE ViT (MoRe4D/models/omnimae.py + omnivision/models/vision_transformer.py) imports `timm`
    (DropPath, trunc_normal_) and `hydra`, both absent: stand-in NAMES only (DropPath at rate 0 is the identity, trunc_normal_
    is torch's; hydra is not touched on this path)."""
    import types
    hy = types.ModuleType("hydra")
    sys.modules["hydra"] = hy
    timm, tm, tl = types.ModuleType("timm"), types.ModuleType("timm.models"), types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            assert p == 0.0

        def forward(self, x):
            return x

    tl.DropPath, tl.trunc_normal_ = DropPath, torch.nn.init.trunc_normal_
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})
    for name in ("MoRe4D.models.omnivision", "MoRe4D.models.omnivision.models", "MoRe4D.models.omnivision.utils"):
        _ref_import._mod(name)
    _ref_import._load("MoRe4D.models.omnivision.models.vision_transformer",
                      "MoRe4D/models/omnivision/models/vision_transformer.py")
    return _ref_import._load("MoRe4D.models.omnimae", "MoRe4D/models/omnimae.py")


def make_omnimae(ref):
    """OmniMAE ViT-B patch features: the reference's vit_base_mae_pretraining(pretrained=False) trunk with
    the recipe weights (the fixed sinusoidal pos_embed kept as built), on a seeded 2x3x96x160 frame normalised as at
    wan_transformer4d.py:1130-1133 -> forward_patch_features."""
    om = _load_reference_omnimae()
    m = om.vit_base_mae_pretraining(pretrained=False).eval()
    pos = m.trunk.pos_embed.detach().clone()
    load_recipe(m, None, seed=555)
    m.trunk.pos_embed.copy_(pos)
    import json
    enc = {k: list(v.shape) for k, v in m.state_dict().items()
           if k.startswith("trunk.") and not k.startswith("trunk.decoder") and k not in ("trunk.mask_token", "trunk.pos_embed")}
    with open(os.path.join(HERE, "omnimae_keys.json"), "w") as fh:
        json.dump(enc, fh, indent=0, sort_keys=True)
    g = torch.Generator().manual_seed(5)
    frame = torch.rand(2, 3, 96, 160, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    feats, cls = m.trunk.forward_patch_features((frame - mean) / std, None)
    print("omnimae feats", tuple(feats.shape), float(feats.abs().mean()), "cls", tuple(cls.shape))
    npz_save("omnimae.npz", frame=frame, feats=feats, cls=cls, pos_head=pos[0, :4, :8], pos_tail=pos[0, 190:196, 760:])


def make_riflex(ref):
    """RIFLEx temporal RoPE table (get_1d_rotary_pos_embed_riflex, wan_transformer4d.py:264-321; enable_riflex :1011-1025) for the
    tiny config's head dim 32 and the 14B head dim 128: sampled rows of the complex table."""
    out = {}
    for d in (32, 128):
        da = d - 4 * (d // 6)
        f = ref.dit.get_1d_rotary_pos_embed_riflex(1024, da, use_real=False, k=6 if d == 128 else 2, L_test=66, L_test_scale=4.886)
        rows = torch.tensor([0, 1, 65, 66, 1023])
        out[f"re{d}"], out[f"im{d}"] = f[rows].real, f[rows].imag
    npz_save("riflex.npz", **out)


def make_teacache_coeffs(ref):
    """get_teacache_coefficients of the reference (cache_utils.py:4-16) for the model names of the released checkpoints and a few
    near misses: the product's table is pinned to it (tests/test_host_logic.py)."""
    import json
    cu = sys.modules["MoRe4D.models.cache_utils"]
    names = ["Wan2.1-Fun-V1.1-14B-Control", "Wan2.1-Fun-14B-InP", "Wan2.1-T2V-1.3B", "Wan2.1-Fun-V1.1-1.3B-Control", "Wan2.1-T2V-14B",
             "Wan2.1-I2V-14B-480P", "Wan2.1-I2V-14B-720P", "Wan2.2-Fun-A14B-Control", "Wan2.2-TI2V-5B", "models/Wan2.1-Fun-V1.1-14B-Control/",
             "SomethingElse-7B"]
    out = {n: cu.get_teacache_coefficients(n) for n in names}
    with open(os.path.join(HERE, "teacache_coeffs.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("wrote teacache_coeffs.json", {k: (v[0] if v else None) for k, v in out.items()})


def make_dit_ops(ref):
    """Per-op vectors: sinusoid, rope (incl. padded tail), rmsnorm, LN-modulate, SDPA,
    self-attn, cross-attn, block (with and without spatial guidance), head."""
    d = ref.dit
    g = torch.Generator().manual_seed(11)
    out = {}
    pos = torch.tensor([0.0, 1.0, 500.0, 999.0])
    out["sin_pos"] = pos
    out["sin_out"] = d.sinusoidal_embedding_1d(256, pos)
    # rope: grid (2,3,4) = 24 tokens, 3 padded rows
    hd = 128
    freqs = torch.cat([d.rope_params(1024, hd - 4 * (hd // 6)), d.rope_params(1024, 2 * (hd // 6)),
                       d.rope_params(1024, 2 * (hd // 6))], dim=1)
    xq = torch.randn(1, 27, 2, hd, generator=g)
    out["rope_x"] = xq
    out["rope_out"] = d.rope_apply(xq, torch.tensor([[2, 3, 4]]), freqs)
    # rmsnorm
    rn = d.WanRMSNorm(256, eps=1e-6)
    rn.weight.copy_(1 + 0.1 * torch.randn(256, generator=g))
    xr = torch.randn(2, 5, 256, generator=g) * 3
    out["rms_x"], out["rms_w"], out["rms_out"] = xr, rn.weight, rn(xr)
    # sdpa
    q = torch.randn(1, 64, 4, 128, generator=g)
    k = torch.randn(1, 80, 4, 128, generator=g)
    v = torch.randn(1, 80, 4, 128, generator=g)
    out["att_q"], out["att_k"], out["att_v"] = q, k, v
    out["att_out"] = d.attention(q, k, v)
    npz_save("dit_ops.npz", **out)

    # one block dim 256 / 2 heads (head_dim 128), with and without guidance
    torch.manual_seed(99)
    for guid in (False, True):
        blk = d.WanAttentionBlock("i2v_cross_attn", 256, 1024, 2, (-1, -1), True, True, 1e-6,
                                  use_spatial_guidance=guid).eval()
        load_recipe(blk, "dit_block_guid_keys.json" if guid else "dit_block_keys.json", seed=99,
                    prefix="blocks.0.")
        grid = (2, 4, 5)
        L = 40 + 3
        x = torch.randn(2, L, 256, generator=g)
        e0 = torch.randn(2, 6, 256, generator=g) * 0.3
        ctx = torch.randn(2, 257 + 16, 256, generator=g)
        freqs = torch.cat([d.rope_params(1024, 128 - 4 * (128 // 6)), d.rope_params(1024, 2 * (128 // 6)),
                           d.rope_params(1024, 2 * (128 // 6))], dim=1)
        feats = cls = None
        if guid:
            feats = torch.randn(2, 40, 768, generator=g)
            cls = torch.randn(2, 1, 768, generator=g)
        y = blk(x, e0, torch.tensor([L, L]), torch.tensor([list(grid)] * 2), freqs, ctx, None,
                dtype=torch.float32, t=0, dino_features=(feats, cls), use_cls_token=False)
        extra = dict(feats=feats, cls=cls) if guid else {}
        npz_save("dit_block_guid.npz" if guid else "dit_block.npz", x=x, e0=e0, ctx=ctx,
                 grid=np.array(grid), out=y, **extra)


def make_block_14b_width(ref):
    """One 14B-width block (dim 5120 / ffn 13824 / 40 heads) at small L=260, grid (5,4,13).
    Weights are NOT stored (1.6 GB): they are regenerated from the seed by a recipe both this
    script and the tests implement (tests/golden/weights.py); only input/output are stored."""
    from weights import block_weights_14b
    d = ref.dit
    blk = d.WanAttentionBlock("i2v_cross_attn", 5120, 13824, 40, (-1, -1), True, True, 1e-6,
                              use_spatial_guidance=False).eval()
    sd = block_weights_14b(seed=0)
    blk.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items()})
    from weights import randn_named
    grid = (5, 4, 13)
    L = 260
    x = randn_named("in.x", (1, L, 5120), 5)
    e0 = randn_named("in.e0", (1, 6, 5120), 5, 0.2)
    ctx = randn_named("in.ctx", (1, 257 + 512, 5120), 5)
    freqs = torch.cat([d.rope_params(1024, 128 - 4 * (128 // 6)), d.rope_params(1024, 2 * (128 // 6)),
                       d.rope_params(1024, 2 * (128 // 6))], dim=1)
    y = blk(x, e0, torch.tensor([L]), torch.tensor([list(grid)]), freqs, ctx, None,
            dtype=torch.float32, t=0, dino_features=None)
    npz_save("dit_block_14b.npz", grid=np.array(grid), out=y)  # inputs: randn_named(..., seed 5)


def make_loop(ref):
    """50-step CFG Euler loop on the tiny DiT: hand-restated loop
    around the imported reference DiT + the in-tree order-1 FlowDPMSolverMultistepScheduler."""
    m = ref.dit.WanTransformer4DModel(**TINY_DIT).eval()
    load_recipe(m, None, seed=1234)
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(1, 16, 1, 32, 32, generator=g)
    y = torch.randn(1, 48, 1, 32, 32, generator=g)
    full_ref = torch.randn(1, 16, 32, 32, generator=g)
    ctx_c = torch.randn(9, 64, generator=g)
    ctx_u = torch.randn(1, 64, generator=g)
    clip = torch.randn(1, 257, 1280, generator=g)
    steps, shift, gs = 50, 5.0, 6.0
    sch = ref.fm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    sig = ref.fm.get_sampling_sigmas(steps, shift)
    # the pipeline passes pre-shifted sigmas through retrieve_timesteps -> set_timesteps(sigmas=...)
    # which applies `shift` (config.shift=1.0 => identity) (fm_solvers.py:226-289)
    sch.set_timesteps(sigmas=sig)
    timesteps = sch.timesteps.clone()
    sigmas = sch.sigmas.clone()
    x = lat.clone()
    seq_len = 16 * 16
    for i, t in enumerate(timesteps):
        inp = torch.cat([x, x])
        tt = t.expand(2)
        v = m(x=inp, t=tt, context=[ctx_u, ctx_c], seq_len=seq_len, clip_fea=torch.cat([clip, clip]),
              y=torch.cat([y, y]), full_ref=torch.cat([full_ref, full_ref]))
        vu, vc = v.chunk(2)
        v = vu + gs * (vc - vu)
        x = sch.step(v, t, x, return_dict=False)[0]
    print("loop final abs-mean", float(x.abs().mean()), "sum", float(x.sum()))
    npz_save("loop_tiny.npz", lat=lat, y=y, full_ref=full_ref, ctx_c=ctx_c, ctx_u=ctx_u, clip=clip,
             timesteps=timesteps, sigmas=sigmas, final=x, steps=np.int64(steps), shift=np.float64(shift),
             guidance=np.float64(gs))


def make_vae(ref):
    """AutoencoderKLWan encode/decode on [1,3,9,32,32] (3 chunks), per-op vectors for
    CausalConv3d / RMS_norm / Resample (first vs later chunk) / ResidualBlock / AttentionBlock,
    and the two trajectory adaptors."""
    import json
    from weights import fill
    v = ref.vae
    vae = v.AutoencoderKLWan().eval()
    shapes = {k: list(t.shape) for k, t in vae.state_dict().items()}
    with open(os.path.join(HERE, "vae_keys.json"), "w") as fh:
        json.dump(shapes, fh, indent=0, sort_keys=True)
    vae.load_state_dict(fill(shapes, seed=2024))  # 127 M params: regenerated by recipe, not stored
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 3, 9, 32, 32, generator=g) * 2 - 1
    enc = vae._encode(x)  # [1,32,3,4,4] (mu normalised | logvar)
    z = enc[:, :16]
    dec = vae._decode(z).sample
    npz_save("vae_roundtrip.npz", x=x, enc=enc, dec=dec)

    # per-op fixtures with small channel counts (weights stored)
    out = {}
    conv = v.CausalConv3d(8, 12, 3, padding=1)
    xx = torch.randn(1, 8, 3, 6, 7, generator=g)
    cache = torch.randn(1, 8, 2, 6, 7, generator=g)
    out.update(cc_w=conv.weight, cc_b=conv.bias, cc_x=xx, cc_cache=cache,
               cc_out_nocache=conv(xx), cc_out_cache=conv(xx, cache),
               cc_out_cache1=conv(xx, cache[:, :, -1:]))
    rn = v.RMS_norm(8, images=False)
    rn.gamma.add_(0.2 * torch.randn(rn.gamma.shape, generator=g))
    out.update(rn_g=rn.gamma, rn_out=rn(xx))
    npz_save("vae_ops.npz", **out)

    # streaming sub-modules: run 3 chunks through each Resample mode / ResidualBlock with cache
    for mode in ("upsample2d", "upsample3d", "downsample2d", "downsample3d"):
        torch.manual_seed(5)
        rs = v.Resample(8, mode).eval()
        chunks = [torch.randn(1, 8, 1, 6, 8, generator=g), torch.randn(1, 8, 2, 6, 8, generator=g),
                  torch.randn(1, 8, 2, 6, 8, generator=g)]
        cache = [None]
        outs = []
        for c in chunks:
            idx = [0]
            outs.append(rs(c, cache, idx))
        arrs = sd_arrays("sd.", rs)
        npz_save(f"vae_resample_{mode}.npz", c0=chunks[0], c1=chunks[1], c2=chunks[2],
                 o0=outs[0], o1=outs[1], o2=outs[2], **arrs)
    torch.manual_seed(6)
    rb = v.ResidualBlock(8, 12).eval()
    chunks = [torch.randn(1, 8, 1, 6, 8, generator=g), torch.randn(1, 8, 4, 6, 8, generator=g),
              torch.randn(1, 8, 1, 6, 8, generator=g)]
    cache = [None, None]
    outs = []
    for c in chunks:
        idx = [0]
        outs.append(rb(c, cache, idx))
    npz_save("vae_resblock.npz", c0=chunks[0], c1=chunks[1], c2=chunks[2], o0=outs[0], o1=outs[1],
             o2=outs[2], **sd_arrays("sd.", rb))
    torch.manual_seed(7)
    ab = v.AttentionBlock(16).eval()
    torch.nn.init.normal_(ab.proj.weight, std=0.1)
    xa = torch.randn(1, 16, 2, 5, 6, generator=g)
    npz_save("vae_attn.npz", x=xa, out=ab(xa), **sd_arrays("sd.", ab))

    # adaptors
    t = ref.traj
    torch.manual_seed(8)
    ea = t.VAEEncoderadaptor().eval()
    da = t.VAEDecoderadaptor().eval()
    for p in ea.conv_out.parameters():
        p.copy_(0.05 * torch.randn(p.shape, generator=g))
    xt = torch.randn(1, 3, 2, 16, 24, generator=g) * 0.3
    npz_save("adaptor_enc.npz", x=xt, out=ea(xt), **sd_arrays("sd.", ea))
    npz_save("adaptor_dec.npz", x=xt, out=da(xt), **sd_arrays("sd.", da))


def _probe_samples(t, step):
    """every `step`-th element (flat) + per-(channel, frame) L2 norms of a [1, C, T, H, W] tensor"""
    return t.reshape(-1)[::step].clone(), t[0].flatten(2).norm(dim=-1)


def make_vae_probe(ref):
    """The whole Motion-Sensitive VAE chain of train_vae.py:434-453 (encoder adaptor -> x*2-1 -> encode -> mode -> decode -> decoder
    adaptor) at map sizes where the PRODUCTION bf16 tile paths of our conv kernels are selected: 9 x 120 x 208 (BASELINE.md section 3's
    probe: 24 x 16 three-pixel-tile patches at the 96-channel level) and 5 x 96 x 128 (12 x 32 patches).  Weights by recipe
    (weights.py); the hand-offs between stages are rounded to fp16 BEFORE the next reference stage consumes them, so every stored
    stage output is the reference's exact output for the stored (fp16) stage input and the stages can be tested in isolation."""
    import json
    from weights import fill
    v, t = ref.vae, ref.traj
    vae = v.AutoencoderKLWan().eval()
    vae.load_state_dict(fill(json.load(open(os.path.join(HERE, "vae_keys.json"))), seed=2024))
    ea, da = t.VAEEncoderadaptor().eval(), t.VAEDecoderadaptor().eval()
    ea.load_state_dict(fill(json.load(open(os.path.join(HERE, "adaptor_enc_keys.json"))), seed=31))
    da.load_state_dict(fill(json.load(open(os.path.join(HERE, "adaptor_dec_keys.json"))), seed=32))
    for name, (T, H, W), seed in (("vae_probe_120x208.npz", (9, 120, 208), 120208), ("vae_probe_96x128.npz", (5, 96, 128), 96128)):
        g = torch.Generator().manual_seed(seed)
        traj = torch.rand(1, 3, T, H, W, generator=g)                  # normalised coordinates (regenerated from the seed by the test)
        pseudo = ea(traj) * 2 - 1
        pv16 = pseudo.half()
        enc = vae._encode(pv16.float())                                # [1, 32, (T-1)/4+1, H/8, W/8]: mu (normalised) | logvar
        z16 = enc[:, :16].half()
        dec = vae._decode(z16.float()).sample
        dec16 = dec.half()
        rec = da(dec16.float())
        ps, pn = _probe_samples(pseudo, 11)
        ds, dn = _probe_samples(dec, 11)
        rs, rn = _probe_samples(rec, 11)
        print(name, "abs-mean pseudo / enc / dec / rec:", float(pseudo.abs().mean()), float(enc.abs().mean()),
              float(dec.abs().mean()), float(rec.abs().mean()))
        npz_save(name, shape=np.array([T, H, W]), seed=np.array(seed), pv16=pv16, enc=enc, dec16=dec16,
                 pseudo_s=ps, pseudo_n=pn, dec_s=ds, dec_n=dn, rec_s=rs, rec_n=rn)


def make_block_14b_long(ref):
    """One 14B-width block at L = 2080 (grid (4, 20, 26)): long enough that the PRODUCTION bf16 kernels are on the path
    (gemm_bt256p_kernel needs M, N >= 512; attn128p_kernel needs Lq > 1024 and >= 2048 keys).  Weights and inputs are
    regenerated from the seeded recipe; stored: 65 sampled output rows and every row's L2 norm (the full output is 42 MB)."""
    from weights import block_weights_14b, randn_named
    d = ref.dit
    blk = d.WanAttentionBlock("i2v_cross_attn", 5120, 13824, 40, (-1, -1), True, True, 1e-6,
                              use_spatial_guidance=False).eval()
    sd = block_weights_14b(seed=0)
    blk.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items()})
    grid = (4, 20, 26)
    L = 2080
    x = randn_named("in.x", (1, L, 5120), 6)
    e0 = randn_named("in.e0", (1, 6, 5120), 6, 0.2)
    ctx = randn_named("in.ctx", (1, 257 + 512, 5120), 6)
    freqs = torch.cat([d.rope_params(1024, 128 - 4 * (128 // 6)), d.rope_params(1024, 2 * (128 // 6)),
                       d.rope_params(1024, 2 * (128 // 6))], dim=1)
    y = blk(x, e0, torch.tensor([L]), torch.tensor([list(grid)]), freqs, ctx, None,
            dtype=torch.float32, t=0, dino_features=None)
    rows = torch.arange(0, L, 32)
    rows = torch.cat([rows, torch.tensor([L - 1])])
    npz_save("dit_block_14b_long.npz", grid=np.array(grid), rows=rows, out_rows=y[0, rows], row_norm=y[0].norm(dim=-1),
             delta_rows=(y - x)[0, rows], delta_norm=(y - x)[0].norm(dim=-1))


def make_teacache_loop(ref):
    """TeaCache pinned to the reference (cache_utils.py:19-74, hooks wan_transformer4d.py:1201-1270, 1336-1339): the tiny DiT
    in a 10-step CFG Euler loop with enable_teacache(coefficients("Wan2.1-Fun-14B-Control"), 10, thresh, num_skip_start_steps=1,
    offload=False): per-step compute/skip decisions, the accumulated distance, every step's latent and the final latent."""
    from MoRe4D.models.cache_utils import get_teacache_coefficients
    m = ref.dit.WanTransformer4DModel(**TINY_DIT).eval()
    load_recipe(m, None, seed=1234)
    g = torch.Generator().manual_seed(17)
    lat = torch.randn(1, 16, 1, 32, 32, generator=g)
    y = torch.randn(1, 48, 1, 32, 32, generator=g)
    full_ref = torch.randn(1, 16, 32, 32, generator=g)
    ctx_c = torch.randn(9, 64, generator=g)
    ctx_u = torch.randn(1, 64, generator=g)
    clip = torch.randn(1, 257, 1280, generator=g)
    steps, shift, gs = 10, 5.0, 6.0
    coeff = get_teacache_coefficients("Wan2.1-Fun-14B-Control")
    out = {}
    for tag, thresh in (("a", 15000.0), ("b", 30000.0)):      # the tiny random model moves e0 by ~100 % per step: rescaled distances are 5e3-2e4
        m.enable_teacache(coeff, steps, thresh, num_skip_start_steps=1, offload=False)
        sch = ref.fm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
        sch.set_timesteps(sigmas=ref.fm.get_sampling_sigmas(steps, shift))
        x = lat.clone()
        calc, acc, traj = [], [], []
        for i, t in enumerate(sch.timesteps):
            v = m(x=torch.cat([x, x]), t=t.expand(2), context=[ctx_u, ctx_c], seq_len=256, clip_fea=torch.cat([clip, clip]),
                  y=torch.cat([y, y]), full_ref=torch.cat([full_ref, full_ref]))
            calc.append(bool(m.should_calc))
            acc.append(float(m.teacache.accumulated_rel_l1_distance) if m.teacache.cnt != 0 else 0.0)
            vu, vc = v.chunk(2)
            x = sch.step(vu + gs * (vc - vu), t, x, return_dict=False)[0]
            traj.append(x.clone())
        print("teacache", tag, thresh, "calc:", calc)
        out.update({f"{tag}_thresh": np.float64(thresh), f"{tag}_calc": np.array(calc), f"{tag}_acc": np.array(acc),
                    f"{tag}_traj": torch.stack(traj), f"{tag}_final": x})
        m.disable_teacache()
    npz_save("teacache_loop.npz", lat=lat, y=y, full_ref=full_ref, ctx_c=ctx_c, ctx_u=ctx_u, clip=clip, steps=np.int64(steps),
             shift=np.float64(shift), guidance=np.float64(gs), coeff=np.array(coeff), **out)


def _reference_functions(path, names, extra=None):
    """Execute only the named top-level functions of a reference script that cannot be imported as a whole (scripts/inference/
    infer.py pulls in UniDepth, gsplat, decord ...): the function definitions are compiled FROM THE REFERENCE FILE at fixture
    generation time — nothing is copied into this repository."""
    import ast
    import typing
    import torch.nn.functional as F_
    src = open(path).read()
    tree = ast.parse(src)
    def wanted(n):
        if isinstance(n, ast.FunctionDef):
            return n.name in names
        if isinstance(n, ast.Assign):      # module-level constants the functions read (e.g. DEFAULT_H_ORI, DEFAULT_W_ORI = 540, 960)
            return any(isinstance(t, ast.Name) and t.id in names for tg in n.targets for t in ast.walk(tg))
        return False
    keep = [n for n in tree.body if wanted(n)]
    assert len(keep) == len(names) - sum(1 for x in names if x.isupper()) + 1 or len(keep) >= 1, [getattr(n, "name", "assign") for n in keep]
    ns = {"torch": torch, "F": F_, "np": np, "Tuple": typing.Tuple, "List": typing.List, "Dict": typing.Dict,
          "Optional": typing.Optional}
    ns.update(extra or {})
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def make_pipeline_chain(ref):
    """The whole stage-1 chain around the denoise loop, hand-restated from pipeline_wan_fun_control.py:626-723 (conditioning),
    :741-840 (loop), :382-386 (decode_latents_no_normalize) and scripts/inference/infer.py:820-871 (depth prologue, decoder
    prompt, coordinate recovery) — every ARITHMETIC step calls the imported reference: AutoencoderKLWan, WanTransformer4DModel,
    VAEDecoderadaptor, FlowDPMSolverMultistepScheduler and infer.py's back_project_coords / inverse_flow_norm_transform_no_diff.
    (diffusers' VaeImageProcessor.preprocess is absent: its published behaviour for same-size tensors in [0, 1] is x*2-1.)"""
    from weights import fill
    inf = _reference_functions(os.path.join(_ref_import.REF_ROOT, "scripts/inference/infer.py"),
                               ["DEFAULT_H_ORI", "get_intrinsic_matrix", "back_project_coords", "inverse_flow_norm_transform_no_diff"])
    H = W = 32
    NF = 5
    vae = ref.vae.AutoencoderKLWan().eval()
    shapes = {k: list(t.shape) for k, t in vae.state_dict().items()}
    vae.load_state_dict(fill(shapes, seed=2024))
    dec_prompt = ref.traj.VAEDecoderadaptor().eval()
    dp_shapes = {k: list(t.shape) for k, t in dec_prompt.state_dict().items()}
    dec_prompt.load_state_dict(fill(dp_shapes, seed=77))
    m = ref.dit.WanTransformer4DModel(**TINY_DIT).eval()
    load_recipe(m, None, seed=1234)
    g = torch.Generator().manual_seed(23)
    image01 = torch.rand(1, 3, 1, H, W, generator=g)                      # get_image_latent: [1,3,1,H,W] in [0,1]
    control_video = image01.repeat(1, 1, NF, 1, 1)                        # create_control_video_from_image(...).repeat (:816-817)
    ref_image = image01.clone()
    depth_pred = torch.rand(24, 24, generator=g) * 3 + 0.5                # depth model output at its own resolution
    depth_pred[5, 6] = 0.0                                                # hits the "< 1e-5 -> 1" branch of the depth prologue
    depth_bad = depth_pred.clone()                                        # NaN / inf depth: pins the prologue's clean-up only (the
    depth_bad[3, 4] = float("nan")                                        # coordinate recovery of the reference turns all-NaN on it)
    depth_bad[7, 9] = float("inf")
    ctx_c = torch.randn(9, 64, generator=g)
    ctx_u = torch.randn(1, 64, generator=g)
    clip = torch.randn(1, 257, 1280, generator=g)
    lat0 = torch.randn(1, 16, 2, H // 8, W // 8, generator=g)
    steps, shift, gs = 4, 3.0, 6.0
    # ---- depth prologue (infer.py:820-828)
    def depth_prologue(dp):
        ffc = inf["back_project_coords"](dp, H, W, torch.device("cpu"))
        ffc = ffc.permute(2, 0, 1).unsqueeze(0).unsqueeze(2)
        dpv = ffc[:, 2, :, :].unsqueeze(1).repeat(1, 3, 1, 1, 1)
        dpv = torch.clamp(dpv, min=0.0, max=10000.0)
        dpv[torch.isinf(dpv) | torch.isnan(dpv) | (dpv < 1e-5)] = 1
        dmin, dmax = dpv.min(), dpv.max()
        return ffc, 2 * (dpv - dmin) / (dmax - dmin + 1e-8) - 1
    first_frame_coords, dpv = depth_prologue(depth_pred)
    _, dpv_bad = depth_prologue(depth_bad)
    # ---- conditioning (pipeline :626-723): preprocess = x*2-1 for [0,1] tensors, encode(...).mode()
    enc = lambda v: vae.encode(v.float())[0].mode()
    control_latents = enc(control_video * 2 - 1)
    depth_latents = enc(dpv.repeat(1, 1, NF, 1, 1).float())
    ref_latents = enc(ref_image * 2 - 1)[:, :, 0]
    start = torch.zeros_like(lat0)
    yv = torch.cat([control_latents, start, depth_latents], dim=1)
    # ---- loop (:741-840)
    sch = ref.fm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    sch.set_timesteps(sigmas=ref.fm.get_sampling_sigmas(steps, shift))
    x = lat0.clone()
    seq_len = (H // 16) * (W // 16) * lat0.shape[2]
    for t in sch.timesteps:
        v = m(x=torch.cat([x, x]), t=t.expand(2), context=[ctx_u, ctx_c], seq_len=seq_len, clip_fea=torch.cat([clip, clip]),
              y=torch.cat([yv, yv]), full_ref=torch.cat([ref_latents, ref_latents]))
        vu, vc = v.chunk(2)
        x = sch.step(vu + gs * (vc - vu), t, x, return_dict=False)[0]
    video = vae.decode(x).sample                                          # decode_latents_no_normalize (clamped to [-1,1], :825-832)
    recon = dec_prompt(video).float()                                     # infer.py:848-849
    flow_rel, diff = inf["inverse_flow_norm_transform_no_diff"](recon, first_frame_coords)          # :862
    # (the --normalize_track_z branch :857-860 adds a [1,3,H,W] tensor to [1,3,F,H,W]: it raises for F != 3 in the reference
    #  itself, so there is nothing to pin; more4d_amd.utils.io.recover_coords implements the evident intent, unpinned)
    coords_rel = torch.cat([first_frame_coords, flow_rel[:, :, 1:]], dim=2)                          # :870
    import json
    with open(os.path.join(HERE, "adaptor_dec_keys.json"), "w") as fh:
        json.dump(dp_shapes, fh, indent=0, sort_keys=True)
    npz_save("pipeline_chain.npz", image01=image01, depth_pred=depth_pred, ctx_c=ctx_c, ctx_u=ctx_u, clip=clip, lat0=lat0,
             steps=np.int64(steps), shift=np.float64(shift), guidance=np.float64(gs), num_frames=np.int64(NF),
             first_frame_coords=first_frame_coords, depth_pixel_values=dpv, depth_bad=depth_bad,
             depth_pixel_values_bad=dpv_bad, control_latents=control_latents,
             depth_latents=depth_latents, ref_latents=ref_latents, y=yv, final_latents=x, video=video, recon=recon,
             flow_rel=flow_rel, diff=diff, coords_rel=coords_rel)


def make_vae_train(ref):
    """One train_vae.py step (scripts/4D_STraG_training/train_vae.py:434-495, loss :173-187: L1 summed over the sample + 1e-6 KL)
    on [1,3,5,32,32] targets with the reference's modules, `--finetune_vae_decoder`:
      A ("as written"): the encode runs under torch.no_grad() (:444-448), so only decoder_prompt and vae.model.decoder (+ conv2)
        receive gradients — through decode_memory_saver = decode_full (wan_vae.py:633-676: per-latent-frame checkpoint, the
        streaming cache DETACHED between slices, then clamp_(-1, 1));
      B ("gradient through the frozen encoder"): the same step with the no_grad removed — what encode_memory_saver = encode_full
        (wan_vae.py:549-613) exists for: gradients reach encoder_prompt through the frozen encoder, and the KL term has a
        gradient path.
    Stored: inputs, eps of posterior.sample(), forward values, loss terms, and for every parameter with a gradient its norm and a
    4096-value subsample (weights come from the seeded recipe)."""
    import json
    from weights import fill
    tv = _reference_functions(os.path.join(_ref_import.REF_ROOT, "scripts/4D_STraG_training/train_vae.py"), ["compute_loss"])
    import types
    args = types.SimpleNamespace(rec_loss="l1", kl_scale=1e-6)
    torch.set_grad_enabled(True)
    try:
        vae = ref.vae.AutoencoderKLWan()
        shapes = {k: list(t.shape) for k, t in vae.state_dict().items()}
        vae.load_state_dict(fill(shapes, seed=2024))
        ea, da = ref.traj.VAEEncoderadaptor(), ref.traj.VAEDecoderadaptor()
        ea_shapes = {k: list(t.shape) for k, t in ea.state_dict().items()}
        da_shapes = {k: list(t.shape) for k, t in da.state_dict().items()}
        with open(os.path.join(HERE, "adaptor_enc_keys.json"), "w") as fh:
            json.dump(ea_shapes, fh, indent=0, sort_keys=True)
        ea.load_state_dict(fill(ea_shapes, seed=78))
        da.load_state_dict(fill(da_shapes, seed=77))
        ea.requires_grad_(True).train()
        da.requires_grad_(True).train()
        vae.model.encoder.requires_grad_(False).eval()          # train_vae.py:355
        vae.model.decoder.requires_grad_(True).train()          # :357-358
        g = torch.Generator().manual_seed(31)
        coords = torch.randn(1, 3, 5, 32, 32, generator=g) * 0.3
        targets = coords - coords[:, :, 0:1]                    # normalize_coordinates default branch (:168-170)
        out = {"targets": targets}

        def named(prefix, mod):
            return {prefix + n: p for n, p in mod.named_parameters()}
        allp = {**named("encoder_prompt.", ea), **named("decoder_prompt.", da), **named("vae.", vae)}

        def record(tag):
            n = 0
            for name, p in allp.items():
                if p.grad is not None:
                    out[f"{tag}/grad/{name}"] = grad_sample(p.grad)
                    out[f"{tag}/norm/{name}"] = p.grad.norm()
                    n += 1
                    p.grad = None
            print(tag, "parameters with gradients:", n)

        for tag in ("A", "B"):
            pseudo = ea(targets) * 2 - 1
            eg = torch.Generator().manual_seed(5)
            if tag == "A":
                with torch.no_grad():
                    posterior = vae.encode_memory_saver(pseudo).latent_dist
                    latents = posterior.sample(generator=eg)
            else:
                posterior = vae.encode_memory_saver(pseudo).latent_dist
                latents = posterior.sample(generator=eg)
            recon = vae.decode_memory_saver(latents).sample
            rec2 = da(recon)
            loss, nll, kl = tv["compute_loss"](rec2, targets, posterior, args)
            loss.backward()
            eps = torch.randn(posterior.mean.shape, generator=torch.Generator().manual_seed(5))
            assert torch.allclose(latents.detach(), posterior.mean.detach() + posterior.std.detach() * eps)
            out.update({f"{tag}/pseudo": pseudo.detach(), f"{tag}/params": posterior.parameters.detach(), f"{tag}/eps": eps,
                        f"{tag}/latents": latents.detach(), f"{tag}/recon": recon.detach(), f"{tag}/reconstructions": rec2.detach(),
                        f"{tag}/loss": loss.detach(), f"{tag}/nll": nll.detach(), f"{tag}/kl": kl.detach()})
            print(tag, "loss", float(loss), "nll", float(nll), "kl", float(kl), "clamped", float((recon.abs() >= 1).float().mean()))
            record(tag)
        npz_save("vae_train.npz", **out)
    finally:
        torch.set_grad_enabled(False)


def make_sched(ref):
    """sigma/timestep tables of the in-tree order-1 solver and one step, 50 steps shift 5."""
    sig = ref.fm.get_sampling_sigmas(50, 5.0)
    sch = ref.fm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    sch.set_timesteps(sigmas=sig)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 16, 2, 4, 4, generator=g)
    v = torch.randn(1, 16, 2, 4, 4, generator=g)
    x1 = sch.step(v, sch.timesteps[0], x, return_dict=False)[0]
    x2 = sch.step(v, sch.timesteps[1], x1, return_dict=False)[0]
    npz_save("sched.npz", sigmas=sch.sigmas, timesteps=sch.timesteps, x=x, v=v, x1=x1, x2=x2,
             sampling_sigmas=sig)


def make_sched_api(ref):
    """The scheduler's img2img / mid-schedule entry points (fm_solvers.py:216-224 set_begin_index, :679-704 index_for_timestep /
    _init_step_index, :815-854 add_noise, :856 __len__) on the 50-step shift-5 schedule."""
    sch = ref.fm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    sch.set_timesteps(sigmas=ref.fm.get_sampling_sigmas(50, 5.0))
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 16, 2, 4, 4, generator=g)
    n = torch.randn(2, 16, 2, 4, 4, generator=g)
    ts = sch.timesteps[[3, 17]]
    out = dict(x=x, n=n, ts=ts, noisy_lookup=sch.add_noise(x, n, ts), length=np.int64(len(sch)),
               idx=np.array([sch.index_for_timestep(t) for t in ts]))
    sch.set_begin_index(10)
    out["noisy_begin"] = sch.add_noise(x, n, ts)                       # before the first step: sigma of begin_index for every sample
    v = torch.randn(2, 16, 2, 4, 4, generator=g)
    out["v"] = v
    out["x_step"] = sch.step(v, sch.timesteps[10], x, return_dict=False)[0]   # step counter starts at begin_index
    out["step_index_after"] = np.int64(sch.step_index)
    out["noisy_mid"] = sch.add_noise(x, n, ts)                         # after a step: sigma of the current step index
    npz_save("sched_api.npz", **out)


def toy_velocity(x, t):
    """Deterministic stand-in for the DiT inside solver fixtures (smooth in x and t)."""
    return 0.3 * x + 0.1 * torch.sin(3.0 * x) + (float(t) / 1000.0 - 0.5)


def make_sched_multistep(ref):
    """DPM-Solver++ multistep orders 2 and 3 of the in-tree flow solver (fm_solvers.py:486-677, step :706-797):
    the reference scheduler driven by a toy velocity for 8 steps (< 15: the lower-order tail rules apply) and 20 steps."""
    out = {}
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(1, 16, 2, 4, 4, generator=g)
    out["x0"] = x0
    for order in (2, 3):
        for steps in (8, 20):
            sch = ref.fm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=order, shift=1.0)
            sch.set_timesteps(sigmas=ref.fm.get_sampling_sigmas(steps, 5.0))
            x = x0.clone()
            traj = []
            for t in sch.timesteps:
                x = sch.step(toy_velocity(x, t), t, x, return_dict=False)[0]
                traj.append(x.clone())
            out[f"o{order}_s{steps}"] = torch.stack(traj)
    npz_save("sched_multistep.npz", **out)


def make_sched_unipc(ref):
    """FlowUniPCMultistepScheduler (fm_solvers_unipc.py: predictor :350-484, corrector :486-626, step :655-739), bh2,
    predict_x0, orders 2 and 3, driven by the toy velocity; sigma tables both from get_sampling_sigmas (first sigma
    exactly 1) and from the scheduler's own linspace (set_timesteps(num_inference_steps))."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_unipc", "/root/reference/MoRe4D/utils/fm_solvers_unipc.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(1, 16, 2, 4, 4, generator=g)
    out["x0"] = x0
    for order in (2, 3):
        # (with get_sampling_sigmas the first sigma is exactly 1 => lambda = -inf and the reference itself returns NaN from
        #  its order >= 2 corrector; UniPC is therefore pinned on the scheduler's own linspace tables)
        for steps, mode in ((12, "lin"), (30, "lin")):
            sch = mod.FlowUniPCMultistepScheduler(num_train_timesteps=1000, solver_order=order, shift=1.0)
            if mode == "sig":
                sch.set_timesteps(sigmas=ref.fm.get_sampling_sigmas(steps, 5.0))
            else:
                sch.set_timesteps(steps, shift=5.0)
            x = x0.clone()
            traj = []
            for t in sch.timesteps:
                x = sch.step(toy_velocity(x, t), t, x, return_dict=False)[0]
                traj.append(x.clone())
            out[f"o{order}_{mode}{steps}"] = torch.stack(traj)
            out[f"o{order}_{mode}{steps}_sigmas"] = sch.sigmas
            out[f"o{order}_{mode}{steps}_timesteps"] = sch.timesteps
    npz_save("sched_unipc.npz", **out)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ref = _ref_import.load_reference()
    if what in ("dit", "all"):
        make_dit_tiny(ref)
        make_dit_ops(ref)
    if what in ("grads", "all"):
        make_dit_grads(ref)
    if what in ("riflex", "all"):
        make_riflex(ref)
    if what in ("teacache", "all"):
        make_teacache_coeffs(ref)
    if what in ("omnimae", "all"):
        make_omnimae(ref)
    if what in ("guidgrads", "all"):
        make_dit_guid_grads(ref)
    if what in ("dit14b", "all"):
        make_block_14b_width(ref)
    if what in ("dit14blong", "all"):
        make_block_14b_long(ref)
    if what in ("teacache_loop", "all"):
        make_teacache_loop(ref)
    if what in ("chain", "all"):
        make_pipeline_chain(ref)
    if what in ("vaetrain", "all"):
        make_vae_train(ref)
    if what in ("loop", "all"):
        make_loop(ref)
    if what in ("vae", "all"):
        make_vae(ref)
    if what in ("vaeprobe", "all"):
        make_vae_probe(ref)
    if what in ("sched", "all"):
        make_sched(ref)
        make_sched_api(ref)
        make_sched_multistep(ref)
        make_sched_unipc(ref)
