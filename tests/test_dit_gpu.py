"""Module-level parity of the HIP DiT path against the golden fixtures produced by the reference itself:
single block (with / without spatial guidance), 14B-width block, tiny DiT forward (with ref row + seq_len
padding, and without), the 50-step CFG/Euler loop (BASELINE.json configs[0]), bf16 production mode."""
import pytest
import torch

from util import load_keys, load_npz, rel_err, rms_rel_err
from weights import block_shapes, fill, randn_named

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3   # north_star: 1e-3 relative fp32 on identical latents/noise

TINY = dict(model_type="i2v", in_dim=64, dim=128, ffn_dim=512, num_heads=4, num_layers=2, text_dim=64, text_len=32,
            freq_dim=256, out_dim=16, add_ref_conv=True, use_dino_guidance=False, cross_attn_norm=True)


def tiny_model(dtype=torch.float32):
    from more4d_amd.models import WanTransformer4DModel
    m = WanTransformer4DModel(**TINY)
    sd = fill(load_keys("dit_tiny_keys.json"), 1234)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV, dtype).eval()


def make_block(dim, ffn, heads, guid, seed, keys_json=None, dtype=torch.float32):
    from more4d_amd.models import WanAttentionBlock
    blk = WanAttentionBlock("i2v_cross_attn", dim, ffn, heads, (-1, -1), True, True, 1e-6, use_spatial_guidance=guid)
    shapes = load_keys(keys_json) if keys_json else block_shapes(dim, ffn, guid)
    sd = {k[len("blocks.0."):]: v for k, v in fill(shapes, seed).items()}
    blk.load_state_dict(sd, strict=True)
    return blk.to(DEV, dtype).eval()


@pytest.mark.parametrize("guid", [False, True])
def test_block_fp32(guid):
    from more4d_amd.models.wan_transformer4d import rope_params
    z = load_npz("dit_block_guid.npz" if guid else "dit_block.npz")
    blk = make_block(256, 1024, 2, guid, 99, "dit_block_guid_keys.json" if guid else "dit_block_keys.json")
    freqs = torch.cat([rope_params(1024, 128 - 4 * (128 // 6)), rope_params(1024, 2 * (128 // 6)),
                       rope_params(1024, 2 * (128 // 6))], dim=1)
    L = z["x"].shape[1]
    feats = (z["feats"], z["cls"]) if guid else None
    with torch.no_grad():
        out = blk(z["x"], z["e0"], torch.tensor([L, L]), z["grid"].view(1, 3).repeat(2, 1), freqs, z["ctx"], None,
                  dtype=torch.float32, t=0, dino_features=feats, use_cls_token=False)
    assert rel_err(out.cpu(), z["out"]) < TOL


def test_block_14b_width_fp32():
    """dim 5120 / ffn 13824 / 40 heads at L=260 against the reference's own output."""
    from more4d_amd.models.wan_transformer4d import rope_params
    z = load_npz("dit_block_14b.npz")
    blk = make_block(5120, 13824, 40, False, 0)
    freqs = torch.cat([rope_params(1024, 128 - 4 * (128 // 6)), rope_params(1024, 2 * (128 // 6)),
                       rope_params(1024, 2 * (128 // 6))], dim=1)
    L = 260
    x = randn_named("in.x", (1, L, 5120), 5)
    e0 = randn_named("in.e0", (1, 6, 5120), 5, 0.2)
    ctx = randn_named("in.ctx", (1, 257 + 512, 5120), 5)
    with torch.no_grad():
        out = blk(x, e0, torch.tensor([L]), z["grid"].view(1, 3), freqs, ctx, None, dtype=torch.float32, t=0)
    assert rel_err(out.cpu(), z["out"]) < TOL


def test_tiny_dit_fp32():
    z = load_npz("dit_tiny.npz")
    m = tiny_model()
    ctx = [z["ctx0"].to(DEV), z["ctx1"].to(DEV)]
    with torch.no_grad():
        out = m(x=z["x"].to(DEV), t=z["t"].to(DEV), context=ctx, seq_len=int(z["seq_len_pad"]),
                clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV), full_ref=z["full_ref"].to(DEV))
        assert rel_err(out.cpu(), z["out_ref"]) < TOL
        out = m(x=z["x"].to(DEV), t=z["t"].to(DEV), context=ctx, seq_len=int(z["seq_len"]),
                clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV), full_ref=None)
        assert rel_err(out.cpu(), z["out_noref"]) < TOL


def test_tiny_dit_bf16_budget():
    """Production dtype: bf16 operands / fp32 accumulate vs the fp32 reference output."""
    z = load_npz("dit_tiny.npz")
    m = tiny_model(torch.bfloat16)
    ctx = [z["ctx0"].to(DEV), z["ctx1"].to(DEV)]
    with torch.no_grad():
        out = m(x=z["x"].to(DEV, torch.bfloat16), t=z["t"].to(DEV), context=ctx, seq_len=int(z["seq_len_pad"]),
                clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV, torch.bfloat16), full_ref=z["full_ref"].to(DEV, torch.bfloat16))
    assert out.dtype == torch.bfloat16
    # budget: the reference's own bf16-autocast run of this call sits 4.3e-3 rms / 4.2e-3 max from its fp32 run
    # (bf16_calibration.json); ours additionally rounds the OUTPUT to bf16 (2^-9 relative per element)
    from util import bf16_budget
    e_rms, e_max = rms_rel_err(out.float().cpu(), z["out_ref"]), rel_err(out.float().cpu(), z["out_ref"])
    print("tiny dit bf16", e_rms, e_max)
    assert e_rms <= bf16_budget("dit_tiny", "rms") + 2e-3
    assert e_max <= bf16_budget("dit_tiny", "max", factor=2.0) + 4e-3


def test_loop_50_steps_fp32():
    """BASELINE.json configs[0]: 50-step CFG Euler loop, tiny DiT, vs the reference's final latent."""
    from more4d_amd.pipeline import denoise_latents
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
    z = load_npz("loop_tiny.npz")
    m = tiny_model()
    sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    ts, _ = retrieve_timesteps(sch, device=DEV, sigmas=get_sampling_sigmas(int(z["steps"]), float(z["shift"])))
    assert torch.equal(ts.cpu(), z["timesteps"])
    assert torch.equal(sch.sigmas, z["sigmas"])
    with torch.no_grad():
        out = denoise_latents(m, sch, z["lat"], ts, float(z["guidance"]), [z["ctx_u"].to(DEV), z["ctx_c"].to(DEV)],
                              clip_fea=z["clip"], y=z["y"], full_ref=z["full_ref"], seq_len=16 * 16)
    assert rel_err(out.cpu(), z["final"]) < TOL


def test_context_cache_and_cfg_skip():
    """prepare_context once == re-embedding every call; cfg_skip duplicates the conditional half."""
    z = load_npz("dit_tiny.npz")
    m = tiny_model()
    ctx = [z["ctx0"].to(DEV), z["ctx1"].to(DEV)]
    args = dict(t=z["t"].to(DEV), seq_len=int(z["seq_len"]), y=z["y"].to(DEV))
    with torch.no_grad():
        a = m(x=z["x"].to(DEV), context=ctx, clip_fea=z["clip"].to(DEV), **args)
        cc = m.prepare_context(ctx, z["clip"].to(DEV))
        b = m(x=z["x"].to(DEV), context=cc, **args)
        b2 = m(x=z["x"].to(DEV), context=cc, **args)   # second use hits the per-layer K/V cache
        assert torch.equal(a, b) and torch.equal(b, b2)
        m.enable_cfg_skip(0.5, 10)
        m.current_steps = 9
        c = m(x=z["x"].to(DEV), context=cc, **args)
        m.disable_cfg_skip()
    assert torch.equal(c[0], c[1])
    assert rel_err(c[1].cpu(), a[1].cpu()) < 1e-5


def test_scheduler_step_matches_golden():
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas
    z = load_npz("sched.npz")
    sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    sch.set_timesteps(sigmas=get_sampling_sigmas(50, 5.0), device=DEV)
    x1 = sch.step(z["v"].to(DEV), sch.timesteps[0], z["x"].to(DEV), return_dict=False)[0]
    x2 = sch.step(z["v"].to(DEV), sch.timesteps[1], x1, return_dict=False)[0]
    assert rel_err(x1.cpu(), z["x1"]) < 1e-5 and rel_err(x2.cpu(), z["x2"]) < 1e-5


def test_teacache_and_small_kernels():
    """axpby / rel_l1 / bilinear kernels vs torch, and TeaCache control flow on the device."""
    import torch.nn.functional as F
    from more4d_amd import ops
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(1000, 33, generator=g), torch.randn(1000, 33, generator=g)
    assert rel_err(ops.axpby(x.to(DEV), y.to(DEV), 1.0, -1.0).cpu(), x - y) < 1e-6
    assert abs(ops.rel_l1(x.to(DEV), y.to(DEV)) - float((y - x).abs().mean() / x.abs().mean())) < 1e-4
    f = torch.randn(2, 14, 14, 96, generator=g)
    ref = F.interpolate(f.permute(0, 3, 1, 2), size=(30, 52), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    assert rel_err(ops.bilinear_cl(f.to(DEV), (30, 52)).cpu(), ref) < 1e-5
    z = load_npz("dit_tiny.npz")
    m = tiny_model()
    args = dict(x=z["x"].to(DEV), context=[z["ctx0"].to(DEV), z["ctx1"].to(DEV)], seq_len=int(z["seq_len"]),
                clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV))
    with torch.no_grad():
        base2 = m(t=z["t"].to(DEV) - 30, **args)
        m.enable_teacache([1.0, 0.0], num_steps=4, rel_l1_thresh=0.0, num_skip_start_steps=1)
        m(t=z["t"].to(DEV), **args)
        b = m(t=z["t"].to(DEV) - 30, **args)
        assert rel_err(b.cpu(), base2.cpu()) < 1e-5
        m.enable_teacache([1.0, 0.0], num_steps=4, rel_l1_thresh=1e9, num_skip_start_steps=1)
        a = m(t=z["t"].to(DEV), **args)
        b = m(t=z["t"].to(DEV) - 30, **args)
        assert torch.isfinite(b).all() and rel_err(b.cpu(), a.cpu()) < 0.5


def test_rccl_gather_plumbing_world1():
    """The RCCL side of the T-sharded path on the real device (world 1 is all a 1-GPU box can offer): async
    all_gather_into_tensor of K and V^T in bf16, stream-ordered wait, segments consumed by the attention kernel,
    head all-gather.  The multi-rank arithmetic is covered by the gloo tests."""
    import os
    import torch.distributed as dist
    from more4d_amd import ops
    from more4d_amd.dist import SequenceParallelGroup
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29611")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sp = SequenceParallelGroup()
        B, Ls, heads, hd = 2, 264, 2, 128
        C = heads * hd
        g = torch.Generator().manual_seed(0)
        q = torch.randn(B * Ls, C, generator=g).to(DEV, torch.bfloat16)
        k = torch.randn(B * Ls, C, generator=g).to(DEV, torch.bfloat16)
        vt = torch.randn(C, B * Ls, generator=g).to(DEV, torch.bfloat16)
        hk = sp.gather_start(k)
        hv = sp.gather_start(vt)
        segs = sp.gather_finish(hk, hv, B, Ls, C, Ls - 3)
        o = ops.attention(q, segs, B=B, Lq=Ls, heads=heads, head_dim=hd, q_bs=Ls * C, q_ls=C)
        ref = ops.attention(q, [ops.KV(k, vt, Ls * C, C, Ls, B * Ls, Ls - 3)], B=B, Lq=Ls, heads=heads, head_dim=hd,
                            q_bs=Ls * C, q_ls=C)
        assert torch.equal(o, ref)
        x = torch.randn(B, Ls, 64, generator=g).to(DEV)
        assert torch.equal(sp.all_gather(x, dim=1), x)
        segs2 = sp.gather_kv(k, vt, B, Ls, C, Ls)
        assert segs2[0].len == Ls and segs2[0].k.data_ptr() != k.data_ptr()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("order,steps", [(2, 8), (3, 20)])
def test_multistep_solver_gpu(order, steps):
    """m4d_lincomb + the order 2 / 3 DPM-Solver++ updates on the device against the reference scheduler's trajectory."""
    from more4d_amd import ops
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas
    z = load_npz("sched_multistep.npz")
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(1000, generator=g) for _ in range(4)]
    got = ops.lincomb([(0.5, xs[0].to(DEV)), (-2.0, xs[1].to(DEV)), (3.0, xs[2].to(DEV)), (0.25, xs[3].to(DEV))])
    assert rel_err(got.cpu(), 0.5 * xs[0] - 2.0 * xs[1] + 3.0 * xs[2] + 0.25 * xs[3]) < 1e-6
    sch = FlowDPMSolverMultistepScheduler(solver_order=order, shift=1.0)
    sch.set_timesteps(sigmas=get_sampling_sigmas(steps, 5.0), device=DEV)
    x, out = z["x0"].to(DEV), []
    for t in sch.timesteps:
        v = 0.3 * x + 0.1 * torch.sin(3.0 * x) + (float(t) / 1000.0 - 0.5)
        x = sch.step(v, t, x, return_dict=False)[0]
        out.append(x.clone())
    assert rel_err(torch.stack(out).cpu(), z[f"o{order}_s{steps}"]) < 1e-5


def test_unipc_solver_gpu():
    from more4d_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    z = load_npz("sched_unipc.npz")
    sch = FlowUniPCMultistepScheduler(solver_order=3, shift=1.0)
    sch.set_timesteps(30, device=DEV, shift=5.0)
    x, out = z["x0"].to(DEV), []
    for t in sch.timesteps:
        v = 0.3 * x + 0.1 * torch.sin(3.0 * x) + (float(t) / 1000.0 - 0.5)
        x = sch.step(v, t, x, return_dict=False)[0]
        out.append(x.clone())
    assert rel_err(torch.stack(out).cpu(), z["o3_lin30"]) < 1e-5


def test_omnimae_vit_patch_features():
    """OmniMAE ViT-B front end on the HIP kernels == the reference's VisionTransformer.forward_patch_features
    (tests/golden/omnimae.npz): fp32 at 1e-3, bf16 within the bf16 budget; then the DiT called with `first_frame`
    (ViT -> adapter -> guidance) equals the DiT called with the precomputed features."""
    from more4d_amd.models import WanTransformer4DModel
    from more4d_amd.models.omnimae import vit_base_mae_pretraining
    z = load_npz("omnimae.npz")
    sd = fill(load_keys("omnimae_keys.json"), 555)
    for dtype, tol in ((torch.float32, TOL), (torch.bfloat16, 3e-2)):
        m = vit_base_mae_pretraining(pretrained=False)
        m.load_state_dict(sd, strict=False)
        m = m.to(DEV, dtype).eval()
        feats, cls = m.trunk.forward_patch_features(z["frame"].to(DEV), None, normalize=True)
        assert feats.dtype == torch.float32 and feats.shape == (2, 196, 768)
        err = rms_rel_err(feats.cpu(), z["feats"]) if dtype == torch.bfloat16 else rel_err(feats.cpu(), z["feats"])
        print("omnimae", dtype, err)
        assert err < tol and rel_err(cls.cpu(), feats[:, 0].cpu()) == 0
    zt = load_npz("dit_tiny.npz")
    g = WanTransformer4DModel(**dict(TINY, use_omnimae_guidance=True))
    g.load_state_dict(fill(load_keys("dit_tiny_guid_keys.json"), 4321), strict=False)
    g.omnimae_extractor.load_state_dict(sd, strict=False)
    g = g.to(DEV, torch.float32).eval()
    kw = dict(x=zt["x"].to(DEV), t=zt["t"].to(DEV), context=[zt["ctx0"].to(DEV), zt["ctx1"].to(DEV)],
              seq_len=int(zt["seq_len_pad"]), clip_fea=zt["clip"].to(DEV), y=zt["y"].to(DEV), full_ref=zt["full_ref"].to(DEV))
    with torch.no_grad():
        a = g(**kw, first_frame=z["frame"].to(DEV))
        ff = g.omnimae_extractor.trunk.forward_patch_features(z["frame"].to(DEV), None, normalize=True)
        b = g(**kw, first_frame_features=ff)
        c = g(**kw)
    assert torch.equal(a, b) and rel_err(a.cpu(), c.cpu()) > 1e-4      # guidance is on and changes the prediction


def test_token_sharded_local_first_schedule_single_gpu():
    """The T-sharded forward's local-first self-attention (local shard, then the gathered remote shards, merged through their
    log-sum-exps) on ONE GPU: a stand-in group whose all-gather hands every rank slot the right shard of a reference run."""
    from more4d_amd.dist import SequenceParallelGroup
    from more4d_amd.models.wan_transformer4d import WanSelfAttention, _Ctx, build_rope_tables, rope_params
    torch.manual_seed(0)
    B, W, Ls, C, heads = 2, 4, 264, 256, 2
    d = C // heads
    L = W * Ls - 19                                       # the last shard is ragged: 19 pad keys masked
    sa = WanSelfAttention(C, heads).to(DEV).eval()
    with torch.no_grad():
        for p_ in sa.parameters():
            p_.normal_(0, 0.05)
        sa.norm_q.weight.add_(1.0)
        sa.norm_k.weight.add_(1.0)
    freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)), rope_params(1024, 2 * (d // 6))], dim=1)
    grid = (W, 12, 22)
    cos, sin = build_rope_tables(freqs, grid, d, DEV)
    xn = torch.randn(B, W * Ls, C, device=DEV)
    gate = torch.ones(B, C, device=DEV)
    with torch.no_grad():
        full = torch.zeros(B, W * Ls, C, device=DEV)
        sa.run(xn, full, gate, C, _Ctx(B, L, W * Ls, grid, cos, sin, L, {}, L))

        class Stub(SequenceParallelGroup):
            def __init__(self, rank, shards):
                self.group, self.world_size, self.rank, self.shards, self.calls = None, W, rank, shards, 0

            def gather_start(self, x):
                buf = torch.stack(self.shards[self.calls])
                assert torch.equal(buf[self.rank], x)      # this rank's K / V^T are exactly what the peers would receive
                self.calls += 1
                return buf, None, x

        # per-rank K and V^T shards from independent single-rank evaluations of the projections
        shards = [[], []]
        for r in range(W):
            xr = xn[:, r * Ls:(r + 1) * Ls].contiguous()
            from more4d_amd import ops
            k = ops.gemm_bt(xr.view(-1, C), sa.k.weight, sa.k.bias)
            ops.rmsnorm_rope(k, sa.norm_k.weight.float(), head_dim=d, eps=sa.eps, cos=cos, sin=sin, rows_per_sample=Ls,
                             rope_len=max(0, min(Ls, L - r * Ls)), pos_offset=r * Ls)
            shards[0].append(k)
            shards[1].append(ops.gemm_bt(sa.v.weight, xr.view(-1, C), sa.v.bias, bias_on_m=True))
        for local_first in (True, False):
            for r in range(W):
                sp = Stub(r, shards)
                sp.local_first = local_first
                out = torch.zeros(B, Ls, C, device=DEV)
                c = _Ctx(B, L, Ls, grid, cos, sin, max(0, min(Ls, L - r * Ls)), {}, L, sp, r * Ls)
                sa.run(xn[:, r * Ls:(r + 1) * Ls].contiguous(), out, gate, C, c)
                assert rel_err(out.cpu(), full[:, r * Ls:(r + 1) * Ls].cpu()) < 1e-5, (local_first, r)


def test_pipeline_call_end_to_end():
    """WanFunControlPipeline.__call__ (reference :477-858) with the tiny DiT and the full-size VAE network on small frames:
    control / depth / reference-image VAE encodes -> 48-channel control input -> CFG denoise loop -> decode.  The result must
    equal the same stages composed by hand (conditioning order [control | start-image zeros | depth], :762-777; ref latent =
    frame 0, :722), and the decoded video stays in [-1, 1]."""
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    from more4d_amd.pipeline import WanFunControlPipeline, denoise_latents
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
    m = tiny_model()
    vae = AutoencoderKLWan().eval()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    vae = vae.to(DEV, torch.float32)
    g = torch.Generator().manual_seed(5)
    F_, H_, W_ = 5, 32, 32
    control = torch.rand(1, 3, F_, H_, W_, generator=g) * 2 - 1
    depth = torch.rand(1, 3, 1, H_, W_, generator=g) * 2 - 1
    ref = torch.rand(1, 3, 1, H_, W_, generator=g) * 2 - 1
    lat0 = torch.randn(1, 16, 2, 4, 4, generator=g)
    pe, ne = [torch.randn(9, 64, generator=g)], [torch.randn(3, 64, generator=g)]
    clip = torch.randn(1, 257, 1280, generator=g)
    pipe = WanFunControlPipeline(vae=vae, transformer=m, scheduler=FlowDPMSolverMultistepScheduler(solver_order=1, shift=1.0))
    kw = dict(height=H_, width=W_, control_video=control, ref_image=ref, depth_image=depth, num_frames=F_, num_inference_steps=3,
              guidance_scale=6.0, latents=lat0, prompt_embeds=[p.to(DEV) for p in pe], negative_prompt_embeds=[p.to(DEV) for p in ne],
              clip_context=clip.to(DEV), shift=5)
    out = pipe(output_type="latent", **kw).videos
    with torch.no_grad():
        enc = lambda v: vae.encode(v.to(DEV))[0].mode()
        ctrl, dl, rl = enc(control), enc(depth.repeat(1, 1, F_, 1, 1)), enc(ref)[:, :, 0]
        y = torch.cat([ctrl, torch.zeros_like(ctrl), dl], dim=1)
        sch = FlowDPMSolverMultistepScheduler(solver_order=1, shift=1.0)
        ts, _ = retrieve_timesteps(sch, device=DEV, sigmas=get_sampling_sigmas(3, 5))
        want = denoise_latents(m, sch, lat0, ts, 6.0, [ne[0].to(DEV), pe[0].to(DEV)], clip_fea=clip.to(DEV), y=y, full_ref=rl)
    assert out.shape == (1, 16, 2, 4, 4) and rel_err(out.cpu(), want.cpu()) < 1e-6
    vid = pipe(output_type="no_normalize", **kw).videos
    assert vid.shape == (1, 3, F_, H_, W_) and bool(torch.isfinite(vid).all()) and float(vid.abs().max()) <= 1.0
    vid01 = pipe(output_type="numpy", **kw).videos
    assert rel_err(vid01, (vid / 2 + 0.5).clamp(0, 1)) < 1e-6
