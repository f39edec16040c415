"""Round-4 GPU parity (VERDICT r3 "next round" items 1 and 9):

* bf16 budgets CALIBRATED against the reference's own bf16-autocast run (tests/golden/bf16_calibration.json,
  tests/golden/make_golden_r4.py) instead of hand-picked constants — see util.bf16_budget;
* four stacked 14B-width blocks at L = 2080 through the production bf16 kernels vs the reference;
* the production GEMM's (gemm_bt256w) ACCUMULATION against float64 on the same bf16 operands: bit-exact on integer-valued
  operands, < 0.5 % flipped bf16 roundings on N(0,1) operands — what the 8e-3 op-level budget (output rounding) cannot say;
* one 14B-width block forward + BACKWARD: at L = 2080 against gradients produced by the reference, at L = 21 840 (BASELINE
  configs[4]'s per-GPU sequence) against fp32 torch autograd of the oracle block on the device;
* configs[4]'s per-GPU work as a test: 2 layers at 14B width, L = 21 840, forward + backward + clip + AdamW through
  training.train_step, stored activations == recompute;
* t2v_cross_attn / cross_attn blocks, per-token timesteps, qk_norm=False against reference fixtures."""
import math

import pytest
import torch

from util import bf16_budget, grad_sample, load_keys, load_npz, rel_err, rms_rel_err
from weights import block_shapes, fill, randn_named

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
TINY = dict(model_type="i2v", in_dim=64, dim=128, ffn_dim=512, num_heads=4, num_layers=2, text_dim=64, text_len=32,
            freq_dim=256, out_dim=16, add_ref_conv=True, use_dino_guidance=False, cross_attn_norm=True)


def gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


def make_block_14b(dtype, seed=0):
    from more4d_amd.models import WanAttentionBlock
    blk = WanAttentionBlock("i2v_cross_attn", 5120, 13824, 40, (-1, -1), True, True, 1e-6, use_spatial_guidance=False)
    sd = {k[len("blocks.0."):]: v for k, v in fill(block_shapes(5120, 13824, False), seed).items()}
    blk.load_state_dict(sd, strict=True)
    return blk.to(DEV, dtype).eval()


def _freqs(d=128):
    from more4d_amd.models.wan_transformer4d import rope_params
    return torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)), rope_params(1024, 2 * (d // 6))], dim=1)


def _long_inputs():
    L = 2080
    return (L, (4, 20, 26), randn_named("in.x", (1, L, 5120), 6), randn_named("in.e0", (1, 6, 5120), 6, 0.2),
            randn_named("in.ctx", (1, 257 + 512, 5120), 6))


def _block_metrics(out, x, z):
    rows = z["rows"].long()
    delta = out - x
    return dict(delta_max=float((delta[rows] - z["delta_rows"]).abs().max() / z["delta_rows"].abs().max()),
                delta_rms=rms_rel_err(delta[rows], z["delta_rows"]), delta_norm=rel_err(delta.norm(dim=-1), z["delta_norm"]),
                out_rms=rms_rel_err(out[rows], z["out_rows"]))


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["fp32", "bf16"])
def test_stack_of_four_14b_blocks_vs_reference(dtype):
    """Error growth over STACKED 14B-width layers: four blocks with different weights at L = 2080 — every
    projection on gemm_bt256w, the self-attention on attn128p in bf16 — against the reference's fp32 output of the same stack
    (wan_transformer4d.py:633-688, four times).  fp32: 1e-3.  bf16: 1.5 x the reference's own bf16-autocast error."""
    from more4d_amd import ops
    z = load_npz("dit_stack4_14b_long.npz")
    L, grid, x, e0, ctx = _long_inputs()
    y = x
    ops.launch_counts(reset=True)
    with torch.no_grad():
        for layer in range(4):
            blk = make_block_14b(dtype, layer)
            y = blk(y, e0, torch.tensor([L]), torch.tensor([list(grid)]), _freqs(), ctx.to(dtype), None, dtype=torch.float32, t=0)
            del blk
    counts = ops.launch_counts()
    out = y.float().cpu()[0]
    got = _block_metrics(out, x[0], z)
    print("stack4", dtype, got)
    if dtype == torch.float32:
        assert rel_err(out[z["rows"].long()], z["out_rows"]) < 1e-3 and rel_err(out.norm(dim=-1), z["row_norm"]) < 1e-3
    else:
        assert counts["gemm_wide"] + counts["gemm_phased"] >= 32 and counts["attn_q64"] + counts["attn_phased"] >= 4, counts
        for k, v in got.items():
            assert v <= bf16_budget("stack4_14b_long", k), (k, v, bf16_budget("stack4_14b_long", k))


@pytest.mark.parametrize("N,K", [(5120, 5120), (13824, 5120), (5120, 13824)])
def test_production_gemm_accumulation_vs_fp64(N, K):
    """gemm_bt256w at the three bench shapes (M = 43 680 = the CFG pair's rows, ragged last tile row), fp32-store epilogue
    (out = round_T(acc + bias) like the reference's autocast Linear, then widened): the 8e-3 budget of the op-level tests is
    dominated by that output rounding and says little about the ACCUMULATION.  Two checks that do:
    (a) integer-valued operands whose exact result is representable in bf16: every output must equal the float64 result bit
    for bit — no K-slice dropped, duplicated or mis-paired anywhere in the 256 x 256 x 64 tiling, the persistent hand-over or the
    ragged last tile row; (b) N(0,1) operands against round_bf16(float64 result): an fp32 accumulation with relative error eps flips
    the bf16 rounding of a fraction ~ eps / 2^-8 of the outputs — require < 0.5 % flipped (eps ~ 1e-5) and each flip to be one ulp."""
    from more4d_amd import ops
    from more4d_amd.ops import EPI_STORE_F32
    M = 43680
    rows = torch.tensor([0, 1, 255, 256, 21839, 21840, 43519, 43520, 43679], device=DEV)
    # (a) entries in {-1, 0, 1} (P(nonzero) = 1/4 each side): |sum| stays below 256 (8 sigma at K = 13 824), integers are exact in bf16
    ai = (torch.randint(0, 8, (M, K), device=DEV, generator=gen(1)) == 0).float() - (torch.randint(0, 8, (M, K), device=DEV, generator=gen(2)) == 0).float()
    wi = (torch.randint(0, 8, (N, K), device=DEV, generator=gen(3)) == 0).float() - (torch.randint(0, 8, (N, K), device=DEV, generator=gen(4)) == 0).float()
    bi = torch.randint(-3, 4, (N,), device=DEV, generator=gen(5)).float()
    ops.launch_counts(reset=True)
    out = ops.gemm_bt(ai.to(BF), wi.to(BF), bi.to(BF), epilogue=EPI_STORE_F32)
    c = ops.launch_counts()
    assert c["gemm_wide"] + c["gemm_phased"] == 1 and c["gemm_generic"] == 0, c
    assert out.dtype == torch.float32
    ref = ai[rows].double() @ wi.double().t() + bi.double()
    assert float(ref.abs().max()) < 256
    assert torch.equal(out[rows].double(), ref)
    # every row, cheaply: row sums of the exact integer result (|sum| < 2^24: exact in fp32 order-independently in float64)
    assert torch.equal(out.double().sum(1), (ai.double() @ wi.double().sum(0)) + bi.double().sum())
    del ai, wi, out
    # (b)
    a = torch.randn(M, K, device=DEV, generator=gen(6)).to(BF)
    w = (torch.randn(N, K, device=DEV, generator=gen(7)) * K ** -0.5).to(BF)
    b = (torch.randn(N, device=DEV, generator=gen(8)) * 0.1).to(BF)
    out = ops.gemm_bt(a, w, b, epilogue=EPI_STORE_F32)
    ref = a[rows].double() @ w.double().t() + b.double()
    want = ref.float().to(BF).float()
    got = out[rows]
    flipped = got != want
    frac = float(flipped.float().mean())
    print("gemm accumulation", N, K, "flipped roundings:", frac)
    assert frac < 5e-3, frac
    # a flip is one bf16 ulp of the value (near zero the fp32 accumulation error itself, ~1e-6 of the row's scale, is many ulps)
    assert bool(((got - want).abs() <= want.abs() * 2 ** -7 + 1e-5 * float(ref.abs().max())).all())
    assert float((got.double() - ref).abs().max() / ref.abs().max()) < 2 ** -8              # never worse than half an ulp of the largest value


def _block_ctx(blk, x, e0, ctx, grid, dres=None):
    """Drive autograd.block_backward directly (what BlockFn does): forward half, then the backward with the stash."""
    from more4d_amd.autograd import block_backward
    from more4d_amd.models.wan_transformer4d import _Ctx, _round8, build_rope_tables
    B, L, C = x.shape
    T = blk.ffn[0].weight.dtype
    Lp = _round8(L)
    x0 = torch.zeros((B, Lp, C), device=DEV, dtype=torch.float32)
    x0[:, :L] = x.to(DEV)
    cos, sin = build_rope_tables(_freqs(blk.self_attn.head_dim), grid, blk.self_attn.head_dim, DEV)
    c = _Ctx(B, L, Lp, grid, cos, sin, min(L, grid[0] * grid[1] * grid[2]), {}, L)

    def padded(src):
        S = src.shape[1]
        o = torch.zeros((B, _round8(S), C), device=DEV, dtype=T)
        o[:, :S] = src.to(DEV)
        return o, S
    img, il = padded(ctx[:, :257])
    txt, tl = padded(ctx[:, 257:])
    e0 = e0.to(DEV).float().contiguous()
    out, stash = block_backward(blk, x0, e0, c, txt, tl, img, il, None, forward_only=True)
    if dres is None:
        return out[:, :L], None
    d = torch.zeros((B, Lp, C), device=DEV, dtype=torch.float32)
    d[:, :L] = dres.to(DEV)
    de, dtxt, dimg, G = block_backward(blk, x0, e0, c, txt, tl, img, il, d, saved=stash)
    G = dict(G)
    G["x"] = d[:, :L]
    return out[:, :L], G


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["fp32", "bf16"])
def test_block_14b_backward_vs_reference_gradients(dtype):
    """One 14B-width block at L = 2080, forward + backward for a fixed seeded cotangent, against gradients the REFERENCE produced
    (torch autograd through wan_transformer4d.py:633-688; tests/golden/dit_block_14b_long_grads.npz: 4096 sampled values + the
    norm of every parameter gradient and of dL/dx).  In bf16 the production kernels are on the path: gemm_bt256w for dgrad /
    wgrad (token axis as K), attn_bwd128 for the attention backward.  fp32: 1e-3.  bf16: rms error <= 1.5 x the error of the
    reference's own bf16-autocast gradients (per tensor, floor 2e-3)."""
    from more4d_amd import ops
    z = load_npz("dit_block_14b_long_grads.npz")
    L, grid, x, e0, ctx = _long_inputs()
    r = randn_named("cot.y", (1, L, 5120), 6)
    blk = make_block_14b(dtype)
    ops.launch_counts(reset=True)
    _, G = _block_ctx(blk, x, e0, ctx.to(dtype), grid, dres=r)
    counts = ops.launch_counts()
    names = [k[5:] for k in z if k.startswith("grad/")]
    assert len(names) > 30
    gmax = max(float(z["grad/" + n].abs().max()) for n in names)
    worst = {}
    for n in names:
        g = G[n].detach().float().cpu().reshape(-1) if n != "x" else G["x"].float().cpu()
        ref = z["grad/" + n]
        s = grad_sample(g.reshape(-1))
        scale = max(float(ref.abs().max()), 1e-3 * gmax)
        e_max = float((s.double() - ref.double()).abs().max()) / scale
        e_rms = float((s.double() - ref.double()).pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt().clamp_min(1e-3 * gmax))
        e_nrm = abs(float(g.norm()) - float(z["norm/" + n])) / max(float(z["norm/" + n]), 1e-30)
        worst[n] = (e_max, e_rms, e_nrm)
        if dtype == torch.float32:
            assert e_max < 1e-3 and e_nrm < 1e-3, (n, e_max, e_nrm)
        else:
            lim = max(bf16_budget("block_14b_long_grads", n, "rms"), 2e-3)
            assert e_rms <= lim, (n, e_rms, lim)
    print("block grads", dtype, sorted(worst.items(), key=lambda kv: -kv[1][1])[:5])
    if dtype == BF:
        assert counts["attn_bwd128"] >= 3, counts


def test_block_14b_forward_backward_full_length_vs_oracle_autograd():
    """BASELINE configs[4]'s per-GPU sequence, L = 21 840 (B = 1): one 14B-width block, forward + backward through the production
    bf16 kernels, against fp32 torch AUTOGRAD of the oracle block (oracle/dit.py:block_forward, pinned to the reference at smaller
    sizes) evaluated on the device — the attention of the oracle is run eight heads at a time under activation checkpointing so
    the [40, L, L] score tensors fit (same arithmetic per head).  Every row of the output and of dL/dx and every parameter
    gradient is compared (rms); budgets: 1.5 x the reference's own bf16-vs-fp32 error at L = 2080 (rms errors of sums of
    independent roundings do not grow with L), floor 2e-3."""
    import oracle.dit as od
    from torch.utils.checkpoint import checkpoint
    L, grid = 21840, (14, 30, 52)
    C = 5120
    x = torch.randn(1, L, C, device=DEV, generator=gen(11))
    e0 = torch.randn(1, 6, C, device=DEV, generator=gen(12)) * 0.2
    ctx = torch.randn(1, 257 + 512, C, device=DEV, generator=gen(13))
    r = torch.randn(1, L, C, device=DEV, generator=gen(14))
    blk = make_block_14b(BF)
    out, G = _block_ctx(blk, x, e0, ctx.to(BF), grid, dres=r)
    out = out.clone()
    G = {k: v.detach().float().clone() for k, v in G.items()}
    del blk
    torch.cuda.empty_cache()
    # ---- oracle: fp32 weights of the same recipe, autograd
    sd = {k: v.to(DEV).requires_grad_(True) for k, v in fill(block_shapes(5120, 13824, False), 0).items()}
    cfg = od.DiTConfig(dim=5120, ffn_dim=13824, num_heads=40, num_layers=1)
    orig = od.sdpa

    def sdpa_by_heads(q, k, v, k_len=None):
        outs = [checkpoint(orig, q[:, :, h:h + 8], k[:, :, h:h + 8], v[:, :, h:h + 8], k_len, use_reentrant=False)
                for h in range(0, q.shape[2], 8)]
        return torch.cat(outs, dim=2)
    od.sdpa = sdpa_by_heads
    try:
        with torch.device(DEV):
            xg = x.clone().requires_grad_(True)
            with torch.enable_grad():
                y = od.block_forward(sd, 0, cfg, xg, e0, grid, ctx)
                (y * r).sum().backward()
    finally:
        od.sdpa = orig
    e_out = rms_rel_err((out - x).cpu(), (y.detach() - x).cpu())
    print("full-length block: update rms err", e_out)
    assert e_out <= max(bf16_budget("block_14b_long", "delta_rms"), 2e-3), e_out
    errs = {"x": rms_rel_err(G["x"].cpu(), xg.grad.cpu())}
    for k, p in sd.items():
        n = k[len("blocks.0."):]
        errs[n] = rms_rel_err(G[n].reshape(p.shape).cpu(), p.grad.cpu())
    print("full-length block grads (worst):", sorted(errs.items(), key=lambda kv: -kv[1])[:6])
    gmax = max(float(p.grad.abs().max()) for p in sd.values())
    for n, e in errs.items():
        ref_g = xg.grad if n == "x" else sd["blocks.0." + n].grad
        if float(ref_g.abs().max()) < 1e-3 * gmax:      # cancellation residue (e.g. the key bias: analytically zero)
            continue
        # (key biases: their gradient is what is left after the softmax cancels a common shift of all keys of a query — a sum of
        # L nearly cancelling terms whose relative error grows with L; the reference's own bf16 run is 2.4 % / 17 % / 16 % off at
        # L = 2080 for exactly these three tensors)
        factor = 3.0 if n.endswith((".k.bias", ".k_img.bias")) else 1.5
        lim = max(bf16_budget("block_14b_long_grads", n, "rms", factor=factor), 2e-3)
        assert e <= lim, (n, e, lim)


def test_train_step_configs4_per_gpu_work_two_layers():
    """BASELINE configs[4] per GPU (train_wan.py:1891-2015): DiT forward + backward + adaptive clip + AdamW at batch 1,
    49 x 480 x 832 latents (L = 21 840 with the ref row), bf16 — two layers at 14B width so the test fits the suite's time.
    (a) stored activations == recompute: the same step with activation_budget_gb = 0 (the reference's per-block gradient
    checkpointing) and with every block's GEMM / attention outputs kept gives bit-identical gradients for deterministic tensors
    and agrees to atomic-order noise elsewhere; (b) the optimizer applied torch.optim.AdamW's update: sampled entries of q / o /
    ffn_down weights recomputed in fp32 from the captured gradients, clip coefficient included; (c) a second step runs on the
    updated weights and the loss stays finite."""
    import bench
    from more4d_amd.optim import AdamW, grad_norm
    from more4d_amd.training import adaptive_max_grad_norm, train_step
    cfg = dict(bench.CFG_14B)
    cfg["num_layers"] = 2
    m = bench.build_model(cfg, torch.device(DEV), BF).train()
    for p in m.parameters():
        p.requires_grad_(True)
    g = gen(4)
    F_, H_, W_ = 13, 60, 104
    lat = torch.randn(1, 16, F_, H_, W_, generator=g, device=DEV)
    noise = torch.randn(1, 16, F_, H_, W_, generator=g, device=DEV)
    fk = dict(context=[torch.randn(512, 4096, generator=g, device=DEV)], seq_len=F_ * (H_ // 2) * (W_ // 2),
              clip_fea=torch.randn(1, 257, 1280, generator=g, device=DEV), y=torch.randn(1, 48, F_, H_, W_, generator=g, device=DEV).to(BF),
              full_ref=torch.randn(1, 16, H_, W_, generator=g, device=DEV).to(BF))
    sig = torch.tensor([0.7], device=DEV)
    names = ["blocks.1.self_attn.q.weight", "blocks.1.self_attn.o.weight", "blocks.1.ffn.2.weight", "blocks.0.ffn.0.weight",
             "blocks.0.modulation", "head.head.weight"]
    params = dict(m.named_parameters())

    def grads(budget):
        from more4d_amd.training import add_noise, custom_mse_loss
        m.activation_budget_gb = budget
        m.zero_grad(set_to_none=True)
        noisy, target = add_noise(lat, noise, sig)
        pred = m(x=noisy.to(BF), t=sig * 1000.0, **fk)
        loss = custom_mse_loss(pred, target)
        loss.backward()
        return float(loss), {n: params[n].grad.detach().float().clone() for n in names}, (m.last_stored_blocks, m.last_full_blocks)
    l0, g0, st0 = grads(0.0)
    l1, g1, st1 = grads(64.0)
    assert st0 == (0, 0) and st1[0] == 2, (st0, st1)          # recompute everywhere vs both blocks stored
    # the recomputing forward is WanAttentionBlock.run (gated residual fused into the GEMM epilogue), the storing forward is
    # block_backward's forward half (separate resid_gate pass): same values up to one fp32 rounding of the residual stream, which
    # moves a few bf16 roundings downstream
    # (measured 3e-6 .. 1.3e-5 depending on which bf16 roundings the last-bit difference happens to move)
    assert math.isfinite(l0) and abs(l0 - l1) <= 5e-5 * abs(l0), (l0, l1)
    for n in names:
        assert float(g0[n].abs().max()) > 0
        # two bf16 evaluations whose residual streams differ in the last fp32 bit: they differ from each other like each differs
        # from fp32 (bf16_calibration.json, block_14b_long_grads: 0.4 % .. 1.2 % rms per tensor)
        e = rms_rel_err(g1[n], g0[n])
        assert e < 2e-2, (n, e)
    # ---- one real step through training.train_step, AdamW recomputed on sampled entries
    hp = dict(lr=2e-5, weight_decay=3e-2, eps=1e-10)
    opt = AdamW(m.parameters(), **hp)
    m.activation_budget_gb = None
    before = {n: params[n].detach().float().clone() for n in names}
    m.zero_grad(set_to_none=True)
    loss, total, actual = train_step(m, opt, latents=lat, noise=noise, sigmas=sig, timesteps=sig * 1000.0, forward_kwargs=fk,
                                     global_step=0, max_grad_norm=0.05)
    assert math.isfinite(float(loss)) and total is not None and total > 0
    assert actual == adaptive_max_grad_norm(total, 0.05, 5.0, 1000, 0)
    coef = min(1.0, actual / (total + 1e-6))
    b1, b2 = 0.9, 0.999
    for n in names:
        gg = g1[n].to(BF).float() * coef            # the step's gradients are the ones just compared (same inputs), bf16 like p.grad
        mhat = (1 - b1) * gg / (1 - b1)
        vhat = ((1 - b2) * gg * gg / (1 - b2)).sqrt() + hp["eps"]
        want = before[n] * (1 - hp["lr"] * hp["weight_decay"]) - hp["lr"] * mhat / vhat
        got = params[n].detach().float()
        moved = (got - before[n]).abs()
        # bf16 parameters: the update (2e-5 relative) is below half an ulp for most entries; compare where the exact result rounds
        # to a different bf16 value than the old one, and require that nothing moved by more than one ulp elsewhere
        exp_bf = want.to(BF).float()
        assert float((got - exp_bf).abs().max()) <= float(before[n].abs().max()) * 2 ** -7, n
        assert float(moved.max()) > 0 or float((exp_bf - before[n]).abs().max()) == 0, n
    loss2, total2, _ = train_step(m, opt, latents=lat, noise=noise, sigmas=sig, timesteps=sig * 1000.0, forward_kwargs=fk,
                                  global_step=1, max_grad_norm=0.05)
    assert math.isfinite(float(loss2)) and total2 is not None
    del grad_norm


# ------------------------------------------------------------------ sub-branches of rows a6 / a11 (VERDICT r3 missing #2, #4, weak #4)
def _small_block(name, norm3, qk_norm, z, prefix, dtype):
    from more4d_amd.models import WanAttentionBlock
    blk = WanAttentionBlock(name, 128, 512, 4, (-1, -1), qk_norm, norm3, 1e-6, use_spatial_guidance=False)
    blk.load_state_dict({k[len(prefix):]: v for k, v in z.items() if k.startswith(prefix)}, strict=True)
    return blk.to(DEV, dtype).eval()


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["fp32", "bf16"])
@pytest.mark.parametrize("name,norm3", [("t2v_cross_attn", True), ("cross_attn", False)])
def test_block_t2v_and_plain_cross_attention_vs_reference(name, norm3, dtype):
    """WanT2VCrossAttention / WanCrossAttention (wan_transformer4d.py:468-497, 558-575) inside a WanAttentionBlock: the whole
    context is text (no 257-token image split, no k_img / v_img), with and without the affine norm3 (:652-654)."""
    z = load_npz("dit_block_xattn.npz")
    blk = _small_block(name, norm3, True, z, f"{name}/w/", dtype)
    assert not hasattr(blk.cross_attn, "k_img")
    x, e0, ctx, grid = z[f"{name}/x"], z[f"{name}/e0"], z[f"{name}/ctx"], z[f"{name}/grid"]
    with torch.no_grad():
        out = blk(x, e0, torch.tensor([x.shape[1]]), grid.view(1, 3), _freqs(32), ctx.to(dtype), None, dtype=torch.float32, t=0)
    ref = z[f"{name}/out"]
    if dtype == torch.float32:
        assert rel_err(out.cpu(), ref) < 1e-3
    else:
        assert rms_rel_err((out.float().cpu() - x), (ref - x)) < 1.5e-2


@pytest.mark.parametrize("dtype", [torch.float32, BF], ids=["fp32", "bf16"])
def test_block_without_qk_norm_vs_reference(dtype):
    """qk_norm=False (wan_transformer4d.py:431-432: norm_q / norm_k / norm_k_img are nn.Identity): RoPE without the RMS norm
    (m4d_rmsnorm_rope with NULL weights), cross-attention q / k straight from their projections."""
    import torch.nn as nn
    z = load_npz("dit_block_noqknorm.npz")
    blk = _small_block("i2v_cross_attn", True, False, z, "w/", dtype)
    assert isinstance(blk.self_attn.norm_q, nn.Identity) and isinstance(blk.cross_attn.norm_k_img, nn.Identity)
    with torch.no_grad():
        out = blk(z["x"], z["e0"], torch.tensor([z["x"].shape[1]]), z["grid"].view(1, 3), _freqs(32), z["ctx"].to(dtype), None,
                  dtype=torch.float32, t=0)
    if dtype == torch.float32:
        assert rel_err(out.cpu(), z["out"]) < 1e-3
    else:
        assert rms_rel_err((out.float().cpu() - z["x"]), (z["out"] - z["x"])) < 1.5e-2


def _tiny_model(dtype=torch.float32):
    from more4d_amd.models import WanTransformer4DModel
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234), strict=True)
    return m.to(DEV, dtype).eval()


def test_per_token_timesteps_vs_reference():
    """t [B, seq_len] (wan_transformer4d.py:1161-1167): e [B, L, C] feeds the head's `e.dim() > 2` branch (:713-715), e0
    [B, L, 6, C] every block's `e.dim() > 3` branch (:655-657) — one modulation / gate vector per token, with the ref row and
    seq_len padding, and without."""
    z, p = load_npz("dit_tiny.npz"), load_npz("dit_tiny_pertoken.npz")
    m = _tiny_model()
    ctx = [z["ctx0"].to(DEV), z["ctx1"].to(DEV)]
    with torch.no_grad():
        out = m(x=z["x"].to(DEV), t=p["t_tok"].to(DEV), context=ctx, seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"].to(DEV),
                y=z["y"].to(DEV), full_ref=z["full_ref"].to(DEV))
        assert rel_err(out.cpu(), p["out_ref"]) < 1e-3
        out = m(x=z["x"].to(DEV), t=p["t_tok_noref"].to(DEV), context=ctx, seq_len=int(z["seq_len"]), clip_fea=z["clip"].to(DEV),
                y=z["y"].to(DEV), full_ref=None)
        assert rel_err(out.cpu(), p["out_noref"]) < 1e-3
        # a per-token t that is constant along the sequence is the per-sample call
        tc = z["t"].view(-1, 1).expand(-1, int(z["seq_len"])).contiguous()
        a = m(x=z["x"].to(DEV), t=tc.to(DEV), context=ctx, seq_len=int(z["seq_len"]), clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV))
        assert rel_err(a.cpu(), z["out_noref"]) < 1e-3
        with pytest.raises(ValueError):
            m(x=z["x"].to(DEV), t=tc[:, :-1].to(DEV), context=ctx, seq_len=int(z["seq_len"]), clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV))


def test_tiny_loop_bf16_calibrated():
    """BASELINE configs[0] in the production dtype: the 50-step CFG / Euler loop of the tiny DiT in bf16 against the reference's
    fp32 final latent, within 1.5 x of what the reference's own bf16-autocast loop loses."""
    from more4d_amd.pipeline import denoise_latents
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
    z = load_npz("loop_tiny.npz")
    m = _tiny_model(BF)
    sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    ts, _ = retrieve_timesteps(sch, device=DEV, sigmas=get_sampling_sigmas(int(z["steps"]), float(z["shift"])))
    with torch.no_grad():
        out = denoise_latents(m, sch, z["lat"], ts, float(z["guidance"]), [z["ctx_u"].to(DEV), z["ctx_c"].to(DEV)],
                              clip_fea=z["clip"], y=z["y"], full_ref=z["full_ref"], seq_len=16 * 16)
    e_max, e_rms = rel_err(out.float().cpu(), z["final"]), rms_rel_err(out.float().cpu(), z["final"])
    print("loop bf16", e_max, e_rms)
    assert e_rms <= bf16_budget("loop_tiny", "rms") and e_max <= bf16_budget("loop_tiny", "max", factor=2.0)


def test_gemm_persistent_sync_poll_path_and_two_streams():
    """the XCD-wide tile rounds of the persistent GEMM only poll when a launch has >= 4 x CUs tiles; M = 21 840,
    N = 5120 is 86 x 20 = 1 720 tiles.  The result must not depend on the pacing hint: two launches give the same bits, and so do
    two launches in flight on two streams that share the arrival counters."""
    from more4d_amd import ops
    a = torch.randn(21840, 512, device=DEV, generator=gen(5)).to(BF)
    w = (torch.randn(5120, 512, device=DEV, generator=gen(6)) * 512 ** -0.5).to(BF)
    b = torch.randn(5120, device=DEV, generator=gen(7)).to(BF)
    ref = ops.gemm_bt(a, w, b)
    assert torch.equal(ops.gemm_bt(a, w, b), ref)
    rows = torch.tensor([0, 255, 256, 21839], device=DEV)
    want = (a[rows].float() @ w.float().t() + b.float())
    assert rel_err(ref[rows].float().cpu(), want.cpu()) < 8e-3
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(3):
        with torch.cuda.stream(s1):
            o1 = ops.gemm_bt(a, w, b)
        with torch.cuda.stream(s2):
            o2 = ops.gemm_bt(a, w, b)
        outs += [o1, o2]
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)


@pytest.mark.gpu
@pytest.mark.parametrize("C,H,W", [(128, 40, 72), (64, 33, 40), (256, 24, 48)])
def test_groupnorm_planar_store_path_equals_channels_last(C, H, W):
    """groupnorm_apply_kernel's planar-16 output goes through a wave-local LDS transpose (full runs of a plane per store instruction);
    blocks that are not full and the rows behind them take the direct stores.  Both must be the channels-last result, bit for bit."""
    from more4d_amd import ops as o
    F, HW = 5, H * W
    g = torch.Generator().manual_seed(3)
    x = torch.randn(F, HW, C, generator=g).bfloat16().to(DEV)
    gw, gb = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    G = C // 8                                                                   # (channels per group must be a multiple of 4)
    ref = o.groupnorm_cl(x, gw, gb, F=F, HW=HW, groups=G, silu=True)             # [F, HW, C]
    parts = o.groupnorm_cl_planar(x, gw, gb, F=F, HW=HW, groups=G, frames_per_group=2)
    f0 = 0
    for pl in parts:
        t = pl.t                                                                 # [C/16, frames, HW, 16]
        got = t.permute(1, 2, 0, 3).reshape(t.shape[1], HW, C)
        assert torch.equal(got, ref[f0:f0 + t.shape[1]])
        f0 += t.shape[1]
    assert f0 == F


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 20, 36, 96, 96, (3, 3, 3)), (2, 16, 24, 128, 128, (1, 3, 3)), (3, 12, 20, 96, 192, (3, 3, 3)),
                                   (5, 9, 13, 384, 384, (3, 3, 3))])
def test_conv_wgrad_stacked_taps_on_the_production_gemm(shape, monkeypatch):
    """vae_autograd.conv_wgrad for Cout % 32 == 0 layers runs ALL taps of a layer as one m4d_gemm_bt_taps launch on the 256 x 256 kernel
    (dy as the shifted operand, taps stacked along M, K-slices in gridDim.y).  Against torch autograd through F.conv3d (fp32 on the
    bf16-rounded operands) and against the batched generic path (M4D_WGRAD_TAPS=0), which sums the same products in another order."""
    import torch.nn.functional as F
    from more4d_amd import ops as o
    from more4d_amd.vae_autograd import conv_wgrad
    t, h, w, ci, co, k = shape
    kt, kh, kw = k
    g = torch.Generator().manual_seed(4)
    Tin = t + kt - 1
    x = torch.randn(Tin, h, w, ci, generator=g).bfloat16()
    dy = torch.randn(t * h * w, co, generator=g).bfloat16()
    xr = x.float().permute(3, 0, 1, 2)[None]
    wr = torch.zeros(co, ci, kt, kh, kw, requires_grad=True)
    y = F.conv3d(F.pad(xr, (kw // 2, kw // 2, kh // 2, kh // 2, 0, 0)), wr)
    y.backward(dy.float().view(t, h, w, co).permute(3, 0, 1, 2)[None])
    want = wr.grad.permute(0, 2, 3, 4, 1)
    before = dict(o.launch_counts())
    got = conv_wgrad(x.to(DEV), ci, Tin, h, w, ci, dy.to(DEV), co, k, (kh // 2, kw // 2))
    after = o.launch_counts()
    assert after["gemm_wide"] > before.get("gemm_wide", 0)                      # the production kernel ran
    monkeypatch.setenv("M4D_WGRAD_TAPS", "0")
    old = conv_wgrad(x.to(DEV), ci, Tin, h, w, ci, dy.to(DEV), co, k, (kh // 2, kw // 2))
    assert rel_err(got.cpu(), want) < 2e-5 and rel_err(old.cpu(), want) < 2e-5
    assert rel_err(got.cpu(), old.cpu()) < 2e-6


@pytest.mark.gpu
def test_attention_backward_reruns_are_bit_identical_full_length():
    """The fused dK / dV pass hands P from one wave to its partner through an LDS mailbox behind a flag, with tiles landing by DMA in two
    rings: a lost hand-over or a tile read early would show as run-to-run differences.  Six reruns at L = 21 840 (ragged last query tile,
    ragged last key workgroup) under a concurrent stream of unrelated launches must agree bit for bit."""
    from more4d_amd import ops as o
    from more4d_amd.ops import KV
    L, heads, D = 21840, 8, 128
    C = heads * D
    g = torch.Generator(device=DEV).manual_seed(7)
    q, k, v = (torch.randn(L, C, device=DEV, generator=g).bfloat16() for _ in range(3))
    d_o = (torch.randn(L, C, device=DEV, generator=g) * 0.1).bfloat16()
    lse = torch.empty(1, heads, L, device=DEV)
    out = o.attention(q, [KV(k, o.transpose(v), L * C, C, L, L, L)], B=1, Lq=L, heads=heads, head_dim=D, q_bs=L * C, q_ls=C, lse=lse).view(L, C)
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    ref = None
    for it in range(6):
        dq, dk, dv = (torch.full_like(q, float("nan")) for _ in range(3))
        with torch.cuda.stream(side):
            for _ in range(4):
                junk = junk @ junk * 1e-4
        o.attention_bwd(q, k, v, out, d_o, lse, B=1, Lq=L, Lk=L, Lk_rows=L, heads=heads, head_dim=D, dq=dq, dk=dk, dv=dv)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(dq.float()).all()) and bool(torch.isfinite(dk.float()).all()) and bool(torch.isfinite(dv.float()).all())
        if ref is None:
            ref = (dq, dk, dv)
        else:
            assert torch.equal(dq, ref[0]) and torch.equal(dk, ref[1]) and torch.equal(dv, ref[2])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world", [("denoise", 2), ("denoise", 4), ("train", 2)])
def test_bench_n_ranks_on_one_gpu_over_gloo(mode, world):
    """`bench.py --gpus N` end to end with REAL processes and the real kernels: RCCL refuses two ranks on one device, so the tool mode
    M4D_BENCH_ONE_GPU=1 puts every rank on cuda:0 over gloo.  Both layouts (cfg2 x sp(N/2) and plain sp-N), the stand-in second pass
    that measures the exposed collective time, the sharded data-parallel train step: the run must finish, stay finite and print one JSON
    line that is marked invalid as a measurement."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--layers", "2", "--steps", "1", "--warmup", "1"]
    cmd += ["--mode", "train"] if mode == "train" else ["--no-secondary", "--no-cpu-baseline"]
    env = dict(os.environ, M4D_BENCH_ONE_GPU="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["rccl_ranks"] == world and d["valid"] is False and "M4D_BENCH_ONE_GPU" in d["env_overrides"]
    assert d["value"] is not None and d["value"] > 0
    if mode == "denoise":
        assert d["secondary"]["sp_layout"]["finite"] and "collectives" in d
    else:
        assert math.isfinite(d["loss"])


@pytest.mark.gpu
@pytest.mark.parametrize("Lq,Lk,Lk_rows", [(2304, 2304, 2304), (1100, 257, 264), (330, 1000, 1000)])
def test_attention_bwd_accumulate_flags_production_kernels(Lq, Lk, Lk_rows):
    """accumulate_dq / accumulate_dkv through the phased dQ kernel and the fused dK / dV kernel (bf16, head_dim 128, B = 2; ragged query and
    key tiles, padded key rows): out = previous content + gradient; padded key rows keep their previous content."""
    from more4d_amd import ops as o
    from more4d_amd.ops import KV
    B, heads, D = 2, 3, 128
    C = heads * D
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B * Lq, C, generator=g).bfloat16().to(DEV)
    k = torch.randn(B * Lk_rows, C, generator=g).bfloat16().to(DEV)
    v = torch.randn(B * Lk_rows, C, generator=g).bfloat16().to(DEV)
    d_o = torch.randn(B * Lq, C, generator=g).bfloat16().to(DEV)
    lse = torch.empty(B, heads, Lq, device=DEV)
    out = o.attention(q, [KV(k, o.transpose(v), Lk_rows * C, C, Lk_rows, B * Lk_rows, Lk)], B=B, Lq=Lq, heads=heads, head_dim=D,
                      q_bs=Lq * C, q_ls=C, lse=lse).view(B * Lq, C)
    kw = dict(B=B, Lq=Lq, Lk=Lk, Lk_rows=Lk_rows, heads=heads, head_dim=D)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    o.attention_bwd(q, k, v, out, d_o, lse, dq=dq, dk=dk, dv=dv, **kw)
    pq, pk, pv = (torch.randn(t.shape, generator=g).bfloat16().to(DEV) for t in (q, k, v))
    aq, ak, av = pq.clone(), pk.clone(), pv.clone()
    o.attention_bwd(q, k, v, out, d_o, lse, dq=aq, dk=ak, dv=av, accumulate_dq=True, accumulate_dkv=True, **kw)
    for got, prev, grad in ((aq, pq, dq), (ak, pk, dk), (av, pv, dv)):
        want = prev.float() + grad.float()
        assert rel_err(got.float(), want) < 8e-3
    pad = ak.view(B, Lk_rows, C)[:, Lk:]
    assert torch.equal(pad, pk.view(B, Lk_rows, C)[:, Lk:]) and torch.equal(av.view(B, Lk_rows, C)[:, Lk:], pv.view(B, Lk_rows, C)[:, Lk:])
