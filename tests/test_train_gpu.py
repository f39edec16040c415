"""Training-step kernels on the MI355X through the C ABI: every backward kernel against fp32 torch
autograd of the same formula (tests/cpu_ops.py stand-ins share the ops.* signatures), then the whole tiny DiT's
loss.backward() against gradients produced by the reference itself (tests/golden/dit_tiny_grads.npz)."""
import math

import pytest
import torch

import cpu_ops
from util import check_grads, same_grads, custom_mse_loss, load_keys, load_npz, rel_err
from weights import fill

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3   # north_star: 1e-3 relative fp32

TINY = dict(model_type="i2v", in_dim=64, dim=128, ffn_dim=512, num_heads=4, num_layers=2, text_dim=64, text_len=32,
            freq_dim=256, out_dim=16, add_ref_conv=True, use_dino_guidance=False, cross_attn_norm=True)


def gen(seed=0):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(64, 64), (1000, 36), (132, 516), (24, 8)])
def test_transpose(dtype, shape):
    from more4d_amd import ops
    x = torch.randn(shape, generator=gen()).to(dtype)
    assert torch.equal(ops.transpose(x.to(DEV)).cpu(), x.t())
    # strided source and destination (column slices of wider buffers)
    wide = torch.randn(shape[0], shape[1] + 8, generator=gen(1)).to(dtype).to(DEV)
    out = torch.zeros(shape[1], shape[0] + 8, dtype=dtype, device=DEV)
    ops.transpose(wide[:, 4:4 + shape[1]], out=out[:, :shape[0]])
    assert torch.equal(out[:, :shape[0]].cpu(), wide[:, 4:4 + shape[1]].t().cpu()) and float(out[:, shape[0]:].abs().sum()) == 0


@pytest.mark.parametrize("ta,tb", [(torch.float32, None), (torch.bfloat16, None), (torch.float32, torch.bfloat16),
                                   (torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32)])
def test_colsum(ta, tb):
    from more4d_amd import ops
    R, C, rpg = 1000, 260, 500
    a = torch.randn(R, C, generator=gen()).to(ta)
    b = torch.randn(R, C, generator=gen(1)).to(tb) if tb is not None else None
    ref = cpu_ops.colsum(a, b, rows_per_group=rpg)
    got = ops.colsum(a.to(DEV), b.to(DEV) if b is not None else None, rows_per_group=rpg)
    assert got.shape == (2, C) and rel_err(got.cpu(), ref) < 1e-5
    one = ops.colsum(a.to(DEV))
    assert rel_err(one.cpu(), a.float().sum(0, keepdim=True)) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scale_cast_resid_gate_add_act_bwd(dtype):
    from more4d_amd import ops
    B, L, C = 2, 50, 64
    x = torch.randn(B, L, C, generator=gen())
    e = torch.randn(B, 6, C, generator=gen(1))
    y = torch.randn(B, L, C, generator=gen(2)).to(dtype)
    ed = e.to(DEV)
    got = ops.scale_cast(x.to(DEV), dtype, gate=ed[:, 2], gate_stride=6 * C, rows_per_sample=L)
    assert torch.equal(got.cpu(), cpu_ops.scale_cast(x, dtype, gate=e[:, 2], gate_stride=6 * C, rows_per_sample=L))
    assert torch.equal(ops.scale_cast(x.to(DEV), dtype).cpu(), x.to(dtype))
    got = ops.resid_gate(x.to(DEV), y.to(DEV), gate=ed[:, 5], gate_stride=6 * C, rows_per_sample=L)
    assert rel_err(got.cpu(), cpu_ops.resid_gate(x, y, gate=e[:, 5], gate_stride=6 * C, rows_per_sample=L)) < 1e-6
    assert rel_err(ops.resid_gate(x.to(DEV), y.to(DEV)).cpu(), x + y.float()) < 1e-6
    z = torch.randn(B, L, C, generator=gen(3)).to(dtype)
    assert torch.equal(ops.add(y.to(DEV), z.to(DEV)).cpu(), (y.float() + z.float()).to(dtype))
    for act in (1, 2, 3):
        pre = (torch.randn(B * L, C, generator=gen(4)) * 2).to(dtype)
        dy = torch.randn(B * L, C, generator=gen(5)).to(dtype)
        ref = cpu_ops.act_bwd_(dy.clone(), pre, act)
        got = ops.act_bwd_(dy.to(DEV), pre.to(DEV), act)
        assert rel_err(got.float().cpu(), ref.float()) < (1e-5 if dtype == torch.float32 else 1e-2)
        assert rel_err(ops.unary(pre.to(DEV), torch.float32, act=act).cpu(),
                       cpu_ops.unary(pre, torch.float32, act=act)) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [128, 5120])
def test_ln_modulate_bwd(dtype, C):
    from more4d_amd import ops
    B, L = 2, 37
    x = torch.randn(B, L, C, generator=gen()) * 2 + 0.3
    dy = torch.randn(B, L, C, generator=gen(1)).to(dtype)
    e = torch.randn(B, 6, C, generator=gen(2)) * 0.3
    lw = 1 + 0.1 * torch.randn(C, generator=gen(3))
    for mode in ("mod", "affine", "plain"):
        kw = dict(B=B, rows_per_sample=L, eps=1e-6)
        dx0 = torch.randn(B, L, C, generator=gen(4))
        if mode == "mod":
            G, rs = B, 6 * C
            ref_red, got_red = torch.zeros(B, 6, C), torch.zeros(B, 6, C, device=DEV)
            cpu = dict(scale=e[:, 4], mod_stride=6 * C, dshift=ref_red[:, 3], dscale=ref_red[:, 4], red_stride=rs)
            ed = e.to(DEV)
            dev = dict(scale=ed[:, 4], mod_stride=6 * C, dshift=got_red[:, 3], dscale=got_red[:, 4], red_stride=rs)
        elif mode == "affine":
            ref_red, got_red = torch.zeros(2, C), torch.zeros(2, C, device=DEV)
            cpu = dict(ln_w=lw, dshift=ref_red[0], dscale=ref_red[1], red_stride=0)
            dev = dict(ln_w=lw.to(DEV), dshift=got_red[0], dscale=got_red[1], red_stride=0)
        else:
            ref_red, got_red = torch.zeros(1), torch.zeros(1)
            cpu, dev = {}, {}
        ref = cpu_ops.ln_modulate_bwd(x, dy, dx0.clone(), **kw, **cpu)
        got = ops.ln_modulate_bwd(x.to(DEV), dy.to(DEV), dx0.to(DEV), **kw, **dev)
        assert rel_err(got.cpu(), ref) < 1e-5, mode
        assert rel_err(got_red.cpu(), ref_red) < 1e-5 or mode == "plain", mode


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,hd", [(128, 32), (5120, 128)])
def test_rmsnorm_rope_bwd(dtype, C, hd):
    from more4d_amd import ops
    from more4d_amd.models.wan_transformer4d import build_rope_tables, rope_params
    B, L = 2, 24 + 3
    freqs = torch.cat([rope_params(1024, hd - 4 * (hd // 6)), rope_params(1024, 2 * (hd // 6)),
                       rope_params(1024, 2 * (hd // 6))], dim=1)
    cos, sin = build_rope_tables(freqs, (2, 3, 4), hd, "cpu")
    xq = (torch.randn(B * L, 3 * C, generator=gen()) * 1.5).to(dtype)      # q | k | v column slices, row stride 3C
    dq = torch.randn(B * L, 3 * C, generator=gen(1)).to(dtype)
    wq, wk = 1 + 0.1 * torch.randn(C, generator=gen(2)), 1 + 0.1 * torch.randn(C, generator=gen(3))
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for rope in (True, False):
        kw = dict(head_dim=hd, eps=1e-6)
        if rope:
            kw.update(rows_per_sample=L, rope_len=24, pos_offset=0)
        rd, rw0, rw1 = dq.clone(), torch.zeros(C), torch.zeros(C)
        cpu_ops.rmsnorm_rope_bwd_(rd[:, :C], xq[:, :C], wq, rw0, rd[:, C:2 * C], xq[:, C:2 * C], wk, rw1,
                                  cos=cos if rope else None, sin=sin if rope else None, **kw)
        gd, gw0, gw1 = dq.to(DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        xd = xq.to(DEV)
        ops.rmsnorm_rope_bwd_(gd[:, :C], xd[:, :C], wq.to(DEV), gw0, gd[:, C:2 * C], xd[:, C:2 * C], wk.to(DEV), gw1,
                              cos=cos.to(DEV) if rope else None, sin=sin.to(DEV) if rope else None, **kw)
        assert rel_err(gd[:, :2 * C].float().cpu(), rd[:, :2 * C].float()) < tol
        assert torch.equal(gd[:, 2 * C:].cpu(), dq[:, 2 * C:])            # v slice untouched
        assert rel_err(gw0.cpu(), rw0) < tol and rel_err(gw1.cpu(), rw1) < tol
        # single-tensor form (cross-attention q / k)
        gd1, gw = dq[:, :C].contiguous().to(DEV), torch.zeros(C, device=DEV)
        ops.rmsnorm_rope_bwd_(gd1, xq[:, :C].contiguous().to(DEV), wq.to(DEV), gw, head_dim=hd, eps=1e-6)
        if not rope:
            assert rel_err(gd1.float().cpu(), rd[:, :C].float()) < tol and rel_err(gw.cpu(), rw0) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hd,heads,Lq,Lk,Lk_rows", [(32, 4, 200, 200, 200), (64, 2, 136, 77, 80), (128, 2, 304, 257, 264),
                                                     (128, 3, 64, 1000, 1000), (128, 2, 2304, 257, 264)])
def test_attention_lse_and_bwd(dtype, hd, heads, Lq, Lk, Lk_rows):
    from more4d_amd import ops
    from more4d_amd.ops import KV
    B, C = 2, heads * hd
    q = torch.randn(B * Lq, C, generator=gen()).to(dtype)
    k = torch.randn(B * Lk_rows, C, generator=gen(1)).to(dtype)
    v = torch.randn(B * Lk_rows, C, generator=gen(2)).to(dtype)
    d_o = torch.randn(B * Lq, C, generator=gen(3)).to(dtype)
    qd, kd, vd, dod = q.to(DEV), k.to(DEV), v.to(DEV), d_o.to(DEV)
    vt = ops.transpose(vd)
    lse = torch.empty(B, heads, Lq, device=DEV)
    o = ops.attention(qd, [KV(kd, vt, Lk_rows * C, C, Lk_rows, B * Lk_rows, Lk)], B=B, Lq=Lq, heads=heads, head_dim=hd,
                      q_bs=Lq * C, q_ls=C, lse=lse).view(B * Lq, C)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float().view(B, Lq, heads, hd), k.float().view(B, Lk_rows, heads, hd)[:, :Lk])
    lse_ref = torch.logsumexp(s / math.sqrt(hd), -1) * math.log2(math.e)
    assert float((lse.cpu() - lse_ref).abs().max()) < (1e-4 if dtype == torch.float32 else 5e-2)
    dq0 = torch.randn(B * Lq, C, generator=gen(4)).to(dtype)
    rq, rk, rv = dq0.clone(), torch.empty_like(k), torch.empty_like(v)
    cpu_ops.attention_bwd(q, k, v, None, d_o, None, B=B, Lq=Lq, Lk=Lk, Lk_rows=Lk_rows, heads=heads, head_dim=hd,
                          dq=rq, dk=rk, dv=rv, accumulate_dq=True)
    gq = dq0.to(DEV)
    gk = torch.full_like(kd, float("nan"))
    gv = torch.full_like(vd, float("nan"))
    ops.attention_bwd(qd, kd, vd, o, dod, lse, B=B, Lq=Lq, Lk=Lk, Lk_rows=Lk_rows, heads=heads, head_dim=hd,
                      dq=gq, dk=gk, dv=gv, accumulate_dq=True)
    tol = TOL if dtype == torch.float32 else 3e-2
    assert rel_err(gq.float().cpu(), rq.float()) < tol
    assert rel_err(gk.float().cpu(), rk.float()) < tol
    assert rel_err(gv.float().cpu(), rv.float()) < tol
    pad = gk.view(B, Lk_rows, C)[:, Lk:]
    assert float(pad.float().abs().sum()) == 0 and float(gv.view(B, Lk_rows, C)[:, Lk:].float().abs().sum()) == 0


def test_attention_bwd_strided_slices_fp32():
    """q/k/v and dq/dk/dv as column slices of one [R, 3C] buffer (how the block backward calls it)."""
    from more4d_amd import ops
    from more4d_amd.ops import KV
    B, L, heads, hd = 1, 136, 2, 64
    C = heads * hd
    qkv = torch.randn(B * L, 3 * C, generator=gen())
    d_o = torch.randn(B * L, C, generator=gen(1))
    dev = qkv.to(DEV)
    q, k, v = dev[:, :C], dev[:, C:2 * C], dev[:, 2 * C:]
    lse = torch.empty(B, heads, L, device=DEV)
    o = ops.attention(q, [KV(k, ops.transpose(v), L * 3 * C, 3 * C, L, B * L, L - 5)], B=B, Lq=L, heads=heads, head_dim=hd,
                      q_bs=L * 3 * C, q_ls=3 * C, lse=lse).view(B * L, C)
    dqkv = torch.empty_like(dev)
    ops.attention_bwd(q, k, v, o, d_o.to(DEV), lse, B=B, Lq=L, Lk=L - 5, Lk_rows=L, heads=heads, head_dim=hd,
                      dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:])
    ref = torch.empty_like(qkv)
    cpu_ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], None, d_o, None, B=B, Lq=L, Lk=L - 5, Lk_rows=L,
                          heads=heads, head_dim=hd, dq=ref[:, :C], dk=ref[:, C:2 * C], dv=ref[:, 2 * C:])
    assert rel_err(dqkv.cpu(), ref) < TOL


@pytest.mark.parametrize("dtype,sdtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                          (torch.bfloat16, torch.float32)])
def test_adamw_and_sumsq(dtype, sdtype):
    from more4d_amd import ops
    n = 10007
    p0 = torch.randn(n, generator=gen()).to(dtype)
    hp = dict(lr=2e-3, betas=(0.9, 0.999), eps=1e-10, weight_decay=3e-2)     # train_wan.sh: wd 3e-2, eps 1e-10
    ref_p = torch.nn.Parameter(p0.float().clone())
    opt = torch.optim.AdamW([ref_p], **hp)
    p = p0.to(DEV)
    m, v = torch.zeros(n, device=DEV, dtype=sdtype), torch.zeros(n, device=DEV, dtype=sdtype)
    scale = torch.tensor(0.5, device=DEV)
    for step in range(1, 4):
        g = torch.randn(n, generator=gen(step)).to(dtype)
        ref_p.grad = g.float() * 0.5
        opt.step()
        ops.adamw_(p, g.to(DEV), m, v, lr=hp["lr"], beta1=0.9, beta2=0.999, eps=hp["eps"], weight_decay=hp["weight_decay"],
                   step=step, grad_scale=scale)
        ss = torch.zeros((), device=DEV)
        ops.sumsq(g.to(DEV), ss)
        assert abs(float(ss) - float(g.float().pow(2).sum())) < 1e-4 * float(ss)
    tol = 1e-6 if dtype == torch.float32 else (3e-2 if sdtype == torch.bfloat16 else 1e-2)
    assert rel_err(p.float().cpu(), ref_p.detach()) < tol

@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,P,L,glen", [(128, 12, 45, 36), (5120, 7, 24, 17), (2048, 5, 16, 16)])
def test_guidance_bwd(dtype, C, P, L, glen):
    """m4d_guidance_bwd: dz -> du in place and the per-position (sum dz*u | sum dz) table; ragged last period, rows past
    g_len untouched, positions without rows zero."""
    from more4d_amd import ops
    B = 2
    x = torch.randn(B, L, C, generator=gen()) * 2 + 0.3
    dz = torch.randn(B, L, C, generator=gen(1)).to(dtype)
    e = torch.randn(B, 6, C, generator=gen(2)) * 0.3
    ss = torch.randn(B, P, 2 * C, generator=gen(3)) * 0.5
    gate = torch.randn(C, generator=gen(4))
    kw = dict(B=B, rows_per_sample=L, mod_stride=6 * C, g_period=P, g_len=glen, eps=1e-6)
    ref_dz = dz.clone()
    ref_ab = cpu_ops.guidance_bwd_(x, ref_dz, shift=e[:, 3], scale=e[:, 4], g_ss=ss, g_gate=gate, **kw)
    ed, got_dz = e.to(DEV), dz.to(DEV)
    got_ab = ops.guidance_bwd_(x.to(DEV), got_dz, shift=ed[:, 3], scale=ed[:, 4], g_ss=ss.to(DEV), g_gate=gate.to(DEV), **kw)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(got_ab.cpu(), ref_ab) < 1e-5
    assert rel_err(got_dz.float().cpu(), ref_dz.float()) < tol
    assert torch.equal(got_dz[:, glen:].cpu(), dz[:, glen:])



def tiny_model(dtype):
    from more4d_amd.models import WanTransformer4DModel
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
    return m.to(DEV, dtype).train()


def test_tiny_dit_gradients_fp32():
    """loss.backward() through the HIP kernels == the reference's gradients for EVERY parameter (1e-3)."""
    z, zg = load_npz("dit_tiny.npz"), load_npz("dit_tiny_grads.npz")
    m = tiny_model(torch.float32)
    pred = m(x=z["x"].to(DEV), t=z["t"].to(DEV), context=[z["ctx0"].to(DEV), z["ctx1"].to(DEV)],
             seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV), full_ref=z["full_ref"].to(DEV))
    assert rel_err(pred.detach().cpu(), zg["pred"]) < TOL
    loss = custom_mse_loss(pred, zg["target"].to(DEV))
    assert abs(float(loss.detach()) - float(zg["loss"])) < 1e-4 * float(zg["loss"])
    loss.backward()
    worst = check_grads({n: p.grad for n, p in m.named_parameters()}, zg, TOL)
    print("worst gradient error", worst)
    assert m.last_stored_blocks == len(m.blocks)           # default policy on a 288 GB part: no recompute
    # plain per-block gradient checkpointing (budget 0) gives the same gradients
    ref_grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    m.activation_budget_gb = 0
    pred2 = m(x=z["x"].to(DEV), t=z["t"].to(DEV), context=[z["ctx0"].to(DEV), z["ctx1"].to(DEV)],
              seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV), full_ref=z["full_ref"].to(DEV))
    custom_mse_loss(pred2, zg["target"].to(DEV)).backward()
    assert m.last_stored_blocks == 0
    same_grads({n: p.grad for n, p in m.named_parameters()}, ref_grads)


def test_tiny_dit_gradients_bf16_budget():
    """Production dtype: bf16 parameters / activations / gradients against the fp32 reference gradients."""
    z, zg = load_npz("dit_tiny.npz"), load_npz("dit_tiny_grads.npz")
    m = tiny_model(torch.bfloat16)
    bf = torch.bfloat16
    pred = m(x=z["x"].to(DEV, bf), t=z["t"].to(DEV), context=[z["ctx0"].to(DEV), z["ctx1"].to(DEV)],
             seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV, bf),
             full_ref=z["full_ref"].to(DEV, bf))
    custom_mse_loss(pred, zg["target"].to(DEV)).backward()
    grads = {n: p.grad for n, p in m.named_parameters()}
    assert all(g is None or g.dtype == bf for g in grads.values())
    check_grads(grads, zg, 0.3, norm_tol=0.08)


def test_train_step_reduces_loss():
    """Three fused clip+AdamW steps on the tiny model move the loss down (optimizer wiring, grad-norm, clip)."""
    from more4d_amd.optim import AdamW, clip_grad_norm_
    z, zg = load_npz("dit_tiny.npz"), load_npz("dit_tiny_grads.npz")
    m = tiny_model(torch.float32)
    opt = AdamW(m.parameters(), lr=1e-3, weight_decay=3e-2, eps=1e-10)
    args = dict(x=z["x"].to(DEV), t=z["t"].to(DEV), context=[z["ctx0"].to(DEV), z["ctx1"].to(DEV)],
                seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV), full_ref=z["full_ref"].to(DEV))
    losses = []
    for _ in range(3):
        loss = custom_mse_loss(m(**args), zg["target"].to(DEV))
        loss.backward()
        ref_norm = torch.norm(torch.stack([p.grad.float().norm() for p in m.parameters() if p.grad is not None]))
        total = clip_grad_norm_(m.parameters(), 0.05, optimizer=opt)
        assert abs(float(total) - float(ref_norm)) < 1e-4 * float(ref_norm)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    assert losses[2] < losses[1] < losses[0]


def test_guided_dit_gradients():
    """Guided training (train_wan.sh --use_omnimae_guidance): gate / spatial_guide / feature_adapter gradients of the HIP
    path against the reference's (tests/golden/dit_tiny_guid_grads.npz), fp32 at 1e-3 and bf16 within the bf16 budget."""
    from more4d_amd.models import WanTransformer4DModel
    z, zg = load_npz("dit_tiny.npz"), load_npz("dit_tiny_guid_grads.npz")
    for dtype in (torch.float32, torch.bfloat16):
        m = WanTransformer4DModel(**dict(TINY, use_omnimae_guidance=True))
        missing = m.load_state_dict(fill(load_keys("dit_tiny_guid_keys.json"), 4321), strict=False)
        assert all(k.startswith("omnimae_extractor.") for k in missing.missing_keys)
        m = m.to(DEV, dtype).train()
        kw = dict(x=z["x"].to(DEV, dtype), t=z["t"].to(DEV), context=[z["ctx0"].to(DEV), z["ctx1"].to(DEV)],
                  seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV, dtype),
                  full_ref=z["full_ref"].to(DEV, dtype), first_frame_features=(zg["patch"].to(DEV), zg["cls"].to(DEV)))
        for budget in (None, 0):
            m.zero_grad(set_to_none=True)
            m.activation_budget_gb = budget
            pred = m(**kw)
            custom_mse_loss(pred, zg["target"].to(DEV)).backward()
            grads = {n: p.grad for n, p in m.named_parameters()}
            assert grads["feature_adapter.0.weight"] is not None and grads["blocks.1.spatial_guidance_ffn.gate"] is not None
            if dtype == torch.float32:
                assert rel_err(pred.detach().cpu(), zg["pred"]) < TOL
                print("worst guided gradient error", check_grads(grads, zg, TOL))
            else:
                check_grads(grads, zg, 0.3, norm_tol=0.08)
        with torch.no_grad():           # inference with the same features: same prediction as the training forward
            out = m(**kw)
        assert rel_err(out.float().cpu(), pred.detach().float().cpu()) < (1e-5 if dtype == torch.float32 else 2e-2)


def test_sharded_data_parallel_flat_buckets_on_device():
    """more4d_amd.dist.data_parallel on the MI355X at world size 1 (no collective: the same code path minus RCCL): parameters and
    gradients live in flat buckets, the fused clip + AdamW kernel updates the bucket slices — two steps must equal two steps of the
    per-parameter optimizer on an identical model (same kernels, different memory layout)."""
    from more4d_amd.dist.data_parallel import ShardedDataParallel
    from more4d_amd.models import WanTransformer4DModel
    from more4d_amd.optim import AdamW, clip_grad_norm_
    from weights import fill
    from util import load_keys, load_npz
    TINY_ = dict(model_type="i2v", in_dim=64, dim=128, ffn_dim=512, num_heads=4, num_layers=2, text_dim=64, text_len=32,
                 freq_dim=256, out_dim=16, add_ref_conv=True, use_dino_guidance=False, cross_attn_norm=True)
    z = load_npz("dit_tiny.npz")
    sd0 = fill(load_keys("dit_tiny_keys.json"), 1234)
    kw = dict(x=z["x"].to("cuda"), t=z["t"].to("cuda"), context=[z["ctx0"].to("cuda"), z["ctx1"].to("cuda")], seq_len=int(z["seq_len_pad"]),
              clip_fea=z["clip"].to("cuda"), y=z["y"].to("cuda"), full_ref=z["full_ref"].to("cuda"))
    tgt = torch.randn(z["x"].shape, generator=torch.Generator().manual_seed(0)).to("cuda")
    hp = dict(lr=1e-3, weight_decay=3e-2, eps=1e-10)

    def make():
        m = WanTransformer4DModel(**TINY_)
        m.load_state_dict(sd0)
        return m.to("cuda").train()
    a, b = make(), make()
    dp = ShardedDataParallel(a, bucket_bytes=400_000, **hp)
    assert len(dp.buckets) > 2 and all(torch.equal(p.detach().cpu(), sd0[n]) for n, p in a.named_parameters())
    opt = AdamW(b.parameters(), **hp)
    for _ in range(2):
        ((a(**kw).float() - tgt) ** 2).mean().backward()
        total = dp.reduce_gradients()
        dp.step(max_norm=0.05, total_norm=total)
        dp.zero_grad()
        ((b(**kw).float() - tgt) ** 2).mean().backward()
        tb = clip_grad_norm_(b.parameters(), 0.05, optimizer=opt)
        opt.step()
        opt.zero_grad()
        assert abs(float(total) - float(tb)) < 1e-5 * float(tb)
    # same kernels, different reduction order of the gradient norm: the clip coefficient differs in the last bits, and AdamW's
    # g / (sqrt(v) + 1e-10) turns that into up to a few 1e-6 on elements whose gradient is ~0 (two steps of lr = 1e-3); the mean bound
    # covers small tensors (a 64-element bias with one such element: 1.5e-6 / 64 plus the atomics' order in its column sums)
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        d = (pa.detach() - pb.detach()).abs()
        assert float(d.max()) < 2e-5 and float(d.mean()) < 5e-7, (n, float(d.max()), float(d.mean()))
