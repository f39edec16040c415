"""Round-4 fixtures, all produced by RUNNING THE REFERENCE in the build container (needs /root/reference):

    python tests/golden/make_golden_r4.py [calib|stack4|xattn|pertoken|noqknorm|all]

* bf16_calibration.json — how far the REFERENCE'S OWN bf16 run (torch.autocast("cpu", bfloat16) around the same call,
  the reference's inner `torch.cuda.amp.autocast(dtype=float32)` islands honoured) sits from its fp32 run, in exactly the
  metrics the GPU tests use.  The bf16 budgets of the GPU tests are 1.5 x these numbers instead of hand-picked constants.
* dit_stack4_14b_long.npz — FOUR stacked 14B-width blocks at L = 2080 (error growth over stacked layers through the
  production bf16 kernels), fp32 reference rows / norms.
* dit_block_xattn.npz — WanAttentionBlock with `t2v_cross_attn` and `cross_attn` (wan_transformer4d.py:468-497, 558-575).
* dit_tiny_pertoken.npz — the tiny DiT with a PER-TOKEN timestep t [B, seq_len] (:655-657, :713-715, :1161-1167).
* dit_block_noqknorm.npz — a block built with qk_norm=False (:431-432: norm_q / norm_k = nn.Identity).
Data only; no reference source is stored."""
import contextlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
from make_golden import TINY_DIT, _probe_samples, load_recipe, npz_save, randomize, sd_arrays  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rms_rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


class _AmpShim:
    """The reference writes `torch.cuda.amp.autocast(dtype=float32)` / `(enabled=False)` around the pieces it wants in fp32
    (sinusoid + time MLP :1160, rope :252-340).  On a CPU-only torch those CUDA contexts do not touch the CPU autocast state, so
    a CPU bf16-autocast run would lose the fp32 islands the GPU run has.  Map them to the CPU equivalent: autocast off."""

    @staticmethod
    def autocast(*a, **k):
        return torch.autocast("cpu", enabled=False)


@contextlib.contextmanager
def ref_bf16(ref):
    old = ref.dit.amp
    ref.dit.amp = _AmpShim
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            yield
    finally:
        ref.dit.amp = old


def _block14(ref, seed):
    from weights import block_weights_14b
    d = ref.dit
    blk = d.WanAttentionBlock("i2v_cross_attn", 5120, 13824, 40, (-1, -1), True, True, 1e-6, use_spatial_guidance=False).eval()
    sd = block_weights_14b(seed=seed)
    blk.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items()})
    return blk


def _long_inputs(ref):
    from weights import randn_named
    d = ref.dit
    L = 2080
    x = randn_named("in.x", (1, L, 5120), 6)
    e0 = randn_named("in.e0", (1, 6, 5120), 6, 0.2)
    ctx = randn_named("in.ctx", (1, 257 + 512, 5120), 6)
    freqs = torch.cat([d.rope_params(1024, 128 - 4 * (128 // 6)), d.rope_params(1024, 2 * (128 // 6)),
                       d.rope_params(1024, 2 * (128 // 6))], dim=1)
    return L, (4, 20, 26), x, e0, ctx, freqs


def _block_metrics(y, yref, x):
    """the metrics of tests/test_round2_gpu.py::test_block_14b_width_long_sequence_vs_reference[bf16]"""
    rows = torch.cat([torch.arange(0, y.shape[1], 32), torch.tensor([y.shape[1] - 1])])
    d, dr = (y - x)[0].float(), (yref - x)[0]
    return dict(delta_max=float((d[rows] - dr[rows]).abs().max() / dr[rows].abs().max()),
                delta_rms=rms_rel_err(d[rows], dr[rows]), delta_norm=rel_err(d.norm(dim=-1), dr.norm(dim=-1)),
                out_rms=rms_rel_err(y[0, rows].float(), yref[0, rows]))


def calib_block_and_stack(ref, out):
    L, grid, x, e0, ctx, freqs = _long_inputs(ref)
    args = (torch.tensor([L]), torch.tensor([list(grid)]), freqs, ctx, None)
    ys32, ys16 = x, x
    rows = torch.cat([torch.arange(0, L, 32), torch.tensor([L - 1])])
    for layer in range(4):
        blk = _block14(ref, layer)
        ys32 = blk(ys32, e0, *args, dtype=torch.float32, t=0, dino_features=None)
        with ref_bf16(ref):
            ys16 = blk(ys16, e0, *args, dtype=torch.bfloat16, t=0, dino_features=None).float()
        if layer == 0:
            out["block_14b_long"] = _block_metrics(ys16, ys32, x)
            print("block_14b_long", out["block_14b_long"])
        del blk
    out["stack4_14b_long"] = _block_metrics(ys16, ys32, x)
    print("stack4_14b_long", out["stack4_14b_long"])
    npz_save("dit_stack4_14b_long.npz", grid=np.array(grid), rows=rows, out_rows=ys32[0, rows], row_norm=ys32[0].norm(dim=-1),
             delta_rows=(ys32 - x)[0, rows], delta_norm=(ys32 - x)[0].norm(dim=-1))


def calib_vae(ref, out):
    """the stages of make_golden.make_vae_probe under bf16 autocast, fed the SAME stored (fp16-rounded) stage inputs"""
    from weights import fill
    v, t = ref.vae, ref.traj
    vae = v.AutoencoderKLWan().eval()
    vae.load_state_dict(fill(json.load(open(os.path.join(HERE, "vae_keys.json"))), seed=2024))
    ea, da = t.VAEEncoderadaptor().eval(), t.VAEDecoderadaptor().eval()
    ea.load_state_dict(fill(json.load(open(os.path.join(HERE, "adaptor_enc_keys.json"))), seed=31))
    da.load_state_dict(fill(json.load(open(os.path.join(HERE, "adaptor_dec_keys.json"))), seed=32))
    for size in ("120x208", "96x128"):
        z = np.load(os.path.join(HERE, f"vae_probe_{size}.npz"))
        T, H, W = (int(a) for a in z["shape"])
        traj = torch.rand(1, 3, T, H, W, generator=torch.Generator().manual_seed(int(z["seed"])))

        def chk(got, s, n):
            g = got.float()
            return dict(max=rel_err(g.reshape(-1)[::11], torch.from_numpy(s)), rms=rms_rel_err(g.reshape(-1)[::11], torch.from_numpy(s)),
                        nrm=rel_err(g[0].flatten(2).norm(dim=-1), torch.from_numpy(n)))
        res = {}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            pseudo = ea(traj) * 2 - 1
            res["enc-adaptor"] = chk(pseudo, z["pseudo_s"], z["pseudo_n"])
            enc = vae._encode(torch.from_numpy(z["pv16"]).float())
            res["encode"] = dict(max=rel_err(enc.float(), torch.from_numpy(z["enc"])), rms=rms_rel_err(enc.float(), torch.from_numpy(z["enc"])))
            dec = vae._decode(torch.from_numpy(z["enc"][:, :16]).half().float()).sample
            res["decode"] = chk(dec, z["dec_s"], z["dec_n"])
            rec = da(torch.from_numpy(z["dec16"]).float())
            res["dec-adaptor"] = chk(rec, z["rec_s"], z["rec_n"])
            chain = da(vae._decode(vae._encode(pseudo.float())[:, :16].float()).sample.float())
            res["chain"] = chk(chain, z["rec_s"], z["rec_n"])
        out[f"vae_probe_{size}"] = res
        print(size, json.dumps(res))


def calib_tiny(ref, out):
    """tiny DiT forward and the 50-step loop (make_golden.make_dit_tiny / make_loop) under bf16 autocast"""
    z = np.load(os.path.join(HERE, "dit_tiny.npz"))
    m = ref.dit.WanTransformer4DModel(**TINY_DIT).eval()
    load_recipe(m, None, seed=1234)
    tt = {k: torch.from_numpy(z[k]) for k in z.files}
    with ref_bf16(ref):
        o = m(x=tt["x"], t=tt["t"], context=[tt["ctx0"], tt["ctx1"]], seq_len=int(tt["seq_len_pad"]), clip_fea=tt["clip"], y=tt["y"],
              full_ref=tt["full_ref"])
    out["dit_tiny"] = dict(max=rel_err(o.float(), tt["out_ref"]), rms=rms_rel_err(o.float(), tt["out_ref"]))
    print("dit_tiny", out["dit_tiny"])
    lz = np.load(os.path.join(HERE, "loop_tiny.npz"))
    lt = {k: torch.from_numpy(lz[k]) for k in lz.files}
    m = ref.dit.WanTransformer4DModel(**TINY_DIT).eval()
    load_recipe(m, None, seed=1234)
    sch = ref.fm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    sch.set_timesteps(sigmas=ref.fm.get_sampling_sigmas(int(lt["steps"]), float(lt["shift"])))
    # the pipeline draws the latents in weight_dtype (prepare_latents, pipeline_wan_fun_control.py:595-606) — with bf16 latents the
    # scheduler's x0 prediction and its final cast (fm_solvers.py:415-483, :789) run / land in bf16 as well
    x = lt["lat"].clone().to(torch.bfloat16)
    gs = float(lt["guidance"])
    for t in sch.timesteps:
        with ref_bf16(ref):
            v = m(x=torch.cat([x, x]), t=t.expand(2), context=[lt["ctx_u"], lt["ctx_c"]], seq_len=256, clip_fea=torch.cat([lt["clip"]] * 2),
                  y=torch.cat([lt["y"]] * 2), full_ref=torch.cat([lt["full_ref"]] * 2))
        # the reference pipeline combines the two branches IN THE MODEL'S OUTPUT DTYPE (bf16 under autocast, every operation rounded,
        # pipeline_wan_fun_control.py:822-825) and hands that to scheduler.step, which keeps the latents in fp32 (:828)
        assert v.dtype == torch.bfloat16
        vu, vc = v.chunk(2)
        x = sch.step(vu + gs * (vc - vu), t, x, return_dict=False)[0]
        assert x.dtype == torch.bfloat16
    x = x.float()
    out["loop_tiny"] = dict(max=rel_err(x, lt["final"]), rms=rms_rel_err(x, lt["final"]))
    print("loop_tiny", out["loop_tiny"])


def calib_block_grads(ref, out):
    """Gradients of one 14B-width block at L = 2080 for a fixed seeded cotangent (loss = sum(y * r)): the reference's fp32
    gradients (sampled values + norms -> dit_block_14b_long_grads.npz) and how far its bf16-autocast gradients sit from them."""
    from weights import randn_named
    L, grid, x, e0, ctx, freqs = _long_inputs(ref)
    r = randn_named("cot.y", (1, L, 5120), 6)
    args = (torch.tensor([L]), torch.tensor([list(grid)]), freqs, ctx, None)
    res = {}
    for mode in ("fp32", "bf16"):
        blk = _block14(ref, 0)
        xg = x.clone().requires_grad_(True)
        with torch.enable_grad():
            if mode == "fp32":
                y = blk(xg, e0, *args, dtype=torch.float32, t=0, dino_features=None)
            else:
                with ref_bf16(ref):
                    y = blk(xg, e0, *args, dtype=torch.bfloat16, t=0, dino_features=None)
            (y.float() * r).sum().backward()
        g = {"x": xg.grad.detach()}
        g.update({n: p.grad.detach().float() for n, p in blk.named_parameters()})
        res[mode] = g
        del blk
    from make_golden import grad_sample
    arrs, cal = {}, {}
    for n, g32 in res["fp32"].items():
        arrs["grad/" + n] = grad_sample(g32)
        arrs["norm/" + n] = g32.norm()
        cal[n] = dict(rms=rms_rel_err(res["bf16"][n], g32), max=rel_err(res["bf16"][n], g32))
    out["block_14b_long_grads"] = cal
    print("block_14b_long_grads", json.dumps(cal))
    npz_save("dit_block_14b_long_grads.npz", **arrs)


def make_calibration(ref, only=None):
    path = os.path.join(HERE, "bf16_calibration.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out["_doc"] = ("err(reference under torch.autocast('cpu', bfloat16), reference fp32) in the metrics of the GPU tests; "
                   "made by tests/golden/make_golden_r4.py calib")
    if only in (None, "tiny"):
        calib_tiny(ref, out)
    if only in (None, "vae"):
        calib_vae(ref, out)
    if only in (None, "block"):
        calib_block_and_stack(ref, out)
    if only in (None, "grads"):
        calib_block_grads(ref, out)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", path)


SMALL_BLOCK = dict(dim=128, ffn_dim=512, num_heads=4)


def _small_block_io(ref, blk, n_ctx, seed):
    d = ref.dit
    g = torch.Generator().manual_seed(seed)
    grid = (2, 6, 8)
    L = 96
    x = torch.randn(1, L, 128, generator=g)
    e0 = torch.randn(1, 6, 128, generator=g) * 0.2
    ctx = torch.randn(1, n_ctx, 128, generator=g)
    hd = 32
    freqs = torch.cat([d.rope_params(1024, hd - 4 * (hd // 6)), d.rope_params(1024, 2 * (hd // 6)), d.rope_params(1024, 2 * (hd // 6))], dim=1)
    y = blk(x, e0, torch.tensor([L]), torch.tensor([list(grid)]), freqs, ctx, None, dtype=torch.float32, t=0, dino_features=None)
    return dict(x=x, e0=e0, ctx=ctx, grid=np.array(grid), out=y)


def make_xattn_variants(ref):
    """t2v_cross_attn / cross_attn blocks (no image branch: the whole context is text) with cross_attn_norm on and off."""
    arrs = {}
    for name, norm3 in (("t2v_cross_attn", True), ("cross_attn", False)):
        torch.manual_seed(11)
        blk = ref.dit.WanAttentionBlock(name, 128, 512, 4, (-1, -1), True, norm3, 1e-6, use_spatial_guidance=False).eval()
        for p in blk.parameters():
            p.data.normal_(0, 0.08)
        for k, v in _small_block_io(ref, blk, 40, 5).items():
            arrs[f"{name}/{k}"] = v
        arrs.update(sd_arrays(f"{name}/w/", blk))
    npz_save("dit_block_xattn.npz", **arrs)


def make_noqknorm(ref):
    torch.manual_seed(12)
    blk = ref.dit.WanAttentionBlock("i2v_cross_attn", 128, 512, 4, (-1, -1), False, True, 1e-6, use_spatial_guidance=False).eval()
    for p in blk.parameters():
        p.data.normal_(0, 0.08)
    arrs = _small_block_io(ref, blk, 257 + 24, 6)
    arrs.update(sd_arrays("w/", blk))
    npz_save("dit_block_noqknorm.npz", **arrs)


def make_pertoken(ref):
    """Per-token timesteps: t [B, seq_len] -> e [B, L, C], e0 [B, L, 6, C]; the blocks' and the head's `e.dim() > 3 / 2` branches."""
    z = np.load(os.path.join(HERE, "dit_tiny.npz"))
    m = ref.dit.WanTransformer4DModel(**TINY_DIT).eval()
    load_recipe(m, None, seed=1234)
    tt = {k: torch.from_numpy(z[k]) for k in z.files}
    seq_len = int(tt["seq_len_pad"])
    g = torch.Generator().manual_seed(99)
    n_ref = (tt["full_ref"].shape[-2] // 2) * (tt["full_ref"].shape[-1] // 2)      # the reference adds the ref row to seq_len (:1088) BEFORE it unflattens t
    t_tok = torch.rand(tt["x"].shape[0], seq_len + n_ref, generator=g) * 1000.0
    o = m(x=tt["x"], t=t_tok, context=[tt["ctx0"], tt["ctx1"]], seq_len=seq_len, clip_fea=tt["clip"], y=tt["y"], full_ref=tt["full_ref"])
    seq_len2 = int(tt["seq_len"])
    t_tok2 = torch.rand(tt["x"].shape[0], seq_len2, generator=g) * 1000.0
    o2 = m(x=tt["x"], t=t_tok2, context=[tt["ctx0"], tt["ctx1"]], seq_len=seq_len2, clip_fea=tt["clip"], y=tt["y"], full_ref=None)
    npz_save("dit_tiny_pertoken.npz", t_tok=t_tok, out_ref=o, t_tok_noref=t_tok2, out_noref=o2)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ref = _ref_import.load_reference()
    if what in ("xattn", "all"):
        make_xattn_variants(ref)
    if what in ("noqknorm", "all"):
        make_noqknorm(ref)
    if what in ("pertoken", "all"):
        make_pertoken(ref)
    if what in ("calib", "all"):
        make_calibration(ref, sys.argv[2] if len(sys.argv) > 2 else None)
