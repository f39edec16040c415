"""Round-4 CPU tests (no GPU): the HOST logic of the branches filled in this round — per-token timesteps, qk_norm=False, the
t2v / plain cross-attention blocks, the pipeline's nearest resize, TeaCache / cfg-skip under CFG-parallel ranks — with the
kernels replaced by tests/cpu_ops.py (test-only stand-ins), against fixtures produced by the reference
(tests/golden/make_golden_r4.py), and the calibration file the bf16 budgets of the GPU tests come from."""
import json
import os

import pytest
import torch

import cpu_ops
from util import GOLDEN, bf16_budget, load_keys, load_npz, rel_err
from weights import fill

TINY = dict(model_type="i2v", in_dim=64, dim=128, ffn_dim=512, num_heads=4, num_layers=2, text_dim=64, text_len=32,
            freq_dim=256, out_dim=16, add_ref_conv=True, use_dino_guidance=False, cross_attn_norm=True)


def _freqs(d):
    from more4d_amd.models.wan_transformer4d import rope_params
    return torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)), rope_params(1024, 2 * (d // 6))], dim=1)


def test_bf16_calibration_file_is_complete():
    """every budget the GPU tests look up exists, is positive, and is the size a bf16 pipeline can have (1e-4 .. 0.25)"""
    with open(os.path.join(GOLDEN, "bf16_calibration.json")) as fh:
        c = json.load(fh)
    for size in ("120x208", "96x128"):
        for stage in ("enc-adaptor", "encode", "decode", "dec-adaptor", "chain"):
            for m in ("max", "rms") + (("nrm",) if stage != "encode" else ()):
                assert 1e-4 < c[f"vae_probe_{size}"][stage][m] < 0.25
    for key in ("block_14b_long", "stack4_14b_long"):
        for m in ("delta_max", "delta_rms", "delta_norm", "out_rms"):
            assert 1e-4 < c[key][m] < 0.05
    assert 1e-4 < c["dit_tiny"]["rms"] < 0.05 and 1e-4 < c["loop_tiny"]["rms"] < 0.05
    g = c["block_14b_long_grads"]
    assert len(g) == 33 and all(1e-4 < v["rms"] < 0.25 for v in g.values())
    assert bf16_budget("block_14b_long", "delta_rms") == 1.5 * c["block_14b_long"]["delta_rms"]


def test_per_token_timesteps_host_logic(monkeypatch):
    """t [B, seq_len]: e0 [B, L, 6, C] / e [B, L, C] tables, one modulation row per token (rows_per_sample = 1 in every kernel call),
    against the reference's output (wan_transformer4d.py:655-657, 713-715, 1161-1167)."""
    cpu_ops.install(monkeypatch)
    from more4d_amd.models import WanTransformer4DModel
    z, p = load_npz("dit_tiny.npz"), load_npz("dit_tiny_pertoken.npz")
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234), strict=True)
    m.eval()
    ctx = [z["ctx0"], z["ctx1"]]
    with torch.no_grad():
        out = m(x=z["x"], t=p["t_tok"], context=ctx, seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"], y=z["y"], full_ref=z["full_ref"])
        assert rel_err(out, p["out_ref"]) < 1e-4
        out = m(x=z["x"], t=p["t_tok_noref"], context=ctx, seq_len=int(z["seq_len"]), clip_fea=z["clip"], y=z["y"], full_ref=None)
        assert rel_err(out, p["out_noref"]) < 1e-4
        with pytest.raises(ValueError):
            m(x=z["x"], t=p["t_tok_noref"][:, :-3], context=ctx, seq_len=int(z["seq_len"]), clip_fea=z["clip"], y=z["y"])


def _small_block(name, norm3, qk_norm, z, prefix):
    from more4d_amd.models import WanAttentionBlock
    blk = WanAttentionBlock(name, 128, 512, 4, (-1, -1), qk_norm, norm3, 1e-6, use_spatial_guidance=False)
    blk.load_state_dict({k[len(prefix):]: v for k, v in z.items() if k.startswith(prefix)}, strict=True)
    return blk.eval()


@pytest.mark.parametrize("name,norm3", [("t2v_cross_attn", True), ("cross_attn", False)])
def test_t2v_and_plain_cross_attention_blocks_host_logic(monkeypatch, name, norm3):
    cpu_ops.install(monkeypatch)
    z = load_npz("dit_block_xattn.npz")
    blk = _small_block(name, norm3, True, z, f"{name}/w/")
    x = z[f"{name}/x"]
    with torch.no_grad():
        out = blk(x, z[f"{name}/e0"], torch.tensor([x.shape[1]]), z[f"{name}/grid"].view(1, 3), _freqs(32), z[f"{name}/ctx"], None,
                  dtype=torch.float32, t=0)
    assert rel_err(out, z[f"{name}/out"]) < 1e-4


def test_block_without_qk_norm_host_logic(monkeypatch):
    cpu_ops.install(monkeypatch)
    z = load_npz("dit_block_noqknorm.npz")
    blk = _small_block("i2v_cross_attn", True, False, z, "w/")
    with torch.no_grad():
        out = blk(z["x"], z["e0"], torch.tensor([z["x"].shape[1]]), z["grid"].view(1, 3), _freqs(32), z["ctx"], None,
                  dtype=torch.float32, t=0)
    assert rel_err(out, z["out"]) < 1e-4


def test_pipeline_preprocess_resizes_like_the_published_image_processor():
    """VaeImageProcessor.preprocess for tensors (diffusers, third-party: restated): target rounded down to a multiple of 8,
    F.interpolate's default (legacy nearest), [0, 1] -> [-1, 1] unless the tensor already holds negative values."""
    import torch.nn.functional as F
    from types import SimpleNamespace
    from more4d_amd.pipeline.pipeline_wan_fun_control import WanFunControlPipeline as P
    me = SimpleNamespace(vae=SimpleNamespace(spatial_compression_ratio=8))      # the factor comes from the VAE (reference :185-186)
    for hin, win, h, w in ((384, 512, 480, 832), (720, 960, 480, 832), (50, 70, 36, 52), (33, 47, 64, 96)):
        v = torch.rand(1, 3, 2, hin, win)
        want = F.interpolate(v[0].transpose(0, 1), size=(h - h % 8, w - w % 8)).transpose(0, 1)[None] * 2 - 1
        assert torch.equal(P._preprocess(me, v, h, w), want)
    v = torch.rand(1, 3, 1, 16, 16) * 2 - 1
    assert torch.equal(P._preprocess(me, v, 16, 16), v)
    me16 = SimpleNamespace(vae=SimpleNamespace(spatial_compression_ratio=16))
    assert P._preprocess(me16, torch.rand(1, 3, 1, 40, 40), 40, 40).shape[-2:] == (32, 32)


# ------------------------------------------------------------------ TeaCache + cfg-skip under CFG-parallel ranks (gloo)
def _cfgp_cache_worker(rank, world, port, q):
    import torch.distributed as dist
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import more4d_amd.ops as real
        for n in cpu_ops.NAMES:
            setattr(real, n, getattr(cpu_ops, n))
        from more4d_amd.dist import init_sequence_parallel
        from more4d_amd.models import WanTransformer4DModel
        from more4d_amd.models.cache_utils import get_teacache_coefficients
        from more4d_amd.pipeline import denoise_latents
        from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
        z = load_npz("teacache_loop.npz")
        steps = int(z["steps"])
        m = WanTransformer4DModel(**TINY)
        m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
        m.eval()
        calc = []

        def loop():
            m.enable_teacache(get_teacache_coefficients("Wan2.1-Fun-14B-Control"), steps, float(z["a_thresh"]), num_skip_start_steps=1,
                              offload=False)
            m.enable_cfg_skip(0.3, steps)
            sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
            ts, _ = retrieve_timesteps(sch, sigmas=get_sampling_sigmas(steps, float(z["shift"])))
            calc.clear()
            with torch.no_grad():
                out = denoise_latents(m, sch, z["lat"], ts, float(z["guidance"]), [z["ctx_u"], z["ctx_c"]], clip_fea=z["clip"], y=z["y"],
                                      full_ref=z["full_ref"], seq_len=256, callback=lambda i, t, x: calc.append(bool(m.should_calc)))
            assert m.cfg_skip_ratio == 0.3                  # restored after the loop
            m.disable_teacache()
            m.disable_cfg_skip()
            return out, list(calc)
        single, calc_single = loop()
        assert not all(calc_single) and any(calc_single[1:])           # the threshold gives a mix of computed and skipped steps
        init_sequence_parallel(cfg_parallel=True)
        m.enable_multi_gpus_inference()
        multi, calc_multi = loop()
        q.put((rank, float(rel_err(multi, single)), calc_multi == calc_single or rank < world // 2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_teacache_and_cfg_skip_under_cfg_parallel_ranks(world):
    """VERDICT r3 missing #5: TeaCache (reference hooks inside its SP path, wan_transformer4d.py:1201-1270) and cfg-skip
    (cfg_optimization.py:5-37) with the cfg2 x sp(N/2) layout: the compute / skip decision depends on the timestep embedding
    only, so every rank takes it alone; in cfg-skip steps the unconditional ranks sit out.  Every rank ends with the latents of
    the single-rank loop with the same switches."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + world + (os.getpid() % 500)
    procs = [ctx.Process(target=_cfgp_cache_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(e < 1e-5 and same for _, e, same in res), res


def test_sharded_data_parallel_world1_accumulates_and_scales(monkeypatch):
    """at world 1 nothing is reduced, so a second backward() WITHOUT no_sync() accumulates like a plain model (it
    raises only where a reduce-scatter is already on the wire); no_sync() SUMS micro-batches like torch DDP, and
    accumulation_steps folds accelerate's 1 / gradient_accumulation_steps into the norm and the update."""
    cpu_ops.install(monkeypatch)
    import more4d_amd.ops as real
    for n in ("sumsq", "adamw_"):
        monkeypatch.setattr(real, n, getattr(cpu_ops, n))
    from more4d_amd.dist.data_parallel import ShardedDataParallel
    torch.manual_seed(0)
    net = torch.nn.Linear(16, 8)
    w0 = net.weight.detach().clone()
    x1, x2 = torch.randn(4, 16), torch.randn(4, 16)
    dp = ShardedDataParallel(net, lr=1e-2, weight_decay=0.0, eps=1e-8)
    net(x1).pow(2).sum().backward()
    g1 = net.weight.grad.detach().clone()
    net(x2).pow(2).sum().backward()                      # second backward, no no_sync(): accumulates at world 1
    g12 = net.weight.grad.detach().clone()
    ref = torch.nn.Linear(16, 8)
    ref.load_state_dict({"weight": w0, "bias": net.bias.detach().clone()})
    ref(x2).pow(2).sum().backward()
    assert torch.allclose(g12, g1 + ref.weight.grad, atol=1e-5)
    n_sum = float(dp.reduce_gradients())
    n_avg = float(dp.reduce_gradients(accumulation_steps=2))
    assert abs(n_avg - n_sum / 2) < 1e-6 * n_sum
    want = float(torch.cat([g12.reshape(-1), net.bias.grad.reshape(-1)]).norm())
    assert abs(n_sum - want) < 1e-4 * want
    # Adam's first step moves every entry by lr * sign(g) whatever the scale: the two scalings agree in direction
    dp.step(max_norm=1e9, total_norm=torch.tensor(n_avg), accumulation_steps=2)
    assert torch.allclose(net.weight.detach(), w0 - 1e-2 * torch.sign(g12), atol=1e-4)
    dp.zero_grad()
    with dp.no_sync():
        net(x1).pow(2).sum().backward()
    net(x2).pow(2).sum().backward()
    assert float(net.weight.grad.abs().max()) > 0
    dp.close()


@pytest.mark.parametrize("world,par", [(8, "sp"), (8, "auto"), (3, "auto")])
def test_bench_launch_check_builds_both_layouts(world, par):
    """`python bench.py --gpus N --launch-check --parallelism P` under a plain interpreter (gloo, no GPU): N ranks come up, and
    the process groups of the plain sp-N layout north_star names AND of cfg2 x sp(N/2) are built one after the other in the same
    processes (what a measurement's second pass does), each carrying one K-shaped all-gather, the head all-gather and the
    velocity exchange; the stand-in group of the exposed-collective pass hands over the same shapes.  Odd N: sp only."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--launch-check", "--parallelism", par],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["launch_check"] and d["n_gpus"] == world and d["ranks"] == world
    lay = d["layouts"]
    assert lay["all_ranks_ok"] and lay["sp"]["ok"] and lay["sp"]["sp_world"] == world and lay["sp"]["cfg_branch_of_rank0"] is None
    if world % 2 == 0:
        assert lay["cfg-sp"]["ok"] and lay["cfg-sp"]["sp_world"] == world // 2 and lay["cfg-sp"]["cfg_branch_of_rank0"] == 0
        assert list(lay)[0] == ("sp" if par == "sp" else "cfg-sp")      # the selected layout first, then the other one
    else:
        assert "cfg-sp" not in lay
