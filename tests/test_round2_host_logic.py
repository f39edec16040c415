"""CPU-side regression tests for host-logic defects found by the round-1 review: stale version-keyed caches after the
raw-pointer AdamW update, the pipeline's per-scheduler-type timestep dispatch (pipeline_wan_fun_control.py:576-590), the
CFG halves of a batched velocity in the multistep / UniPC fused step, cfg-skip with tuple-valued guidance features, and
the single gradient-norm pass of the train step.  Kernels are replaced by tests/cpu_ops.py (test-only stand-in)."""
import numpy as np
import pytest
import torch

import cpu_ops
from util import load_keys, load_npz, rel_err
from weights import fill

TINY = dict(model_type="i2v", in_dim=64, dim=128, ffn_dim=512, num_heads=4, num_layers=2, text_dim=64, text_len=32,
            freq_dim=256, out_dim=16, add_ref_conv=True, use_dino_guidance=False, cross_attn_norm=True)


def _toy_velocity(x, t):
    return 0.3 * x + 0.1 * torch.sin(3.0 * x) + (float(t) / 1000.0 - 0.5)


def test_adamw_bumps_version_and_f32_cache_refreshes(monkeypatch):
    """ops.adamw_ writes through raw pointers; AdamW.step must bump the parameter's version so the (data_ptr, _version)
    keyed fp32 copies / packed weights are rebuilt (bf16 parameters: the copy is a different tensor)."""
    from more4d_amd.models.wan_transformer4d import _f32
    from more4d_amd.optim import AdamW
    cpu_ops.install(monkeypatch)
    import more4d_amd.ops as real
    monkeypatch.setattr(real, "adamw_", cpu_ops.adamw_)
    p = torch.nn.Parameter(torch.randn(4, 8).to(torch.bfloat16))
    cache = {}
    before = _f32(p, cache).clone()
    assert torch.equal(before, p.detach().float())
    v0 = p._version
    p.grad = torch.ones_like(p)
    opt = AdamW([p], lr=0.1, weight_decay=0.0)
    opt.step()
    assert p._version > v0
    after = _f32(p, cache)
    assert torch.equal(after, p.detach().float())
    assert not torch.equal(after, before), "the parameter moved; the fp32 copy must follow"


def test_bf16_training_moves_gates_and_modulation(monkeypatch):
    """Multi-step bf16 training with the product AdamW: parameters read through the fp32 cache (block modulation, q/k norm
    weights, norm3 affine) keep changing between steps, i.e. the forward of step n+1 sees the update of step n."""
    from more4d_amd.models import WanTransformer4DModel
    from more4d_amd.optim import AdamW
    cpu_ops.install(monkeypatch)
    import more4d_amd.ops as real
    monkeypatch.setattr(real, "adamw_", cpu_ops.adamw_)
    monkeypatch.setattr(real, "sumsq", cpu_ops.sumsq)
    z = load_npz("dit_tiny.npz")
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
    m = m.to(torch.bfloat16).train()
    opt = AdamW(m.parameters(), lr=1e-2, weight_decay=0.0)
    tgt = torch.randn(z["x"].shape, generator=torch.Generator().manual_seed(0))
    preds = []
    for _ in range(3):
        pred = m(x=z["x"].to(torch.bfloat16), t=z["t"], context=[z["ctx0"], z["ctx1"]], seq_len=int(z["seq_len_pad"]),
                 clip_fea=z["clip"], y=z["y"].to(torch.bfloat16), full_ref=z["full_ref"].to(torch.bfloat16))
        preds.append(pred.detach().float().clone())
        ((pred.float() - tgt) ** 2).mean().backward()
        opt.step()
        opt.zero_grad()
        cache = getattr(m, "_f32cache", None)
        if cache is not None:
            for name, p in m.named_parameters():
                hit = cache.get(id(p))
                if hit is not None and p.dtype != torch.float32:
                    # whatever the cache holds must be rebuilt on the next read
                    from more4d_amd.models.wan_transformer4d import _f32
                    assert torch.equal(_f32(p, cache), p.detach().float()), name
    assert not torch.equal(preds[0], preds[1]) and not torch.equal(preds[1], preds[2])


@pytest.mark.parametrize("kind", ["euler", "unipc", "dpm"])
def test_pipeline_timestep_dispatch(kind):
    from more4d_amd.pipeline import WanFunControlPipeline
    from more4d_amd.utils.flow_match_euler import FlowMatchEulerDiscreteScheduler
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler
    from more4d_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    if kind == "euler":      # the reference's default "Flow" sampler: shifted ONCE by its own config.shift (infer.py:670-682)
        sch = FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=5.0)
        pipe = WanFunControlPipeline(scheduler=sch)
        ts = pipe._prepare_timesteps(50, "cpu", None, shift=5)
        raw = np.linspace(sch.sigma_max * 1000, sch.sigma_min * 1000, 50) / 1000
        want = 5.0 * raw / (1 + 4.0 * raw)
        assert np.allclose(sch.sigmas[:-1].numpy(), want, rtol=1e-6)
        assert np.allclose(ts.numpy(), want * 1000, rtol=1e-6) and len(ts) == 50
    elif kind == "unipc":    # set_timesteps(n, shift=shift): the reference's own table (sched_unipc.npz)
        z = load_npz("sched_unipc.npz")
        sch = FlowUniPCMultistepScheduler(solver_order=2, shift=1.0)
        pipe = WanFunControlPipeline(scheduler=sch)
        ts = pipe._prepare_timesteps(12, "cpu", None, shift=5)
        assert torch.equal(ts, z["o2_lin12_timesteps"]) and torch.equal(sch.sigmas, z["o2_lin12_sigmas"])
        assert float(sch.sigmas[0]) < 1.0          # no sigma == 1 head: the order-2 corrector is finite
    else:
        z = load_npz("sched.npz")
        sch = FlowDPMSolverMultistepScheduler(solver_order=1, shift=1.0)
        pipe = WanFunControlPipeline(scheduler=sch)
        ts = pipe._prepare_timesteps(50, "cpu", None, shift=5)
        assert torch.equal(ts, z["timesteps"]) and torch.equal(sch.sigmas, z["sigmas"])


@pytest.mark.parametrize("kind,order", [("dpm", 2), ("dpm", 3), ("unipc", 2)])
def test_step_cfg_batch2_uses_cfg_halves(monkeypatch, kind, order):
    """v is [2B, ...] = (uncond samples..., cond samples...): with B = 2 the fused multistep / UniPC step must combine
    sample b's two halves, i.e. equal B independent B = 1 runs."""
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas
    from more4d_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    cpu_ops.install(monkeypatch)
    import more4d_amd.ops as real
    monkeypatch.setattr(real, "lincomb", cpu_ops.lincomb)
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(2, 16, 2, 4, 4, generator=g)
    gs = 3.0

    def make():
        if kind == "dpm":
            s = FlowDPMSolverMultistepScheduler(solver_order=order, shift=1.0)
            s.set_timesteps(sigmas=get_sampling_sigmas(8, 5.0))
        else:
            s = FlowUniPCMultistepScheduler(solver_order=order, shift=1.0)
            s.set_timesteps(8, shift=5.0)
        return s

    def vel(x, t):      # distinct unconditional / conditional velocities
        return torch.cat([_toy_velocity(x, t) * 0.5, _toy_velocity(x, t) + 0.1 * x])

    sch = make()
    lat = x0.clone()
    for i, t in enumerate(sch.timesteps):
        sch.step_cfg_(lat, vel(lat, t).contiguous(), gs, i)
    assert lat.shape == x0.shape
    for b in range(2):
        s1 = make()
        l1 = x0[b:b + 1].clone()
        for i, t in enumerate(s1.timesteps):
            s1.step_cfg_(l1, vel(l1, t).contiguous(), gs, i)
        assert rel_err(lat[b:b + 1], l1) < 1e-6


def test_cfg_skip_slices_tuple_features():
    """cfg_skip with first_frame_features = (patch [2B,..], cls [2B,..]): every member's batch axis is halved (the tuple
    itself is not sliced), context lists keep the reference's per-sample slicing."""
    from more4d_amd.utils.cfg_optimization import cfg_skip

    class M:
        cfg_skip_ratio, current_steps, num_inference_steps = 0.5, 9, 10
        seen = None

        @cfg_skip()
        def forward(self, x, t, context=None, first_frame_features=None, y=None):
            M.seen = (x, t, context, first_frame_features, y)
            return x * 2

    x = torch.arange(4.0).view(4, 1)
    patch, cls = torch.arange(8.0).view(4, 2), torch.arange(4.0).view(4, 1)
    out = M().forward(x, torch.arange(4), context=[0, 1, 2, 3], first_frame_features=(patch, cls), y=torch.ones(4, 3))
    sx, st, sc, sf, sy = M.seen
    assert sx.shape[0] == 2 and st.tolist() == [2, 3] and sc == [2, 3] and sy.shape[0] == 2
    assert isinstance(sf, tuple) and len(sf) == 2
    assert torch.equal(sf[0], patch[2:]) and torch.equal(sf[1], cls[2:])
    assert out.shape[0] == 4 and torch.equal(out[:2], out[2:])


def test_train_step_single_norm_pass(monkeypatch):
    """train_step computes the global gradient norm once (two sumsq passes over 33 GB at 14B)."""
    from more4d_amd import optim, training
    calls = []
    real_gn = optim.grad_norm

    def counting(params):
        calls.append(1)
        return torch.tensor(2.0)
    monkeypatch.setattr(optim, "grad_norm", counting)
    monkeypatch.setattr(training, "grad_norm", counting)
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3)
    total = optim.clip_grad_norm_([p], 1.0, total_norm=torch.tensor(2.0))
    assert not calls and float(total) == 2.0 and torch.allclose(p.grad, torch.full((3,), 0.5), atol=1e-5)
    del real_gn


def test_denoise_latents_runs_without_outer_no_grad(monkeypatch):
    """denoise_latents on a model whose parameters require grad must take the inference path (it is a sampler)."""
    from more4d_amd.models import WanTransformer4DModel
    from more4d_amd.pipeline import denoise_latents
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
    cpu_ops.install(monkeypatch)
    z = load_npz("loop_tiny.npz")
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
    m.train()
    assert any(p.requires_grad for p in m.parameters())
    sch = FlowDPMSolverMultistepScheduler(solver_order=1, shift=1.0)
    ts, _ = retrieve_timesteps(sch, device="cpu", sigmas=get_sampling_sigmas(2, 5.0))
    out = denoise_latents(m, sch, z["lat"], ts, 6.0, [z["ctx_u"], z["ctx_c"]], clip_fea=z["clip"], y=z["y"],
                          full_ref=z["full_ref"], seq_len=256)
    assert not out.requires_grad and torch.isfinite(out).all()


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` under a plain interpreter (no torchrun): the script re-execs itself under
    torch.distributed.run, the two ranks rendezvous on 127.0.0.1 and count themselves with an all-reduce (gloo here, RCCL on
    a GPU node).  VERDICT r1 item 1: the driver's `--gpus N` form must return rc 0."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["launch_check"] and d["n_gpus"] == 2 and d["ranks"] == 2


def test_bench_flags_environment_overrides(monkeypatch):
    import bench
    monkeypatch.setenv("M4D_GEMM_VARIANT", "1")
    assert bench.m4d_overrides() == ["M4D_GEMM_VARIANT"]
    monkeypatch.delenv("M4D_GEMM_VARIANT")
    assert [k for k in bench.m4d_overrides() if k != "M4D_LIB"] == []


def test_shipping_library_has_no_ablation_switches():
    """The timing ablations (kernels that skip work) are compiled out of the shipping library: their environment variables
    do not even appear in it."""
    from more4d_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"M4D_GEMM_ABL" not in blob and b"M4D_ATTN_ABL" not in blob
    assert b"M4D_GEMM_VARIANT" in blob          # (the A/B switches between correct kernels are still there)


def _teacache_loop(m, z, tag, dev="cpu", dtype=torch.float32):
    """The loop of make_golden.py:make_teacache_loop on the product model; returns (decisions, trajectory)."""
    from more4d_amd.models.cache_utils import get_teacache_coefficients
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas
    steps = int(z["steps"])
    coeff = get_teacache_coefficients("Wan2.1-Fun-14B-Control")
    assert np.allclose(coeff, z["coeff"].numpy())
    m.enable_teacache(coeff, steps, float(z[f"{tag}_thresh"]), num_skip_start_steps=1, offload=False)
    sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    sch.set_timesteps(sigmas=get_sampling_sigmas(steps, float(z["shift"])), device=dev)
    x = z["lat"].to(dev).float().clone()
    y2 = torch.cat([z["y"]] * 2).to(dev, dtype)
    ref2 = torch.cat([z["full_ref"]] * 2).to(dev, dtype)
    clip2 = torch.cat([z["clip"]] * 2).to(dev)
    ctx = [z["ctx_u"].to(dev), z["ctx_c"].to(dev)]
    calc, traj = [], []
    with torch.no_grad():
        for i, t in enumerate(sch.timesteps):
            v = m(x=torch.cat([x, x]).to(dtype), t=t.to(dev).expand(2), context=ctx, seq_len=256, clip_fea=clip2, y=y2, full_ref=ref2)
            calc.append(bool(m.should_calc))
            sch.step_cfg_(x, v.contiguous(), float(z["guidance"]), i, round_dtype=dtype)
            traj.append(x.clone())
    m.disable_teacache()
    return calc, torch.stack(traj)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_teacache_matches_reference_run(monkeypatch, tag):
    """TeaCache (cache_utils.py:19-74 + the hooks wan_transformer4d.py:1201-1270, 1336-1339) pinned to a reference run: the
    same compute / skip decision at every step and the same latent trajectory (host logic over the torch stand-ins)."""
    from more4d_amd.models import WanTransformer4DModel
    cpu_ops.install(monkeypatch)
    import more4d_amd.ops as real
    for n in ("rel_l1", "axpby"):
        monkeypatch.setattr(real, n, getattr(cpu_ops, n))
    z = load_npz("teacache_loop.npz")
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
    m.eval()
    calc, traj = _teacache_loop(m, z, tag)
    assert calc == [bool(c) for c in z[f"{tag}_calc"]], (calc, z[f"{tag}_calc"])
    assert not all(calc) and any(calc[1:])
    assert rel_err(traj, z[f"{tag}_traj"]) < 2e-4
    assert rel_err(traj[-1], z[f"{tag}_final"]) < 2e-4


@pytest.mark.parametrize("mode,world", [("allgather", 2), ("allgather", 3), ("ulysses", 2), ("ulysses", 4)])
def test_sequence_parallel_schedules_in_process(monkeypatch, mode, world):
    """The N-rank token-sharded forward (all-gather of K / V^T with the local-first merge, and the Ulysses head-split all-to-all)
    driven by N threads over the in-process group (more4d_amd.dist.emulation): every rank's output == the single-rank output ==
    the reference's (ragged: 197 tokens + ref row over 2 / 3 / 4 ranks, padded key rows masked)."""
    import copy
    from more4d_amd.dist import SequenceParallelGroup
    from more4d_amd.dist.emulation import run_ranks
    from more4d_amd.models import WanTransformer4DModel
    cpu_ops.install(monkeypatch)
    monkeypatch.setattr(SequenceParallelGroup, "mode", mode)
    z = load_npz("dit_tiny.npz")
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
    m.eval()
    kw = dict(x=z["x"], t=z["t"], context=[z["ctx0"], z["ctx1"]], seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"], y=z["y"],
              full_ref=z["full_ref"])
    with torch.no_grad():
        single = m(**kw)
    assert rel_err(single, z["out_ref"]) < 1e-4

    def rank_fn(group):
        mr = copy.copy(m)
        mr.sp_world_size, mr.sp_world_rank, mr._sp = group.world_size, group.rank, group
        mr.all_gather = group.all_gather
        with torch.no_grad():
            return mr(**kw)
    outs = run_ranks(world, rank_fn)
    for o in outs:
        assert rel_err(o, single) < 1e-5


def _ulysses_gloo_worker(rank, world, port, q):
    import os as _os
    torch.set_num_threads(max(1, (_os.cpu_count() or 8) // world))      # `world` processes share the host's cores
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import more4d_amd.ops as real
        for n in cpu_ops.NAMES:
            setattr(real, n, getattr(cpu_ops, n))
        from more4d_amd.dist import init_sequence_parallel
        from more4d_amd.models import WanTransformer4DModel
        z = load_npz("dit_tiny.npz")
        m = WanTransformer4DModel(**TINY)
        m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
        m.eval()
        g = init_sequence_parallel(mode="ulysses")
        x = torch.arange(world * 3, dtype=torch.float32).view(world, 3) + 100 * rank
        got = g.all_to_all(x)
        want = torch.stack([torch.arange(world * 3, dtype=torch.float32).view(world, 3)[rank] + 100 * j for j in range(world)])
        m.enable_multi_gpus_inference()
        with torch.no_grad():
            out = m(x=z["x"], t=z["t"], context=[z["ctx0"], z["ctx1"]], seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"], y=z["y"],
                    full_ref=z["full_ref"])
        if rank == 0:
            q.put((bool(torch.equal(got, want)), float(rel_err(out, z["out_ref"]))))
    finally:
        from more4d_amd.dist import SequenceParallelGroup
        SequenceParallelGroup.mode = "allgather"
        dist.destroy_process_group()


def test_ulysses_under_gloo_world2():
    """The all-to-all plumbing of SequenceParallelGroup over a real process group (gloo) + the Ulysses forward == the reference."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_ulysses_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, err = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok and err < 1e-4
