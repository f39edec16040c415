"""CPU-side tests (no GPU): C-ABI library loads and exports every declared symbol, state-dict contract,
scheduler tables, and the HOST logic of the DiT / loop / sequence-parallel path with the kernels replaced
by tests/cpu_ops.py (test-only stand-in; the product has no CPU path)."""
import os
import re

import numpy as np
import pytest
import torch

import cpu_ops
from util import load_keys, load_npz, rel_err
from weights import fill

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TINY = dict(model_type="i2v", in_dim=64, dim=128, ffn_dim=512, num_heads=4, num_layers=2, text_dim=64, text_len=32,
            freq_dim=256, out_dim=16, add_ref_conv=True, use_dino_guidance=False, cross_attn_norm=True)


def test_library_exports_every_declared_symbol():
    import ctypes
    from more4d_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "more4d_hip.h")).read()
    declared = set(re.findall(r"\b(m4d_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -m more4d_amd.build"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    lib.m4d_version.restype = ctypes.c_int
    assert lib.m4d_version() >= 100
    # ... and nothing else: the m4d_* dynamic symbols are exactly the header's entry points
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (m4d_[a-z0-9_]+)", nm))
    assert exported == declared, exported ^ declared
    # every binding has exactly as many argtypes as the header declares parameters
    flat = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for name, params in re.findall(r"\b(m4d_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", flat):
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        assert n == len(_lib.SIGNATURES[name][1]), f"{name}: header has {n} parameters, binding {len(_lib.SIGNATURES[name][1])}"


def test_library_is_built_from_the_tracked_sources(tmp_path):
    """build() proves itself: the sha256 of csrc/ + include/ is compiled into the library (m4d_source_hash) and _lib.load()
    refuses a binary whose hash is not the tree's — a stale shipped .so cannot pass."""
    import ctypes
    import shutil
    import subprocess
    import sys
    from more4d_amd import _lib, build
    assert build.built_hash(_lib.LIB_PATH) == build.source_hash()
    lib = _lib.load()
    assert lib.m4d_source_hash().decode() == build.source_hash()
    # a library with another hash is refused: copy the package skeleton, change one source byte, keep the binary
    pkg = tmp_path / "more4d_amd"
    shutil.copytree(os.path.join(ROOT, "more4d_amd"), pkg, ignore=shutil.ignore_patterns("build", "__pycache__", "libmore4d_hip_*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    with open(pkg / "csrc" / "common.h", "a") as fh:
        fh.write("\n// edited after the build\n")
    code = "from more4d_amd import _lib\ntry:\n    _lib.load()\nexcept _lib.More4DHipError as e:\n    print('REFUSED', e)\n"
    r = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, capture_output=True, text=True,
                       env={**os.environ, "PYTHONPATH": str(tmp_path)})
    assert "REFUSED" in r.stdout and "other sources" in r.stdout, r.stdout + r.stderr


def test_product_path_fails_loudly_without_gpu():
    from more4d_amd import ops
    from more4d_amd._lib import More4DHipError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(More4DHipError):
        ops.gemm_bt(torch.zeros(8, 8), torch.zeros(8, 8))


def test_state_dict_contract():
    from more4d_amd.models import WanAttentionBlock, WanTransformer4DModel
    m = WanTransformer4DModel(**TINY)
    keys = load_keys("dit_tiny_keys.json")
    sd = m.state_dict()
    assert set(sd) == set(keys)
    assert all(tuple(sd[k].shape) == keys[k] for k in keys)
    for guid, name in ((False, "dit_block_keys.json"), (True, "dit_block_guid_keys.json")):
        blk = WanAttentionBlock("i2v_cross_attn", 256, 1024, 2, (-1, -1), True, True, 1e-6, use_spatial_guidance=guid)
        keys = load_keys(name)
        sd = {"blocks.0." + k: v for k, v in blk.state_dict().items()}
        assert set(sd) == set(keys) and all(tuple(sd[k].shape) == keys[k] for k in keys)


def test_scheduler_tables_match_reference():
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas
    z = load_npz("sched.npz")
    sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    sch.set_timesteps(sigmas=get_sampling_sigmas(50, 5.0))
    assert torch.equal(sch.timesteps, z["timesteps"]) and torch.equal(sch.sigmas, z["sigmas"])
    with pytest.raises(NotImplementedError):
        FlowDPMSolverMultistepScheduler(solver_order=4)
    with pytest.raises(NotImplementedError):
        FlowDPMSolverMultistepScheduler(algorithm_type="sde-dpmsolver++")


def _toy_velocity(x, t):
    return 0.3 * x + 0.1 * torch.sin(3.0 * x) + (float(t) / 1000.0 - 0.5)


@pytest.mark.parametrize("order,steps", [(2, 8), (2, 20), (3, 8), (3, 20)])
def test_multistep_solver_matches_reference(monkeypatch, order, steps):
    """DPM-Solver++ orders 2 / 3 (fm_solvers.py:486-677, :741-779): oracle and product scheduler (torch stand-ins for
    the kernels) against trajectories produced by the reference scheduler; the fused CFG entry point agrees with step()."""
    from oracle import sched as osch
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas
    cpu_ops.install(monkeypatch)
    z = load_npz("sched_multistep.npz")
    ref = z[f"o{order}_s{steps}"]
    ts, sigmas = osch.set_timesteps(osch.sampling_sigmas(steps, 5.0))
    traj = osch.dpmpp_multistep_loop(_toy_velocity, z["x0"], sigmas, ts, order)
    assert rel_err(torch.stack(traj), ref) < 2e-6
    sch = FlowDPMSolverMultistepScheduler(solver_order=order, shift=1.0)
    sch.set_timesteps(sigmas=get_sampling_sigmas(steps, 5.0))
    x, out = z["x0"].clone(), []
    for t in sch.timesteps:
        x = sch.step(_toy_velocity(x, t), t, x, return_dict=False)[0]
        out.append(x.clone())
    assert rel_err(torch.stack(out), ref) < 2e-6
    sch.set_timesteps(sigmas=get_sampling_sigmas(steps, 5.0))
    lat = z["x0"].clone()
    for i, t in enumerate(sch.timesteps):     # guidance g with v_u = v - d, v_c = v + (1/g - 1) d... use g = 1: v_c = v
        v = _toy_velocity(lat, t)
        sch.step_cfg_(lat, torch.stack([torch.zeros_like(v), v]), 1.0, i)
    assert rel_err(lat, ref[-1]) < 2e-6


def test_rope_tables_match_oracle():
    from oracle import dit as odit
    from more4d_amd.models.wan_transformer4d import build_rope_tables, rope_params
    d = 128
    freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                       rope_params(1024, 2 * (d // 6))], dim=1)
    cos, sin = build_rope_tables(freqs, (3, 4, 5), d, "cpu")
    oc, os_ = odit.rope_token_table(d, (3, 4, 5))
    assert torch.equal(cos, oc.float()) and torch.equal(sin, os_.float())


def tiny_cpu_model(monkeypatch):
    from more4d_amd.models import WanTransformer4DModel
    cpu_ops.install(monkeypatch)
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
    return m.eval()


def test_host_logic_forward_equals_reference(monkeypatch):
    """Token layout, ref row, seq_len padding, context cache, head/unpatchify bookkeeping."""
    z = load_npz("dit_tiny.npz")
    m = tiny_cpu_model(monkeypatch)
    ctx = [z["ctx0"], z["ctx1"]]
    with torch.no_grad():
        out = m(x=z["x"], t=z["t"], context=ctx, seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"], y=z["y"],
                full_ref=z["full_ref"])
        assert rel_err(out, z["out_ref"]) < 1e-4
        out = m(x=z["x"], t=z["t"], context=ctx, seq_len=int(z["seq_len"]), clip_fea=z["clip"], y=z["y"])
        assert rel_err(out, z["out_noref"]) < 1e-4


def test_host_logic_loop(monkeypatch):
    from more4d_amd.pipeline import denoise_latents
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
    z = load_npz("loop_tiny.npz")
    m = tiny_cpu_model(monkeypatch)
    sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    ts, _ = retrieve_timesteps(sch, sigmas=get_sampling_sigmas(int(z["steps"]), float(z["shift"])))
    with torch.no_grad():
        out = denoise_latents(m, sch, z["lat"], ts, float(z["guidance"]), [z["ctx_u"], z["ctx_c"]],
                              clip_fea=z["clip"], y=z["y"], full_ref=z["full_ref"], seq_len=256)
    assert rel_err(out, z["final"]) < 1e-3


def _sp_worker(rank, world, port, q):
    import os as _os
    torch.set_num_threads(max(1, (_os.cpu_count() or 8) // world))      # `world` processes share the host's cores
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
        init_sequence_parallel()
        m = WanTransformer4DModel(**TINY)
        m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
        m.eval()
        m.enable_multi_gpus_inference()
        z = load_npz("dit_tiny.npz")
        with torch.no_grad():
            out = m(x=z["x"], t=z["t"], context=[z["ctx0"], z["ctx1"]], seq_len=int(z["seq_len_pad"]),
                    clip_fea=z["clip"], y=z["y"], full_ref=z["full_ref"])
        if rank == 0:
            q.put(float(rel_err(out, z["out_ref"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sequence_parallel_equals_single_rank(world):
    """T-(token-)sharded forward under gloo == the reference output (ragged shards: 197 tokens over 2/3 ranks)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + world + (os.getpid() % 1000)
    procs = [ctx.Process(target=_sp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert err < 1e-4


def _sp_guid_worker(rank, world, port, q):
    import os as _os
    torch.set_num_threads(max(1, (_os.cpu_count() or 8) // world))      # `world` processes share the host's cores
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
        z, zg = load_npz("dit_tiny.npz"), load_npz("dit_tiny_guid_grads.npz")
        m = WanTransformer4DModel(**dict(TINY, use_omnimae_guidance=True))
        m.load_state_dict(fill(load_keys("dit_tiny_guid_keys.json"), 4321), strict=False)
        m.eval()
        kw = dict(x=z["x"], t=z["t"], context=[z["ctx0"], z["ctx1"]], seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"],
                  y=z["y"], full_ref=z["full_ref"], first_frame_features=(zg["patch"], zg["cls"]))
        with torch.no_grad():
            single = m(**kw)
            init_sequence_parallel()
            m.enable_multi_gpus_inference()
            multi = m(**kw)
        q.put((rank, float(rel_err(multi, single)), float(rel_err(single, zg["pred"]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sequence_parallel_with_spatial_guidance(world):
    """Token-sharded forward WITH spatial guidance (the released 4D-STraG config): each rank rotates the T-periodic guidance
    table to its first global token; N ranks == 1 rank == the reference's guided prediction."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + world + (os.getpid() % 1000)
    procs = [ctx.Process(target=_sp_guid_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(e_multi < 1e-5 and e_ref < 1e-4 for _, e_multi, e_ref in res), res


def _cfgp_worker(rank, world, port, q):
    import torch.distributed as dist
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))      # world processes share the host's cores
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import more4d_amd.ops as real
        for n in cpu_ops.NAMES:
            setattr(real, n, getattr(cpu_ops, n))
        from more4d_amd.dist import get_cfg_parallel_rank, init_sequence_parallel
        from more4d_amd.models import WanTransformer4DModel
        from more4d_amd.pipeline import denoise_latents
        from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps
        z = load_npz("loop_tiny.npz")
        m = WanTransformer4DModel(**TINY)
        m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
        m.eval()

        def loop():
            sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
            ts, _ = retrieve_timesteps(sch, sigmas=get_sampling_sigmas(int(z["steps"]), float(z["shift"])))
            with torch.no_grad():
                return denoise_latents(m, sch, z["lat"], ts[:4], float(z["guidance"]), [z["ctx_u"], z["ctx_c"]],
                                       clip_fea=z["clip"], y=z["y"], full_ref=z["full_ref"], seq_len=256)
        single = loop()                                  # CFG batched on one rank, no groups yet
        init_sequence_parallel(cfg_parallel=True)
        m.enable_multi_gpus_inference()
        assert get_cfg_parallel_rank() == rank // (world // 2) and m.sp_world_size == world // 2
        multi = loop()
        q.put((rank, float(rel_err(multi, single))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_cfg_parallel_loop_equals_single_rank(world):
    """CFG-parallel x token-sharded denoise loop under gloo (world 2 = one branch per rank, no per-layer collective; world 4
    = 2 branches x 2 token shards; world 8 = the cfg2 x sp4 layout of BASELINE configs[3]): every rank ends with the latents
    of the single-rank loop."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + world + (os.getpid() % 1000)
    procs = [ctx.Process(target=_cfgp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    errs = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(errs) == list(range(world)) and max(errs.values()) < 1e-5, errs


def test_teacache_and_guidance_adapter_host_logic(monkeypatch):
    """TeaCache: threshold 0 must reproduce the plain forward; a huge threshold must re-use the residual (second call
    == first call + nothing recomputed).  Guidance adapter: conv-SiLU-conv + bilinear resize equals torch's."""
    import torch.nn.functional as F
    from more4d_amd.models import WanTransformer4DModel
    cpu_ops.install(monkeypatch)
    z = load_npz("dit_tiny.npz")
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
    m.eval()
    args = dict(x=z["x"], context=[z["ctx0"], z["ctx1"]], seq_len=int(z["seq_len"]), clip_fea=z["clip"], y=z["y"])
    with torch.no_grad():
        base = m(t=z["t"], **args)
        base2 = m(t=z["t"] - 30, **args)
        m.enable_teacache([1.0, 0.0], num_steps=4, rel_l1_thresh=0.0, num_skip_start_steps=1)
        a = m(t=z["t"], **args)
        b = m(t=z["t"] - 30, **args)                       # threshold 0: always recomputed
        assert rel_err(a, base) < 1e-6 and rel_err(b, base2) < 1e-6
        m.enable_teacache([1.0, 0.0], num_steps=4, rel_l1_thresh=1e9, num_skip_start_steps=1)
        a = m(t=z["t"], **args)
        calls = []
        orig = m.blocks[0].run
        m.blocks[0].run = lambda *aa, **kk: calls.append(1) or orig(*aa, **kk)
        b = m(t=z["t"] - 30, **args)                       # skipped: blocks not executed, residual re-used
        assert not calls
        assert torch.isfinite(b).all() and rel_err(b, a) < 0.5
        m.disable_teacache()
    # guidance adapter
    g = WanTransformer4DModel(**{**TINY, "use_omnimae_guidance": True})
    g.eval()
    gen = torch.Generator().manual_seed(0)
    for p_ in g.feature_adapter.parameters():
        p_.data = torch.randn(p_.shape, generator=gen) * 0.02
    feats = torch.randn(2, 196, 768, generator=gen)
    with torch.no_grad():
        mine = g._adapt_features(feats, (6, 9))
        ref = g.feature_adapter(feats.view(2, 14, 14, 768).permute(0, 3, 1, 2))
        ref = F.interpolate(ref, size=(6, 9), mode="bilinear", align_corners=False).flatten(2).transpose(1, 2)
    assert rel_err(mine, ref) < 1e-5


@pytest.mark.parametrize("order,steps", [(2, 12), (2, 30), (3, 12), (3, 30)])
def test_unipc_solver_matches_reference(monkeypatch, order, steps):
    """UniPC (fm_solvers_unipc.py: UniC :486-626, UniP :350-484, step :655-739): oracle and product scheduler against
    trajectories of the reference scheduler on its own sigma table; a table starting at exactly sigma = 1 is refused."""
    from oracle import sched as osch
    from more4d_amd.utils.fm_solvers import get_sampling_sigmas
    from more4d_amd.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    cpu_ops.install(monkeypatch)
    z = load_npz("sched_unipc.npz")
    key = f"o{order}_lin{steps}"
    traj = osch.unipc_loop(_toy_velocity, z["x0"], z[key + "_sigmas"], z[key + "_timesteps"], order)
    assert rel_err(torch.stack(traj), z[key]) < 2e-6
    sch = FlowUniPCMultistepScheduler(solver_order=order, shift=1.0)
    sch.set_timesteps(steps, shift=5.0)
    assert torch.equal(sch.sigmas, z[key + "_sigmas"]) and torch.equal(sch.timesteps, z[key + "_timesteps"])
    x, out = z["x0"].clone(), []
    for t in sch.timesteps:
        x = sch.step(_toy_velocity(x, t), t, x, return_dict=False)[0]
        out.append(x.clone())
    assert rel_err(torch.stack(out), z[key]) < 2e-6
    if order == 3 and steps == 12:
        sch.set_timesteps(sigmas=get_sampling_sigmas(8, 5.0))
        x = z["x0"].clone()
        with pytest.raises(FloatingPointError):
            for t in sch.timesteps:
                x = sch.step(_toy_velocity(x, t), t, x, return_dict=False)[0]


def test_flow_match_euler_restatement(monkeypatch):
    """The diffusers default sampler is restated (parity unpinned): its tables follow the documented formula and its step
    is the Euler update of the pinned in-tree order-1 solver."""
    import numpy as np
    from more4d_amd.utils.flow_match_euler import FlowMatchEulerDiscreteScheduler
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler
    cpu_ops.install(monkeypatch)
    s = FlowMatchEulerDiscreteScheduler(num_train_timesteps=1000, shift=5.0)
    s.set_timesteps(50)
    raw = np.linspace(s.sigma_max * 1000, s.sigma_min * 1000, 50) / 1000
    want = 5.0 * raw / (1 + 4.0 * raw)
    assert np.allclose(s.sigmas[:-1].numpy(), want, rtol=1e-6) and float(s.sigmas[-1]) == 0.0
    assert s.timesteps.dtype == torch.float32 and np.allclose(s.timesteps.numpy(), want * 1000, rtol=1e-6)
    z = load_npz("sched.npz")
    d = FlowDPMSolverMultistepScheduler(solver_order=1, shift=1.0)
    d.set_timesteps(sigmas=z["sampling_sigmas"].numpy())
    e = FlowMatchEulerDiscreteScheduler(shift=1.0)
    e.set_timesteps(sigmas=z["sampling_sigmas"].numpy())
    xa = d.step(z["v"], d.timesteps[0], z["x"], return_dict=False)[0]
    xb = e.step(z["v"], e.timesteps[0], z["x"], return_dict=False)[0]
    assert rel_err(xb, z["x1"]) < 1e-6 and rel_err(xa, xb) < 1e-6
    noisy = e.scale_noise(z["x"], e.timesteps[3:4], z["v"])
    sg = float(e.sigmas[3])
    assert torch.allclose(noisy, sg * z["v"] + (1 - sg) * z["x"], atol=1e-6)


def test_omnimae_vit_host_logic(monkeypatch):
    """OmniMAE ViT-B front end (more4d_amd/models/omnimae.py): token / padding / K-V bookkeeping, packed stem, fused
    normalisation and the reference state-dict names, against the reference ViT's output (arithmetic by the stand-ins)."""
    from more4d_amd.models.omnimae import sinusoid_table, vit_base_mae_pretraining
    cpu_ops.install(monkeypatch)
    z = load_npz("omnimae.npz")
    m = vit_base_mae_pretraining(pretrained=False)
    assert torch.equal(m.trunk.pos_embed[0, :4, :8], z["pos_head"])
    assert rel_err(sinusoid_table(1568, 768)[0, 190:196, 760:], z["pos_tail"]) < 1e-6
    sd = fill(load_keys("omnimae_keys.json"), 555)
    res = m.load_state_dict(sd, strict=False)
    assert res.missing_keys == ["trunk.pos_embed"] and not res.unexpected_keys
    feats, cls = m.trunk.forward_patch_features(z["frame"], None, normalize=True)
    assert feats.shape == (2, 196, 768) and cls.shape == (2, 768)
    assert rel_err(feats, z["feats"]) < 1e-4 and rel_err(cls, z["cls"]) < 1e-4


def test_checkpoint_and_pointcloud_wire_formats(tmp_path):
    """save_pretrained <-> from_pretrained (diffusers directory layout, sharded and single-file), the sampler pickle of the
    training hooks and the point-cloud text dump."""
    import json
    import pickle
    import numpy as np
    from more4d_amd.models import WanTransformer4DModel
    from more4d_amd.utils import io
    m = WanTransformer4DModel(**TINY)
    m.load_state_dict(fill(load_keys("dit_tiny_keys.json"), 1234))
    for sub, shard_gb in (("one", 10.0), ("many", 2e-4)):
        d = str(tmp_path / sub)
        m.save_pretrained(d, max_shard_size_gb=shard_gb)
        names = sorted(os.listdir(d))
        assert "config.json" in names and (len(names) == 2) == (sub == "one")
        assert json.load(open(os.path.join(d, "config.json")))["dim"] == TINY["dim"]
        m2 = WanTransformer4DModel.from_pretrained(d, torch_dtype=torch.float32)
        assert all(torch.equal(v, m2.state_dict()[k]) for k, v in m.state_dict().items())
    io.save_sampler_state(str(tmp_path), 1000, 3)
    assert pickle.load(open(tmp_path / "sampler_pos_start.pkl", "rb")) == [1000, 3]
    assert io.load_sampler_state(str(tmp_path), dataloader_num_workers=4, num_processes=8) == (936, 3)
    assert io.load_sampler_state(str(tmp_path / "one")) is None
    g = torch.Generator().manual_seed(0)
    recon = torch.randn(1, 3, 3, 2, 4, generator=g)
    first = torch.randn(1, 3, 1, 2, 4, generator=g)
    coords = io.recover_coords(recon, first)
    assert coords.shape == (1, 3, 3, 2, 4) and torch.equal(coords[:, :, 0], first[:, :, 0])
    assert torch.allclose(coords[:, :, 2], recon[:, :, 2] + first[:, :, 0])
    colors = io.image_colors(torch.rand(1, 3, 2, 4, generator=g) * 2 - 1)
    files = io.save_pointcloud_data(coords, colors, "vid", str(tmp_path), 7)
    assert [os.path.basename(f) for f in files] == [f"vid_frame_{i:04d}.txt" for i in range(3)]
    rows = np.loadtxt(files[1])
    assert rows.shape == (8, 6) and np.allclose(rows[:, :3], coords[0, :, 1].permute(1, 2, 0).reshape(-1, 3).numpy())
    assert np.array_equal(rows[:, 3:], colors[0].float().numpy())


def test_bench_flop_accounting_matches_survey():
    """bench.py's algorithmic FLOPs == SURVEY §8d: F(L) = 40 [12 L d^2 + 4 L^2 d + 4 L d ffn + 4 L 769 d] + embed =
    837.9 TFLOP (L = 20 280) / 930.0 TFLOP (L = 21 840) per forward, cached context K/V excluded (0.35 %)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for L, want in ((20280, 837.9e12), (21840, 930.0e12)):
        g, a = bench.flops_per_forward(bench.CFG_14B, L, 1)
        assert abs((g + a) - want) / want < 6e-3, (L, (g + a) / 1e12)
    g2, a2 = bench.flops_per_forward(bench.CFG_14B, 21840, 2)
    assert abs((g2 + a2) / 1e12 - 1853.4) < 1.0
    assert bench.MFMA_BF16_PEAK_TF == 2500.0


def test_teacache_coefficients_match_reference():
    """The TeaCache rescaling polynomial per model name == the reference's get_teacache_coefficients (fixture generated from it)."""
    import json
    from more4d_amd.models.cache_utils import TeaCache, get_teacache_coefficients
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "teacache_coeffs.json")))
    for name, coeff in ref.items():
        assert get_teacache_coefficients(name) == coeff, name
    assert ref["Wan2.1-Fun-V1.1-14B-Control"] is None      # the reference's table does not know its own released checkpoint name
    tc = TeaCache(ref["Wan2.1-Fun-14B-InP"], num_steps=50, rel_l1_thresh=0.1, num_skip_start_steps=5)
    assert tc.cnt == 0 and tc.should_calc and tc.previous_residual_cond is None and abs(float(tc.rescale_func(0.1)) - float(
        np.polyval(ref["Wan2.1-Fun-14B-InP"], 0.1))) < 1e-12
    tc.cnt = 7
    tc.reset()
    assert tc.cnt == 0
    for bad in (dict(num_steps=0), dict(num_steps=5, rel_l1_thresh=-1.0), dict(num_steps=5, num_skip_start_steps=6)):
        with pytest.raises(ValueError):
            TeaCache([1.0, 0.0], **bad)


def test_riflex_table_matches_reference():
    """enable_riflex (reference :1011-1025 / :264-321): the edited temporal frequency table, sampled rows (tests/golden/riflex.npz)."""
    from more4d_amd.models import WanTransformer4DModel
    z = load_npz("riflex.npz")
    rows = torch.tensor([0, 1, 65, 66, 1023])
    for d, heads, k in ((32, 4, 2), (128, 1, 6)):
        m = WanTransformer4DModel(**dict(TINY, dim=d * heads, num_heads=heads, ffn_dim=64, num_layers=1))
        base = m.freqs.clone()
        m.enable_riflex(k=k, L_test=66, L_test_scale=4.886)
        da = d - 4 * (d // 6)
        got = m.freqs[rows, :da // 2]
        assert rel_err(got.real, z[f"re{d}"]) < 1e-12 and rel_err(got.imag, z[f"im{d}"]) < 1e-12
        assert torch.equal(m.freqs[:, da // 2:], base[:, da // 2:])          # spatial axes untouched
        m.disable_riflex()
        assert torch.equal(m.freqs, base)


def test_scheduler_entry_points_match_reference(monkeypatch):
    """set_begin_index / index_for_timestep / add_noise / __len__ / config of FlowDPMSolverMultistepScheduler against the
    reference scheduler (tests/golden/sched_api.npz)."""
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas
    cpu_ops.install(monkeypatch)
    z = load_npz("sched_api.npz")
    sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    sch.set_timesteps(sigmas=get_sampling_sigmas(50, 5.0))
    assert len(sch) == int(z["length"]) and sch.config.num_train_timesteps == 1000 and sch.order == 1 and sch.begin_index is None
    assert [sch.index_for_timestep(t) for t in z["ts"]] == z["idx"].tolist()
    assert rel_err(sch.add_noise(z["x"], z["n"], z["ts"]), z["noisy_lookup"]) < 1e-6
    sch.set_begin_index(10)
    assert rel_err(sch.add_noise(z["x"], z["n"], z["ts"]), z["noisy_begin"]) < 1e-6
    out = sch.step(z["v"], sch.timesteps[10], z["x"], return_dict=False)[0]
    assert rel_err(out, z["x_step"]) < 1e-6 and sch.step_index == int(z["step_index_after"])
    assert rel_err(sch.add_noise(z["x"], z["n"], z["ts"]), z["noisy_mid"]) < 1e-6


def test_production_kernels_do_not_spill():
    """Code-object metadata of the built objects (no GPU needed): the production MFMA kernels use no scratch memory.  A spill in one of
    them is a build regression (hipcc under register pressure), and scratch loads share vmcnt with the hand-counted tile waits."""
    import importlib.util
    import glob
    spec = importlib.util.spec_from_file_location("kernel_regs", os.path.join(ROOT, "tools", "kernel_regs.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    objs = {os.path.basename(o): o for o in glob.glob(os.path.join(ROOT, "more4d_amd", "build", "*.hip.o")) if ".abl." not in o and o.count(".") == 2}
    if "attention.hip.o" not in objs:
        pytest.skip("objects not built (python -m more4d_amd.build)")
    want = {"attention.hip.o": ("attn128p_kernel", "attn128x_kernel", "attn128_kernel"),
            "attention_bwd.hip.o": ("attn_bwd_kvp_kernel", "attn_bwd_dqp_kernel"),
            "gemm_wide_store.hip.o": ("gemm_bt256w_kernel",), "gemm_wide_resid.hip.o": ("gemm_bt256w_kernel",),
            "gemm_wide_gelu.hip.o": ("gemm_bt256w_kernel",), "conv.hip.o": ("conv_halo_kernel",)}
    seen = 0
    for obj, names in want.items():
        if obj not in objs:
            continue
        for k in kr.kernels(objs[obj]):
            if any(n in k["name"] for n in names):
                seen += 1
                assert int(k["scratch"]) == 0 and int(k["spill"]) == 0, (obj, k)
    assert seen >= 8
