"""Import the reference's hot-path modules in THIS container (fixture generation only).

The reference (/root/reference, read-only, never shipped) needs `diffusers` and its own
missing `MoRe4D.dist` package.  This helper registers minimal
stand-in *names* for those third-party imports (our own code, no reference source) and
loads the five reference files by path, bypassing `MoRe4D/models/__init__.py`.

Used ONLY by tests/golden/make_golden.py.  Nothing under tests/ that runs on the GPU box
imports this file; the GPU box has no /root/reference.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("MORE4D_REFERENCE", "/root/reference")


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave like a package
    sys.modules[name] = m
    return m


def _install_third_party_names():
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_more4d_stub", False):
        return
    d = _mod("diffusers")
    d._more4d_stub = True

    cu = _mod("diffusers.configuration_utils")

    class ConfigMixin:
        pass

    class _Cfg(dict):
        __getattr__ = dict.__getitem__

    def register_to_config(fn):
        # published diffusers behaviour: ctor kwargs (with defaults) become self.config
        import functools
        import inspect
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def wrapper(self, *a, **k):
            ba = sig.bind(self, *a, **k)
            ba.apply_defaults()
            cfg = _Cfg({n: v for n, v in ba.arguments.items() if n != "self"})
            object.__setattr__(self, "config", cfg)
            return fn(self, *a, **k)
        return wrapper

    cu.ConfigMixin = ConfigMixin
    cu.register_to_config = register_to_config

    _mod("diffusers.loaders")
    sf = _mod("diffusers.loaders.single_file_model")

    class FromOriginalModelMixin:
        pass

    sf.FromOriginalModelMixin = FromOriginalModelMixin

    _mod("diffusers.models")
    mu = _mod("diffusers.models.modeling_utils")

    class ModelMixin(nn.Module):
        @property
        def dtype(self):
            return next(self.parameters()).dtype

    mu.ModelMixin = ModelMixin

    ut = _mod("diffusers.utils")
    import logging as _logging

    class _L:
        @staticmethod
        def get_logger(name):
            return _logging.getLogger(name)

    ut.logging = _L
    ut.is_torch_version = lambda op, v: True
    ut.deprecate = lambda *a, **k: None
    ut.is_scipy_available = lambda: False

    tu = _mod("diffusers.utils.torch_utils")

    def randn_tensor(shape, generator=None, device=None, dtype=None):
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)

    tu.randn_tensor = randn_tensor

    au = _mod("diffusers.utils.accelerate_utils")
    au.apply_forward_hook = lambda fn: fn

    _mod("diffusers.models.autoencoders")
    vae = _mod("diffusers.models.autoencoders.vae")

    class DecoderOutput:
        def __init__(self, sample):
            self.sample = sample

    class DiagonalGaussianDistribution:
        # published diffusers semantics: chunk, clamp logvar to [-30, 20]
        def __init__(self, parameters):
            self.parameters = parameters
            self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
            self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
            self.std = torch.exp(0.5 * self.logvar)
            self.var = torch.exp(self.logvar)

        def kl(self, other=None):        # published: summed over dims [1, 2, 3] (the 5-D video latent keeps its last axis)
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])

        def sample(self, generator=None):
            eps = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
            return self.mean + self.std * eps

        def mode(self):
            return self.mean

    vae.DecoderOutput = DecoderOutput
    vae.DiagonalGaussianDistribution = DiagonalGaussianDistribution

    mo = _mod("diffusers.models.modeling_outputs")

    class AutoencoderKLOutput:
        def __init__(self, latent_dist):
            self.latent_dist = latent_dist

        def __getitem__(self, i):        # published BaseOutput behaviour: out[0] is the first field
            return (self.latent_dist,)[i]

    mo.AutoencoderKLOutput = AutoencoderKLOutput

    _mod("diffusers.schedulers")
    su = _mod("diffusers.schedulers.scheduling_utils")
    import enum

    class KarrasDiffusionSchedulers(enum.Enum):
        X = 0

    class SchedulerMixin:
        pass

    class SchedulerOutput:
        def __init__(self, prev_sample):
            self.prev_sample = prev_sample

    su.KarrasDiffusionSchedulers = KarrasDiffusionSchedulers
    su.SchedulerMixin = SchedulerMixin
    su.SchedulerOutput = SchedulerOutput


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    """Returns a namespace with the reference modules: dit, vae, traj, cfg, fm."""
    _install_third_party_names()
    pkg = _mod("MoRe4D")
    _mod("MoRe4D.models")
    dist = _mod("MoRe4D.dist")
    dist.get_sequence_parallel_rank = lambda: 0
    dist.get_sequence_parallel_world_size = lambda: 1
    dist.get_sp_group = lambda: None
    dist.usp_attn_forward = None
    dist.xFuserLongContextAttention = None
    utils = _mod("MoRe4D.utils")
    cfg = _load("MoRe4D.utils.cfg_optimization", "MoRe4D/utils/cfg_optimization.py")
    utils.cfg_skip = cfg.cfg_skip
    _load("MoRe4D.models.cache_utils", "MoRe4D/models/cache_utils.py")
    dit = _load("MoRe4D.models.wan_transformer4d", "MoRe4D/models/wan_transformer4d.py")
    vae = _load("MoRe4D.models.wan_vae", "MoRe4D/models/wan_vae.py")
    traj = _load("MoRe4D.models.trajectory_module", "MoRe4D/models/trajectory_module.py")
    fm = _load("MoRe4D.utils.fm_solvers", "MoRe4D/utils/fm_solvers.py")
    ns = types.SimpleNamespace(dit=dit, vae=vae, traj=traj, cfg=cfg, fm=fm)
    return ns
