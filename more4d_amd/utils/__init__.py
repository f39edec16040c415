from .cfg_optimization import cfg_skip  # noqa: F401
