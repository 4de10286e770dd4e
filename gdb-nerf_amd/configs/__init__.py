from .config import make_cfg, DEFAULTS  # noqa: F401
