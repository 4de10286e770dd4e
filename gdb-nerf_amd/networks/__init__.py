from .make_network import make_network  # noqa: F401
