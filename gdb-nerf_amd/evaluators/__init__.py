from .make_evaluator import make_evaluator  # noqa: F401
