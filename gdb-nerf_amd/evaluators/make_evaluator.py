"""Plugin factory, same contract as the reference's evaluators/make_evaluator.py:4-15."""
from ..networks.make_network import load_source


def make_evaluator(cfg):
    if cfg.skip_eval:
        return None
    return load_source(cfg.evaluator_module, cfg.evaluator_path).Evaluator(cfg)
