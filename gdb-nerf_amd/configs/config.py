"""Config loader compatible with the reference's configs/config.py (same keys, same precedence:
defaults < parent_cfg < YAML < `key value` overrides; `*_module` -> `*_path`), but as a function with
no import-time argument parsing, no required `workspace` environment variable and no git calls."""
import ast
import copy
import os
from types import SimpleNamespace
from typing import Optional, Sequence

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))

DEFAULTS = {
    "save_tag": "default", "exp_name_tag": "", "exp_name": "default", "local_rank": 0, "write_video": False, "fps": 24,
    "distributed": False, "task": "hello", "gpus": [0], "resume": True, "ep_iter": -1, "save_ep": 1, "save_latest_ep": 1,
    "eval_ep": 1, "log_interval": 20, "sample_on_mask": False, "skip_eval": False, "fix_random": False,
    "save_result": False, "eval_lpips": False,
    "train": {"pretrain": "", "epoch": 10000, "num_workers": 8, "collator": "default", "batch_sampler": "default", "shuffle": True,
              "eps": 1.0e-8, "sampler_meta": {"input_views_num": [], "input_views_prob": []}, "optim": "adam", "lr": 5.0e-4,
              "weight_decay": 0.0, "scheduler": {"type": "multi_step", "milestones": [80, 120, 200, 240], "gamma": 0.5}, "batch_size": 4},
    "test": {"batch_size": 1, "collator": "default", "epoch": -1, "batch_sampler": "default",
             "sampler_meta": {"input_views_num": [], "input_views_prob": []}, "eval_depth": False, "eval_center": False},
}


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _namespace(d):
    if isinstance(d, dict):
        return SimpleNamespace(**{k: _namespace(v) for k, v in d.items()})
    return d


def _read(path: str) -> dict:
    if not os.path.exists(path) and os.path.exists(os.path.join(os.path.dirname(HERE), path)):
        path = os.path.join(os.path.dirname(HERE), path)  # reference-style "configs/x.yaml" resolves inside this package
    with open(path) as f:
        return yaml.safe_load(f) or {}


def make_cfg(cfg_file: str, opts: Optional[Sequence[str]] = None, workspace: Optional[str] = None) -> SimpleNamespace:
    cfg = copy.deepcopy(DEFAULTS)
    ws = workspace or os.environ.get("workspace", os.path.join(os.getcwd(), "workspace"))
    cfg.update(workspace=ws, trained_model_dir=os.path.join(ws, "trained_model"), record_dir=os.path.join(ws, "record"),
               result_dir=os.path.join(ws, "result"))
    cur = _read(cfg_file)
    if "parent_cfg" in cur:  # one level, as the reference
        _merge(cfg, _read(cur["parent_cfg"]))
    _merge(cfg, cur)
    opts = list(opts or [])
    for key, val in zip(opts[0::2], opts[1::2]):
        try:
            val = ast.literal_eval(val)
        except (ValueError, SyntaxError):
            pass
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = val
    for sub in ("trained_model_dir", "record_dir"):
        cfg[sub] = os.path.join(cfg[sub], cfg["task"], cfg["exp_name"])
    cfg["result_dir"] = os.path.join(cfg["result_dir"], cfg["task"], cfg["exp_name"], cfg["save_tag"])
    for k in [k for k in cfg if k.endswith("_module")]:
        cfg[k.replace("_module", "_path")] = cfg[k].replace(".", "/") + ".py"
    return _namespace(cfg)
