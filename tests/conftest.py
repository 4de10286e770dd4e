import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def nerf_weights_of(fx):
    return {k[2:]: v for k, v in fx.items() if k.startswith("w.")}


FRAME_KEYS = ("src_images", "img_feat", "feat_volume", "depth_range", "vol_range",
              "src_exts", "src_ints", "tar_ext", "tar_int", "near_far")


def frame_of(fx):
    return {k: fx[k] for k in FRAME_KEYS}


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


@pytest.fixture(scope="session")
def golden():
    return load_golden
