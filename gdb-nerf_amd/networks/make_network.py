"""Plugin factory, same contract as the reference's networks/make_network.py:5-9: load the module at
`cfg.network_path` under the name `cfg.network_module` and return its `Network(cfg)`."""
import importlib.util
import os
import sys


def load_source(module_name: str, path: str):
    """importlib equivalent of the (removed in 3.12) imp.load_source the reference uses.  A path that is
    not found as given is also tried relative to this package, so the reference's relative
    `networks/gdb_nerf/network.py` resolves to the MI355X implementation."""
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path)
    pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.abspath(path).startswith(pkg_root + os.sep):  # inside this package: import normally so relative imports work
        rel = os.path.relpath(os.path.abspath(path), pkg_root)[:-3].replace(os.sep, ".")
        return importlib.import_module(__name__.rsplit(".", 2)[0] + "." + rel)
    spec = importlib.util.spec_from_file_location(module_name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[module_name] = mod
    spec.loader.exec_module(mod)
    return mod


def make_network(cfg):
    return load_source(cfg.network_module, cfg.network_path).Network(cfg)
