"""Import alias: the package lives in the directory ``neural-lam-dev_amd/`` (a
name Python cannot import directly); ``import neural_lam_amd`` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neural-lam-dev_amd")
_spec = importlib.util.spec_from_file_location(
    "neural_lam_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["neural_lam_amd"] = _mod
_spec.loader.exec_module(_mod)
