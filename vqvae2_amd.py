"""Import shim: makes the package directory `vq-vae-2-pytorch_amd/` (hyphens are not a
valid Python identifier) importable as `vqvae2_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vq-vae-2-pytorch_amd")
_spec = importlib.util.spec_from_file_location(
    "vqvae2_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vqvae2_amd"] = _mod
_spec.loader.exec_module(_mod)
