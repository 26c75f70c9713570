"""Import shim: ``import pgas_amd`` loads the package directory
``bayesian-inference-with-explicit-and-implicit-prior-knowledge_amd/`` (not a valid identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bayesian-inference-with-explicit-and-implicit-prior-knowledge_amd")
_spec = importlib.util.spec_from_file_location("pgas_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pgas_amd"] = _mod
_spec.loader.exec_module(_mod)
