"""Import shim: the package directory is named `pq-vector_amd/` (not a legal Python
identifier), so `import pq_vector_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pq-vector_amd")
_spec = importlib.util.spec_from_file_location(
    "pq_vector_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pq_vector_amd"] = _mod
_spec.loader.exec_module(_mod)
