"""Import shim: the product package lives in the directory `cmblensing.jl_amd/` (a dot is not a legal
Python package name), so `import cmblensing_jl_amd` loads that directory as a package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cmblensing.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "cmblensing_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cmblensing_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
