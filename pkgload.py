"""Imports the package directory `lilac-benchmarks_amd/` (not a valid Python identifier) under
the module name `lilac_benchmarks_amd`.  Used by bench.py, __graft_entry__.py and tests/."""
import importlib.util
import os
import sys

_NAME = "lilac_benchmarks_amd"
_ROOT = os.path.dirname(os.path.abspath(__file__))


def load_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = os.path.join(_ROOT, "lilac-benchmarks_amd")
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
