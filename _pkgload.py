"""Imports the package directory `path-tracer_amd/` (not a valid identifier) as module
`path_tracer_amd`, and puts the repository root on sys.path for `oracle.*`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "path_tracer_amd" in sys.modules:
        return sys.modules["path_tracer_amd"]
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    pkg_dir = os.path.join(ROOT, "path-tracer_amd")
    spec = importlib.util.spec_from_file_location("path_tracer_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["path_tracer_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
