"""Imports the package directory ``exponentialutilities.jl_amd`` (not a valid Python identifier)
under the module name ``exponentialutilities_jl_amd``."""
import importlib.util
import os
import sys

NAME = "exponentialutilities_jl_amd"
PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exponentialutilities.jl_amd")


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(NAME, None)
        raise
    return mod


def build(force=False):
    """Compile libexpv_mi.so (hipcc --offload-arch=gfx950) without importing the package."""
    spec = importlib.util.spec_from_file_location(NAME + "_build", os.path.join(PKG_DIR, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = mod.build(force=force)
    mod.build_callback_example()                 # tests/c_harness/libstencil_cb.so (compiled matrix-free operator: tests / bench only)
    return lib
