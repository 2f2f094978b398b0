"""pytest configuration: registers the ``gpu`` marker and makes the repo root importable."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest


@pytest.fixture
def natural_grid_ordering():
    """Tests of the WAVE form on 2-D grid stencils: keep the operators of the default context in their natural ordering (context
    option patch = 0; by default such an operator is stored in the grid-patch ordering and takes the patch form, which has its own
    tests)."""
    import expv_mi_loader
    eu = expv_mi_loader.load()
    ctx = eu.default_context()
    ctx.set_option("patch", 0)
    eu.clear_operator_cache()
    yield
    ctx.set_option("patch", 1)
    eu.clear_operator_cache()
