import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import _pkgload  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "library_choice: a GPU test that runs once, with the library's own kernel choice")


def pytest_generate_tests(metafunc):
    """Every GPU test runs twice: once with the pooled megakernel forced, once with the per-tile one
    (the library itself picks by mesh size, which would leave one of them untested on small scenes)."""
    own = hasattr(metafunc.module, "megakernel")  # a module that picks kernels itself overrides the fixture
    once = metafunc.definition.get_closest_marker("library_choice") is not None
    if metafunc.definition.get_closest_marker("gpu") is not None and "megakernel" in metafunc.fixturenames and not own and not once:
        metafunc.parametrize("megakernel", ["pool", "tile"], indirect=True)


@pytest.fixture(autouse=True)
def megakernel(request):
    kind = getattr(request, "param", None)
    if kind is None:
        yield None
        return
    pt = _pkgload.load()
    pt.force_kernel(kind)  # PT_FLAG_POOL_KERNEL / PT_FLAG_TILE_KERNEL on every Renderer the test makes
    yield kind
    pt.force_kernel(None)


def _ensure_built():
    """The .so files are built by __graft_entry__.build() and travel with the tree; this only
    covers a fresh checkout on a machine with the toolchain."""
    lib = os.path.join(ROOT, "path-tracer_amd", "libptrt.so")
    ora = os.path.join(ROOT, "oracle", "libpt_oracle.so")
    if not os.path.exists(ora):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "path-tracer_amd", "csrc")])


@pytest.fixture(scope="session")
def pt():
    _ensure_built()
    return _pkgload.load()


@pytest.fixture(scope="session")
def scenes(pt):
    import importlib
    return importlib.import_module("path_tracer_amd.scenes")


@pytest.fixture(scope="session")
def po(pt):
    from oracle import pt_oracle
    pt_oracle.lib()
    return pt_oracle
