import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """The package directory name contains '.' and '-', so import it by path as `eagle_amd`."""
    if "eagle_amd" in sys.modules:
        return sys.modules["eagle_amd"]
    p = os.path.join(ROOT, "eagle-in-llama.cpp_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("eagle_amd", p, submodule_search_locations=[os.path.dirname(p)])
    m = importlib.util.module_from_spec(spec)
    sys.modules["eagle_amd"] = m
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def ea():
    return load_package()


@pytest.fixture(scope="session")
def gpu(ea):
    """The MI355X backend through the plugin's C ABI; fails loudly (no CPU fallback) when absent."""
    return ea.Backend.mi355x(0)


def have_ref():
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libggml-ref.so"))


@pytest.fixture(scope="session")
def ref_cpu(ea):
    if not have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference in the build container)")
    import refapi
    return refapi.reference_cpu(ea)


@pytest.fixture(scope="session")
def ref_scalar(ea):
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    import refapi
    return refapi.reference_cpu(ea, scalar=True)
