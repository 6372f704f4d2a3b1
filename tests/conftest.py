import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """the package directory is `chatllm.cpp_amd` (with a dot): import it under the name chatllm_cpp_amd"""
    if "chatllm_cpp_amd" in sys.modules:
        return sys.modules["chatllm_cpp_amd"]
    pkg_dir = os.path.join(ROOT, "chatllm.cpp_amd")
    spec = importlib.util.spec_from_file_location("chatllm_cpp_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["chatllm_cpp_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu(pkg):
    """GPU tests fail loudly (not skip) when the extension or the device is missing"""
    pkg.lib.get()
    pkg.lib.require_gpu()
    return pkg


import contextlib


@contextlib.contextmanager
def prefill_mode(pkg, mode):
    """switch how > 32-column mat-muls and the prompt's attention block are computed (1: the reference's order, the default; 0: the fast tolerance-tier kernels)"""
    L = pkg.lib.get()
    old = L.cllm_get_prefill_mode()
    assert L.cllm_set_prefill_mode(mode) == 0
    try:
        yield
    finally:
        L.cllm_set_prefill_mode(old)
