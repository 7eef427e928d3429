"""pytest plumbing: the `gpu` marker, import paths for the product package (its directory name
has a hyphen) and for the CPU checker under oracle/ (test infrastructure only)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_package():
    """import simian-spacemonkey_amd/ under the importable name simian_spacemonkey_amd"""
    name = "simian_spacemonkey_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "simian-spacemonkey_amd")
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once, the way
    __graft_entry__.build() does (hipcc cross-compiles without a GPU), instead of failing every test
    that needs the library, the checker or the host-side drivers."""
    need = [os.path.join(ROOT, "simian-spacemonkey_amd", "csrc", "libsmk_hip.so"),
            os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "tests", "host", "adapter_main"),
            os.path.join(ROOT, "tests", "host", "files_main"),
            os.path.join(ROOT, "tests", "host", "tf_main")]
    if all(os.path.exists(p) for p in need):
        return
    spec = importlib.util.spec_from_file_location("smk_graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()


@pytest.fixture(scope="session")
def smk():
    return load_package()


@pytest.fixture(scope="session")
def O():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_renderer_factory(smk):
    """GPU tests must run the HIP library: no fallback, fail loudly if it is missing."""
    import torch  # noqa: F401  (device bring-up is torch's job on the box; plumbing only)
    lib = smk.library_path()
    assert os.path.exists(lib), "libsmk_hip.so missing: run __graft_entry__.build() first"

    def make(device=0):
        return smk.Renderer(device)
    return make
