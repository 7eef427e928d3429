"""The C-ABI library loads and exports every symbol include/smk.h declares; without a GPU the
product fails loudly instead of falling back to anything (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "smk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(smk):
    so = smk.library_path()
    assert os.path.exists(so), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(so)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libsmk_hip.so does not export %s" % s
    assert sorted(smk.ABI_SYMBOLS) == syms, "binding.ABI_SYMBOLS out of sync with include/smk.h"


def test_product_never_links_the_checker(smk):
    """oracle/ is test infrastructure: the shipped library must not depend on it"""
    import subprocess
    out = subprocess.run(["ldd", smk.library_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "simian-spacemonkey_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".py", ".cpp")):
                for line in open(os.path.join(root, f)):
                    assert not re.search(r"#\s*include.*oracle|liboracle|^\s*(import|from)\s+oracle|smk_oracle", line), (f, line)


def test_no_cpu_fallback(smk):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(smk.SmkError, match="no HIP device"):
        smk.Renderer(0)
    lib = smk.load_library()
    assert b"no HIP device" in lib.smk_last_error(None)
