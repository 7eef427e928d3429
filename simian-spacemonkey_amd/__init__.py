"""MI355X-native volume ray-marcher behind Simian's renderer slot.

The product is the C-ABI shared library csrc/libsmk_hip.so (include/smk.h); this package is the
thin ctypes front end used by the tests, bench.py and the multi-GPU driver.  The directory name
contains a hyphen, so load it with `importlib` (see tests/conftest.py::load_package) or through
__graft_entry__.build().
"""
from .binding import (SmkError, Renderer, build_library, library_path, load_library,  # noqa: F401
                      ABI_SYMBOLS)
