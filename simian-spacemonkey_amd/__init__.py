"""MI355X-native volume ray-marcher behind Simian's renderer slot.

The product is the C-ABI shared library csrc/libsmk_hip.so (include/smk.h); this package is the
thin ctypes front end used by the tests, bench.py and the multi-GPU driver.  The directory name
contains a hyphen, so load it with `importlib` (see tests/conftest.py::load_package) or through
__graft_entry__.build().
"""
from . import binding  # noqa: F401
from .binding import (SmkError, Renderer, Exchange, exchange_unique_id, build_library, library_path,  # noqa: F401
                      load_library, ABI_SYMBOLS)
