"""The restated Perlin noise against (a) the committed probes and (b) the reference's own
genvol/perlin.c built into oracle/_ref/libperlin_ref.so (oracle/Makefile) -- the one piece of
the reference that compiles here.  (b) also proves the in-repo rand() clone equals glibc's."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_probes(O):
    p = np.load(os.path.join(ROOT, "tests", "golden", "perlin_probes.npy"))
    L = O.lib()
    L.orc_srand(1)
    L.orc_perlin_reset()
    L.orc_perlin_init()
    for x, y, z, a, b in p:
        assert L.orc_perlin3d(x, y, z, 2.0, 2.0, 10) == a
        assert L.orc_perlin3d_abs(x, y, z, 2.0, 2.0, 10) == b


def test_against_reference_build(O):
    so = os.path.join(ROOT, "oracle", "_ref", "libperlin_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    R = C.CDLL(so)
    libc = C.CDLL("libc.so.6")
    for f in (R.PerlinNoise3D, R.PerlinNoise3DABS):
        f.restype = C.c_double
        f.argtypes = [C.c_double] * 5 + [C.c_int]
    libc.srand(1)
    R.init()                       # genvol main's explicit init(); noise3 re-inits on first call
    L = O.lib()
    L.orc_srand(1)
    L.orc_perlin_reset()
    L.orc_perlin_init()
    p = np.load(os.path.join(ROOT, "tests", "golden", "perlin_probes.npy"))
    for x, y, z, a, b in p:
        assert R.PerlinNoise3D(x, y, z, 2.0, 2.0, 10) == a == L.orc_perlin3d(x, y, z, 2.0, 2.0, 10)
        assert R.PerlinNoise3DABS(x, y, z, 2.0, 2.0, 10) == b


def test_rand_clone_equals_glibc(O):
    libc = C.CDLL("libc.so.6")
    L = O.lib()
    for seed in (1, 2, 77):
        libc.srand(seed)
        L.orc_srand(seed)
        assert [libc.rand() for _ in range(500)] == [L.orc_rand() for _ in range(500)]
