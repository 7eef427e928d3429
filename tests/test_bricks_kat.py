"""The empty-space rule of smk_bricks.hip, restated in oracle/bricks.py, checked on the CPU: known answers, and the
property everything rests on -- a sample whose cell lies in a brick with a clear flag ends at a clear occupancy bit,
i.e. its table lookup returns alpha == 0 exactly."""
import numpy as np
import pytest

from _scenes import make_scene
import bricks as B            # oracle/bricks.py (tests put oracle/ on the path)
import oracle as O


def _vg(sc):
    vol = sc.data
    if vol.dtype == np.uint8:
        v = vol[..., 0].astype(np.float32) * np.float32(1.0 / 255.0)
        g = vol[..., 1].astype(np.float32) * np.float32(1.0 / 255.0)
    else:
        v, g = vol[..., 0].astype(np.float32), vol[..., 1].astype(np.float32)
    return np.ascontiguousarray(v), np.ascontiguousarray(g)


def test_uniform_volume_follows_its_texel():
    """a constant volume: every brick has the range of one value; the flag is the occupancy of that texel's surroundings"""
    occ = np.zeros((256, 256), bool)
    occ[100:104, 50:54] = True
    for val, expect in ((52.0 / 256, 1), (10.0 / 256, 0), (58.0 / 256, 0), (55.4 / 256, 1)):   # 55.4 -> base 54: within one texel of 53
        v = np.full((20, 17, 9), val, np.float32)
        g = np.full((20, 17, 9), 102.0 / 256, np.float32)
        f = B.brick_flags(v, g, occ)
        assert f.shape == (3, 3, 2)      # (D - 1) // 8 + 1 bricks per axis
        assert (f == expect).all(), (val, f.ravel())


def test_brick_sees_the_voxels_its_cells_touch():
    """a single bright voxel at index 8 belongs to the cells of brick 0 (upper corner) and brick 1 (lower corner)"""
    occ = np.zeros((256, 256), bool)
    occ[:, 200:] = True
    v = np.zeros((25, 9, 9), np.float32)
    g = np.zeros_like(v)
    v[8, 4, 4] = 0.9
    f = B.brick_flags(v, g, occ)
    assert f[:, 0, 0].tolist() == [1, 1, 0, 0][:f.shape[0]]


@pytest.mark.parametrize("kind,f32", [("cfg3", True), ("cfg3", False), ("cfg2", False)])
def test_a_clear_flag_means_an_exactly_transparent_sample(kind, f32):
    """random sample positions, trilinear (v, g) as the kernels interpolate them (fma lerps, x then y then z), the
    occupancy bit of the base texel: whenever it is set, the sample's brick must be flagged"""
    sc = make_scene(kind, n=40, f32=f32, shade=0)
    v, g = _vg(sc)
    tf = sc.tf_vg
    occ = B.occupancy(tf[..., 3])
    flags = B.brick_flags(v, g, occ)
    assert 0 < flags.mean() < 1 or kind == "cfg2", "vacuous: every brick or none flagged"
    rng = np.random.default_rng(5)
    D = np.array(v.shape)
    p = rng.uniform(0, 1, (200000, 3)) * (D - 1)
    i0 = np.minimum(p.astype(np.int64), D - 2)
    f = (p - i0).astype(np.float32)

    def lerp(a, b, t):
        return (t.astype(np.float64) * (b.astype(np.float64) - a.astype(np.float64)).astype(np.float32).astype(np.float64)
                + a.astype(np.float64)).astype(np.float32)      # fma(t, b - a, a): b - a rounded, then one rounding

    def tri(ch):
        c = lambda dz, dy, dx: ch[i0[:, 0] + dz, i0[:, 1] + dy, i0[:, 2] + dx]
        fx, fy, fz = f[:, 2], f[:, 1], f[:, 0]
        return lerp(lerp(lerp(c(0, 0, 0), c(0, 0, 1), fx), lerp(c(0, 1, 0), c(0, 1, 1), fx), fy),
                    lerp(lerp(c(1, 0, 0), c(1, 0, 1), fx), lerp(c(1, 1, 0), c(1, 1, 1), fx), fy), fz)

    if sc.data.dtype == np.uint8:       # byte voxels: interpolate the bytes, scale afterwards (as the kernels do)
        vb, gb = sc.data[..., 0].astype(np.float32), sc.data[..., 1].astype(np.float32)
        sv_, sg_ = tri(vb) * np.float32(1 / 255.0), tri(gb) * np.float32(1 / 255.0)
    else:
        sv_, sg_ = tri(v), tri(g)
    s0 = B.base_texel(sv_, occ.shape[1])
    t0 = B.base_texel(sg_, occ.shape[0])
    visible = occ[t0, s0]
    brick = flags[i0[:, 0] // B.BRICK, i0[:, 1] // B.BRICK, i0[:, 2] // B.BRICK]
    assert visible.any()
    assert not (visible & (brick == 0)).any(), "a possibly visible sample in a brick flagged empty"
