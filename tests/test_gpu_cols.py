"""The column-stream kernel (smk_cols.hip, option kernel = 3) against the CPU checker and the gather kernel: every
principal axis and marching direction, both voxel types, classification modes, ragged sizes, chunk and ring knobs.

Every sample's source colour is computed by the gather kernel's operations in its order; what differs is the
association of the front-to-back blend (a ray's samples are composited per column job and the partial composites
merged in order), so the bound against the gather kernel is a few ulp per segment -- 2e-5 -- not bit identity; against
the CPU checker it is the project's stated 1e-4."""
import numpy as np
import pytest

from _scenes import POSES, make_scene, push_scene

pytestmark = pytest.mark.gpu
TOL = 1e-4
TOL_G = 2e-5


@pytest.fixture(scope="module")
def R(gpu_renderer_factory):
    r = gpu_renderer_factory()
    yield r
    r.close()


def _both(R, sc, grid=(1, 1, 1), upload=True, **opts):
    push_scene(R, sc, grid, upload=upload)
    R.set_option("kernel", 1)
    a = R.render()
    for k, v in opts.items():
        R.set_option(k, v)
    R.set_option("kernel", 3)           # forced: raises if the column-stream kernel does not apply
    R.set_option("cols_counts", 1)
    b = R.render()
    assert R.last_frame_info()[0] == 4
    assert R.stat("slab_failures") == 0
    for k in opts:
        R.set_option(k, 0)
    R.set_option("kernel", 0)
    return a, b


@pytest.mark.parametrize("pose", sorted(POSES) + ["id", "rot", "back", "side"])
@pytest.mark.parametrize("f32", [False, True])
def test_every_axis_and_direction(R, pose, f32):
    sc = make_scene("cfg4", n=32, size=72, steps=80, pose=pose, f32=f32, shade=1)
    ref = sc.render()
    a, b = _both(R, sc)
    assert ref[..., 3].max() > 0.05
    assert np.abs(a - b).max() <= TOL_G, "column-stream and gather kernels differ: %g" % np.abs(a - b).max()
    assert np.abs(b - ref).max() <= TOL


@pytest.mark.parametrize("kind,shade,f32", [("cfg2", 0, True), ("cfg3", 1, True), ("cfg4", 0, True), ("cfg2", 1, False),
                                             ("tf3d", 1, True), ("tf3d_panes", 1, True), ("tf3d_panes", 1, False), ("cfg1", 0, False)])
def test_modes(R, kind, shade, f32):
    sc = make_scene(kind, n=32, size=64, steps=64, pose="diag", f32=f32, shade=shade)
    ref = sc.render()
    a, b = _both(R, sc)
    assert np.abs(a - b).max() <= TOL_G
    assert np.abs(b - ref).max() <= TOL


@pytest.mark.parametrize("pose", ["z-", "y+", "x-"])
def test_ragged_volume_and_window(R, pose):
    sc = make_scene("cfg2", dims=(40, 24, 18), shade=1, pose=pose, f32=True)
    sc.width, sc.height, sc.steps = 93, 41, 70
    ref = sc.render()
    a, b = _both(R, sc)
    assert np.abs(a - b).max() <= TOL_G and np.abs(b - ref).max() <= TOL


@pytest.mark.parametrize("chunk,ns", [(4, 3), (7, 4), (16, 0), (33, 6), (256, 0)])
def test_chunks_and_ring_sizes(R, chunk, ns):
    """short chunks put many job seams (and segments) on every ray; a three-slot ring is the protocol's minimum"""
    sc = make_scene("cfg3", n=32, size=96, steps=120, pose="rot", f32=True, shade=1)
    ref = sc.render()
    a, b = _both(R, sc, cols_chunk=chunk, cols_ns=ns)
    assert np.abs(a - b).max() <= TOL_G and np.abs(b - ref).max() <= TOL


def test_zoom_changes_rebuild_the_layout_and_stay_right(R):
    """the column size follows the pixel footprint of a cell: a viewport four times larger needs smaller columns"""
    sc = make_scene("cfg3", n=32, size=40, steps=64, pose="rot", f32=True, shade=1)
    a, b = _both(R, sc)
    assert np.abs(a - b).max() <= TOL_G
    n0 = R.stat("cols_builds")
    sc.width = sc.height = 200
    a, b = _both(R, sc, upload=False)
    assert np.abs(a - b).max() <= TOL_G
    assert R.stat("cols_builds") > n0
    assert np.abs(b - sc.render()).max() <= TOL


def test_max_blend(R):
    sc = make_scene("cfg3", n=32, size=64, steps=64, pose="side", f32=True, shade=1)
    sc.blend = 2
    push_scene(R, sc)
    R.set_blend(2)
    try:
        R.set_option("kernel", 1)
        a = R.render()
        R.set_option("kernel", 3)
        b = R.render()
        assert R.last_frame_info()[0] == 4
        assert np.array_equal(a, b)      # a maximum does not care about association
    finally:
        R.set_blend(0)
        R.set_option("kernel", 0)


def test_sample_count_equals_the_in_volume_samples(R):
    """every in-volume sample is taken exactly once, whatever job it falls into: the kernel's own count against the
    CPU checker's count of the same frame"""
    sc = make_scene("cfg3", n=32, size=80, steps=90, pose="rot", f32=True, shade=1)
    import oracle
    ref = sc.render()
    want = oracle.inside_samples()      # (the checker marches every plane of every ray: no early termination)
    a, b = _both(R, sc, cols_chunk=8)
    assert np.abs(b - ref).max() <= TOL
    assert R.stat("cols_samples") == want
