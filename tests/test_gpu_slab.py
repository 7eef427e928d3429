"""The slice-ring (LDS-staged) kernel against the CPU checker AND the generic gather kernel:
every principal axis and marching direction, both voxel types, sharded regions, ragged sizes.
Same tolerance as the rest of the GPU parity suite (max-abs 1e-4); against the gather kernel the
frames must be bit-identical (same fma chains, same sample order)."""
import numpy as np
import pytest

from _scenes import POSES, make_scene, push_scene

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def R(gpu_renderer_factory):
    r = gpu_renderer_factory()
    yield r
    r.close()


def _both(R, sc, grid=(1, 1, 1), depth=False, upload=True):
    push_scene(R, sc, grid, upload=upload)
    R.set_option("kernel", 1)
    a = R.render(depth=depth)
    R.set_option("kernel", 2)           # forced: raises if the slab kernel does not apply
    b = R.render(depth=depth)
    assert R.last_frame_info()[0] == 2
    R.set_option("kernel", 0)
    return a, b


@pytest.mark.parametrize("pose", sorted(POSES) + ["id", "rot", "back", "side"])
@pytest.mark.parametrize("f32", [False, True])
def test_every_axis_and_direction(R, pose, f32):
    sc = make_scene("cfg4", n=32, size=72, steps=80, pose=pose, f32=f32, shade=1)
    ref = sc.render()
    a, b = _both(R, sc)
    assert ref[..., 3].max() > 0.05
    assert np.array_equal(a, b), "slab and gather kernels differ: %g" % np.abs(a - b).max()
    assert np.abs(b - ref).max() <= TOL


@pytest.mark.parametrize("kind,shade", [("cfg2", 0), ("cfg3", 1), ("cfg3", 2), ("cfg4", 0)])
def test_modes(R, kind, shade):
    sc = make_scene(kind, n=32, size=64, steps=64, pose="diag", f32=True, shade=shade)
    ref = sc.render()
    a, b = _both(R, sc)
    assert np.array_equal(a, b)
    assert np.abs(b - ref).max() <= TOL


def test_depth_request_uses_the_gather_kernel(R):
    """First-hit depth is a register the slice-ring kernel does not spend: a frame that asks for
    it runs on the gather kernel in auto mode (same RGBA, bit for bit) and is refused when the
    slice-ring kernel is forced."""
    sc = make_scene("cfg3", n=32, size=64, steps=64, pose="diag", f32=True, shade=1)
    (ref, rd) = sc.render(depth=True)
    push_scene(R, sc)
    R.set_option("kernel", 0)
    rgba, dep = R.render(depth=True)
    assert R.last_frame_info()[0] == 1
    plain = R.render()
    assert R.last_frame_info()[0] == 2
    assert np.array_equal(rgba, plain)
    fin = np.isfinite(rd)
    assert np.array_equal(fin, np.isfinite(dep)) and np.abs(rd[fin] - dep[fin]).max() <= 1e-4
    R.set_option("kernel", 2)
    with pytest.raises(Exception, match="depth"):
        R.render(depth=True)
    R.set_option("kernel", 0)


@pytest.mark.parametrize("pose", ["z-", "y+", "x-"])
def test_ragged_volume_and_window(R, pose):
    sc = make_scene("cfg2", dims=(40, 24, 18), shade=1, pose=pose, f32=True)
    sc.width, sc.height, sc.steps = 93, 41, 70
    ref = sc.render()
    a, b = _both(R, sc)
    assert np.array_equal(a, b) and np.abs(b - ref).max() <= TOL


def test_u8_odd_extent_falls_back_in_auto_mode(R):
    sc = make_scene("cfg2", dims=(39, 24, 18), shade=1, pose="z+")
    push_scene(R, sc)
    R.set_option("kernel", 0)
    img = R.render()
    assert R.last_frame_info()[0] == 1          # 8-byte voxels need an even U extent for the DMA
    assert np.abs(img - sc.render()).max() <= TOL


def test_sample_rate_mode_and_many_planes(R):
    sc = make_scene("cfg3", n=32, size=64, pose="y-", f32=True, shade=1)
    sc.steps, sc.sample_rate = 0, 3.3          # more planes than slices: several samples per slice
    ref = sc.render()
    a, b = _both(R, sc)
    assert np.array_equal(a, b) and np.abs(b - ref).max() <= TOL
    sc.steps, sc.sample_rate = 9, 0.0           # fewer planes than slices: slices without samples
    ref = sc.render()
    a, b = _both(R, sc)
    assert np.array_equal(a, b) and np.abs(b - ref).max() <= TOL


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_regions(gpu_renderer_factory, smk, world):
    from simian_spacemonkey_amd import sortlast
    sc = make_scene("cfg4", n=32, size=64, steps=64, pose="x+", f32=True, shade=1)
    for rank in (0, world - 1):
        r = gpu_renderer_factory()
        try:
            r.set_shard(rank, world)
            a, b = _both(r, sc)
            sc.region = sortlast.shard_region(sc.dims, rank, world)
            assert np.array_equal(a, b)
            assert np.abs(b - sc.render()).max() <= TOL
        finally:
            sc.region = ((0, 0, 0), sc.dims)
            r.close()


def test_forced_slab_reports_why_it_cannot_run(R, smk):
    sc = make_scene("tf3d", f32=True)
    push_scene(R, sc)
    R.set_option("kernel", 2)
    with pytest.raises(smk.SmkError, match="gather-only|not applicable"):
        R.render()
    R.set_option("kernel", 0)
