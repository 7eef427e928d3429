"""Brick flags (smk_bricks.hip): layers of cells in which no sample can be visible under the current table are
neither streamed nor sampled by the slice-ring kernel.  The skipped samples are exactly transparent, so the
frame must not change by a single bit -- against the same kernel with the flags off (option "bricks" 0),
against the gather kernel, and (1e-4) against the CPU checker, which knows nothing of bricks."""
import numpy as np
import pytest

from _scenes import POSES, make_scene, push_scene

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def R(gpu_renderer_factory):
    r = gpu_renderer_factory()
    yield r
    r.set_option("bricks", 1)
    r.close()


def _three(R, sc, grid=(1, 1, 1)):
    """frames of: gather kernel WITHOUT the flags (the reference the others must equal), slice ring without flags,
    slice ring with flags (+ streamed fractions); the gather kernel with the flags is checked on the way"""
    R.set_option("bricks", 0)
    push_scene(R, sc, grid)
    R.set_option("kernel", 1)
    g = R.render()
    R.set_option("kernel", 2)
    s0 = R.render()
    assert R.last_frame_info()[0] == 2
    f0 = R.stat("slab_streamed_fraction")
    R.set_option("bricks", 1)
    s1 = R.render()
    assert R.last_frame_info()[0] == 2
    f1 = R.stat("slab_streamed_fraction")
    R.set_option("kernel", 1)
    g1 = R.render()
    assert np.array_equal(g1, g), "gather kernel: the flags changed the frame"
    R.set_option("kernel", 0)
    assert R.stat("slab_failures") == 0
    return g, s0, s1, f0, f1


@pytest.mark.parametrize("pose", sorted(POSES) + ["rot"])
@pytest.mark.parametrize("f32", [False, True])
def test_sparse_table_every_axis(R, pose, f32):
    """the LevWidget table leaves most of a 96^3 volume transparent: slices are skipped, the frame is not touched"""
    sc = make_scene("cfg3", n=96, size=160, steps=200, pose=pose, f32=f32, shade=1)
    g, s0, s1, f0, f1 = _three(R, sc)
    assert g[..., 3].max() > 0.05
    assert np.array_equal(s0, g) and np.array_equal(s1, g)
    assert f1 <= f0


def test_flags_skip_something_and_the_checker_agrees(R):
    sc = make_scene("cfg3", n=96, size=128, steps=160, pose="rot", f32=True, shade=1)
    ref = sc.render()
    g, s0, s1, f0, f1 = _three(R, sc)
    assert np.array_equal(s1, g)
    assert np.abs(s1 - ref).max() <= TOL
    assert f1 < 0.95 * f0, "no slice was skipped (%g vs %g)" % (f1, f0)


def test_flags_follow_the_table(R):
    """a new table (and a new correction rate) brings new flags: first a table that hides everything but a band,
    then an opaque one, then the first again -- each frame equal to the gather kernel's"""
    sc = make_scene("cfg3", n=64, size=96, steps=128, pose="diag", f32=True, shade=1)
    R.set_option("bricks", 1)
    push_scene(R, sc)
    band = np.zeros((256, 256, 4), np.uint8)
    band[:, 100:120] = (200, 120, 40, 90)
    opaque = np.full((256, 256, 4), 255, np.uint8)
    for tf, steps in ((band, 128), (opaque, 128), (band, 96), (sc.tf_vg, 77)):
        R.set_tf2d(tf, None)
        R.set_sampling(0.0, steps, 1.0, 1)
        R.set_option("kernel", 1)
        g = R.render()
        R.set_option("kernel", 2)
        s = R.render()
        assert R.last_frame_info()[0] == 2
        assert np.array_equal(s, g)
    R.set_option("kernel", 0)


@pytest.mark.parametrize("kind,f32", [("tf3d_panes", True), ("tf3d_panes", False), ("cfg4", True), ("cfg2", False)])
def test_other_tables(R, kind, f32):
    """dense 3-D table (flags from its occupancy folded over the sheets), (v,g) x third axis, the default ramp"""
    sc = make_scene(kind, n=64, size=96, steps=128, pose="side", f32=f32, shade=1)
    ref = sc.render()
    g, s0, s1, f0, f1 = _three(R, sc)
    assert np.array_equal(s0, g) and np.array_equal(s1, g)
    assert np.abs(s1 - ref).max() <= TOL


@pytest.mark.parametrize("rank", [0, 3, 5])
def test_shard_boxes(gpu_renderer_factory, rank):
    """a shard stores its region + halo at an offset: the flags are indexed in stored-box coordinates"""
    sc = make_scene("cfg3", n=64, size=96, steps=128, pose="rot", f32=True, shade=1)
    r = gpu_renderer_factory()
    try:
        r.set_shard(rank, 8)
        g, s0, s1, f0, f1 = _three(r, sc)
        assert np.array_equal(s1, g) and np.array_equal(s0, g)
    finally:
        r.close()


def test_ragged_and_thin(R):
    for dims, pose in (((40, 24, 18), "z-"), ((17, 70, 9), "y+"), ((9, 9, 130), "x+"), ((70, 17, 33), "diag")):
        sc = make_scene("cfg3", dims=dims, shade=1, pose=pose, f32=True)
        sc.width, sc.height, sc.steps = 93, 61, 150
        g, s0, s1, f0, f1 = _three(R, sc)
        assert np.array_equal(s0, g) and np.array_equal(s1, g), (dims, pose)


def test_depth_and_clip_plane_frames_on_both_kernels(R):
    """first-hit depth requested, with a free clip plane (since round 3 the slice-ring kernel renders such frames too):
    flags on == flags off, depth included, on the gather kernel and on the slice-ring kernel, and the two kernels agree
    bit for bit"""
    sc = make_scene("cfg3", n=64, size=96, steps=128, pose="rot", f32=True, shade=1)
    n = np.array([0.35, -0.2, -0.9])
    n /= np.linalg.norm(n)
    mv = np.array(sc.mv(), np.float64).reshape(4, 4).T   # column-major -> rows
    centre = mv @ np.array([float(sc.fsize[0]) / 2, float(sc.fsize[1]) / 2, float(sc.fsize[2]) / 2, 1.0])
    sc.clip_plane = (n[0], n[1], n[2], -float(n @ centre[:3]) + 0.03)   # through (almost) the volume's middle
    out = {}
    try:
        for kern in (1, 2):
            for b in (0, 1):
                R.set_option("bricks", b)
                push_scene(R, sc)
                R.set_option("kernel", kern)
                out[kern, b] = R.render(depth=True)
                assert R.last_frame_info()[0] == kern
            assert np.array_equal(out[kern, 0][0], out[kern, 1][0])
            assert np.array_equal(out[kern, 0][1], out[kern, 1][1])
        assert np.array_equal(out[1, 1][0], out[2, 1][0])
        assert np.array_equal(out[1, 1][1], out[2, 1][1])
        assert out[1, 0][0][..., 3].max() > 0.05
    finally:
        sc.clip_plane = None
        push_scene(R, sc)
        R.set_option("bricks", 1)
        R.set_option("kernel", 0)


@pytest.mark.parametrize("f32", [False, True])
def test_shadow_frames(R, f32):
    """half-angle slicing: the eye pass and the light pass skip the same transparent samples -- frame and light buffer
    bit for bit with the flags on and off"""
    sc = make_scene("cfg3", n=64, size=96, steps=128, f32=f32, shade=1)
    sc.light_pos = (3, 4, -3)
    sc.shadow = (128, 0.5)
    out = {}
    for b in (0, 1):
        R.set_option("bricks", b)
        push_scene(R, sc)
        frame = R.render()
        assert R.last_frame_info()[0] in (1, 2)
        out[b] = (frame, R.light_buffer())
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert out[0][0][..., 3].max() > 0.05 and out[0][1][..., 3].max() > 0.05
    sc.shadow = None
    push_scene(R, sc)
    R.set_option("bricks", 1)


@pytest.mark.parametrize("weights", [(.2, .1, 0, 0), (.05, .02, 0, 0), (.6, .3, 0, 0)])
@pytest.mark.parametrize("kind", ["cfg3", "tf3d_panes"])
def test_perturbed_fetch(R, kind, weights):
    """a noise-displaced fetch lands within 0.5 (|w0| + |w1|) N voxels of where it started: the flags, spread over the
    bricks within that reach, stop a sample before its noise lookups (reach of one or two bricks; beyond that the
    sample's own displaced brick is still tested) -- frames equal with the flags on and off, and the checker agrees"""
    sc = make_scene(kind, n=64, size=96, steps=128, pose="rot", f32=True, shade=1, pert=True)
    sc.pert_w = weights
    ref = sc.render()
    out = {}
    for b in (0, 1):
        R.set_option("bricks", b)
        push_scene(R, sc)
        R.set_option("kernel", 0)
        out[b] = R.render()
        assert R.last_frame_info()[0] == 1
    assert np.array_equal(out[0], out[1])
    assert np.abs(out[1] - ref).max() <= TOL
    assert out[1][..., 3].max() > 0.05
    sc.noise = None
    push_scene(R, sc)
    R.set_option("bricks", 1)


@pytest.mark.parametrize("kind,f32,dims", [("cfg3", True, None), ("cfg3", False, None), ("cfg2", False, None), ("cfg4", True, None),
                                           ("tf3d_panes", True, None), ("cfg3", True, (41, 23, 70))])
def test_flags_equal_the_numpy_restatement(R, kind, f32, dims):
    """the flags themselves, byte for byte, against oracle/bricks.py (value ranges over the voxels a brick's cells touch,
    the occupancy bitmap of the EFFECTIVE table, the summed-area range test widened by one texel)"""
    import bricks as B      # oracle/bricks.py
    sc = make_scene(kind, n=48, size=64, steps=96, pose="rot", f32=f32, shade=1, dims=dims)
    R.set_option("bricks", 1)
    push_scene(R, sc)
    got, in_use = R.brick_flags()
    vol = sc.data
    if vol.dtype == np.uint8:
        v = vol[..., 0].astype(np.float32) * np.float32(1.0 / 255.0)
        g = vol[..., 1].astype(np.float32) * np.float32(1.0 / 255.0)
    else:
        v, g = vol[..., 0].astype(np.float32), vol[..., 1].astype(np.float32)
    if sc.tf_mode == 2:
        occ = B.fold_occupancy(sc.tf3d[..., 3])
    else:
        eff, _ = R.tf2d_effective(sc.tf_vg.shape[1], sc.tf_vg.shape[0])
        occ = B.occupancy(eff[..., 3])
    want = B.brick_flags(np.ascontiguousarray(v), np.ascontiguousarray(g), occ)
    assert got.shape == want.shape
    assert np.array_equal(got, want), "%d of %d flags differ" % ((got != want).sum(), got.size)
    assert in_use == (want.mean() <= 0.9) or in_use    # (dropped only once the count came back above 90 %)


def test_a_table_edited_every_frame_without_synchronising(R):
    """the interactive case: a new table AND a new correction rate every frame, frames enqueued without waiting (the raw
    table, its effective versions, their bitmaps and brick flags all rotate behind stream events); every eighth frame
    is compared with the gather kernel's, flags off, after the fact"""
    import torch
    sc = make_scene("cfg3", n=64, size=96, steps=128, pose="rot", f32=True, shade=1)
    R.set_option("bricks", 1)
    push_scene(R, sc)
    R.set_option("kernel", 2)
    rng = np.random.default_rng(11)
    frames = [torch.zeros((96 * 96, 4), dtype=torch.float32, device="cuda") for _ in range(5)]
    kept = []
    for f in range(40):
        tf = np.zeros((256, 256, 4), np.uint8)
        lo = int(rng.integers(20, 200))
        tf[int(rng.integers(0, 100)):int(rng.integers(120, 256)), lo:lo + int(rng.integers(4, 50))] = (
            int(rng.integers(30, 255)), int(rng.integers(30, 255)), 60, int(rng.integers(1, 255)))
        steps = 100 + (f * 7) % 60
        R.set_tf2d(tf, None)
        R.set_sampling(0.0, steps, 1.0, 1)
        keep = f % 8 == 7
        R.render_device(frames[len(kept) if keep else 4].data_ptr())
        if keep:
            kept.append((tf, steps))
    torch.cuda.synchronize()
    assert R.stat("slab_failures") == 0
    R.set_option("kernel", 1)
    R.set_option("bricks", 0)
    for k, (tf, steps) in enumerate(kept):
        R.set_tf2d(tf, None)
        R.set_sampling(0.0, steps, 1.0, 1)
        ref = R.render().reshape(-1, 4)
        got = frames[k].cpu().numpy()
        assert np.array_equal(got, ref), "frame %d differs by %g" % (k, np.abs(got - ref).max())
        assert ref[:, 3].max() > 0
    R.set_option("kernel", 0)
    R.set_option("bricks", 1)
    push_scene(R, sc)


def test_a_nan_voxel_keeps_its_brick_flagged(R):
    """fminf / fmaxf drop a NaN operand: a brick that mixes one NaN voxel with values the table leaves transparent would
    get a finite, 'empty' range, while a sample interpolated from the NaN corner classifies at base texel 0 -- which this
    table makes opaque.  A brick that saw a non-finite voxel is flagged whatever the table: frames with and without the
    flags stay bit-identical, on both kernels."""
    import oracle as O
    sc = make_scene("cfg3", n=32, size=64, steps=64, pose="rot", f32=True, shade=0)
    data = np.full(sc.data.shape, 0.9, np.float32)        # everything up here, where the table below is transparent
    data[12, 13, 14, 0] = np.nan
    data[20, 5, 7, 1] = np.inf
    sc = O.Scene(data, grad=sc.grad)
    sc.tf_mode = 1
    tf = np.zeros((256, 256, 4), np.uint8)
    tf[:4, :4] = (255, 200, 100, 255)                     # texel (0, 0) and its neighbours: opaque
    sc.tf_vg = tf
    sc.width = sc.height = 64
    sc.steps = 64
    sc.xform = O.rotation((1, 1, 0), 30)
    g, s0, s1, f0, f1 = _three(R, sc)
    assert np.array_equal(np.nan_to_num(s0, nan=-1.0), np.nan_to_num(g, nan=-1.0))
    assert np.array_equal(np.nan_to_num(s1, nan=-1.0), np.nan_to_num(g, nan=-1.0))
    flags, in_use = R.brick_flags()                         # [z][y][x] of bricks
    assert in_use and flags[12 // 8, 13 // 8, 14 // 8] == 1 and flags[20 // 8, 5 // 8, 7 // 8] == 1
    assert flags.sum() < flags.size                        # (the finite rest is not flagged)
