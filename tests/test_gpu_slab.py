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


@pytest.mark.parametrize("kind,shade,f32", [("cfg2", 0, True), ("cfg3", 1, True), ("cfg3", 2, True), ("cfg4", 0, True),
                                             ("tf3d", 1, True), ("tf3d", 0, False), ("tf3d", 2, False),
                                             ("tf3d_panes", 1, True), ("tf3d_panes", 1, False), ("tf3d_panes", 0, True)])
def test_modes(R, kind, shade, f32):
    """2-D table, 2-D x third axis, and the dense 3-D (v,g,h) table of BASELINE configs 4/5 on the slice-ring kernel"""
    sc = make_scene(kind, n=32, size=64, steps=64, pose="diag", f32=f32, shade=shade)
    ref = sc.render()
    a, b = _both(R, sc)
    assert np.array_equal(a, b)
    assert np.abs(b - ref).max() <= TOL


@pytest.mark.parametrize("kind,pose,f32,blend", [("cfg3", "diag", True, 0), ("cfg3", "rot", False, 0), ("cfg4", "side", True, 0),
                                                  ("cfg3", "back", True, 2), ("cfg2", "diag", True, 0)])
def test_first_hit_depth_on_the_slice_ring_kernel(R, kind, pose, f32, blend):
    """First-hit depth (round 3): the first sample that passes classification finds the accumulated alpha exactly 0 -- the
    slice-ring kernel stores the depth there and carries nothing through its loop for it.  Same depth as the gather
    kernel bit for bit (the same fma of the same plane index), same RGBA as without the request, and the checker's."""
    sc = make_scene(kind, n=32, size=64, steps=64, pose=pose, f32=f32, shade=1 if kind != "cfg2" else 0)
    (ref, rd) = sc.render(depth=True, blend=blend)
    push_scene(R, sc)
    R.set_blend(blend)
    try:
        R.set_option("kernel", 1)
        ga, gd = R.render(depth=True)
        R.set_option("kernel", 2)
        sa, sd = R.render(depth=True)
        assert R.last_frame_info()[0] == 2
        plain = R.render()
        assert np.array_equal(sa, plain) and np.array_equal(sa, ga)
        assert np.array_equal(np.isfinite(gd), np.isfinite(sd))
        fin = np.isfinite(gd)
        assert fin.any() and not fin.all()
        assert np.array_equal(gd[fin], sd[fin])
        assert np.array_equal(fin, np.isfinite(rd)) and np.abs(rd[fin] - sd[fin]).max() <= 1e-4
        assert np.abs(sa - ref).max() <= TOL
    finally:
        R.set_blend(0)
        R.set_option("kernel", 0)


@pytest.mark.parametrize("kind,pose,f32", [("cfg3", "diag", True), ("cfg3", "back", False), ("tf3d", "rot", True), ("cfg1", "side", False)])
def test_back_to_front_frames_on_the_slice_ring_kernel(R, kind, pose, f32):
    """Back-to-front frames (VolumeRenderer.cpp:590) are composited front to back by the slice-ring kernel: "over" is
    associative, the slices stream one way.  Against the gather kernel, which walks the planes in the reference's order,
    the blend is re-associated: a few ulp per sample (2e-5 asserted), the checker's tolerance against the checker; the
    first-hit depth -- the nearest contributing sample either way -- is the same number."""
    sc = make_scene(kind, n=32, size=64, steps=64, pose=pose, f32=f32, shade=1 if kind == "cfg3" else 0)
    (ref, rd) = sc.render(depth=True, blend=1)
    push_scene(R, sc)
    R.set_blend(1)
    try:
        R.set_option("kernel", 1)
        ga, gd = R.render(depth=True)
        R.set_option("kernel", 2)
        sa, sd = R.render(depth=True)
        assert R.last_frame_info()[0] == 2
        assert ref[..., 3].max() > 0.05
        assert np.abs(sa - ga).max() <= 2e-5
        assert np.abs(sa - ref).max() <= TOL
        fin = np.isfinite(gd)
        assert np.array_equal(fin, np.isfinite(sd)) and np.array_equal(gd[fin], sd[fin])
        R.set_option("kernel", 0)       # auto mode may now choose either kernel for such frames
        R.render()
        R.render()
        assert R.last_frame_info()[0] in (1, 2)
    finally:
        R.set_blend(0)
        R.set_option("kernel", 0)


@pytest.mark.parametrize("pose", ["z-", "y+", "x-"])
def test_ragged_volume_and_window(R, pose):
    sc = make_scene("cfg2", dims=(40, 24, 18), shade=1, pose=pose, f32=True)
    sc.width, sc.height, sc.steps = 93, 41, 70
    ref = sc.render()
    a, b = _both(R, sc)
    assert np.array_equal(a, b) and np.abs(b - ref).max() <= TOL


@pytest.mark.parametrize("dims,pose", [((39, 24, 18), "z+"), ((39, 25, 18), "y+"), ((24, 39, 19), "x+"), ((39, 39, 39), "diag")])
def test_u8_odd_extents_take_the_slice_ring_kernel(R, dims, pose):
    """8-byte voxels reach LDS in 16-byte units: an odd row length gets one pad voxel at upload
    (index N, which no clamped texel pair reaches), so these volumes are staged like any other."""
    sc = make_scene("cfg2", dims=dims, shade=1, pose=pose)
    ref = sc.render()
    a, b = _both(R, sc)
    assert np.array_equal(a, b) and np.abs(b - ref).max() <= TOL


@pytest.mark.parametrize("dims", [(4, 19, 3), (19, 3, 4), (3, 4, 19), (2, 2, 2), (2, 30, 30), (38, 56, 31)])
@pytest.mark.parametrize("pose", ["z+", "y-", "x+", "diag"])
def test_thin_and_tiny_volumes(R, dims, pose):
    """Two-, three- and four-slice volumes along the marching axis (the window of slice 1 of a
    three-slice volume reaches both faces) and volumes that project inside ONE pixel tile (all four
    corner rays of the tile miss them): found by the random-frame test, pinned here."""
    for f32 in (False, True):
        sc = make_scene("cfg3", dims=dims, shade=1, pose=pose, f32=f32)
        sc.width, sc.height = 31, 25
        sc.steps, sc.sample_rate = 0, 1.64
        ref = sc.render()
        a, b = _both(R, sc)
        assert np.array_equal(a, b) and np.abs(b - ref).max() <= TOL
        sc.width, sc.height = 139, 107
        sc.trans = (-0.15, 0.24, -0.21)
        ref = sc.render()
        a, b = _both(R, sc, upload=False)
        assert np.array_equal(a, b) and np.abs(b - ref).max() <= TOL


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
    sc = make_scene("cfg3", pert=True)          # a perturbed fetch leaves every staged window: the gather kernel's alone
    push_scene(R, sc)
    R.set_option("kernel", 2)
    with pytest.raises(smk.SmkError, match="not applicable"):
        R.render()
    R.set_option("kernel", 0)
    R.set_perturb(None, None, None)


@pytest.mark.parametrize("pose", ["id", "rot", "back", "side", "x+", "y-"])
def test_scalar_volume_with_the_1d_colour_table(R, pose):
    """cfg 1 -- VolumeRenderer's own case (u8 scalar, post-filter colour table, VolumeRenderer.cpp:576-587) -- on the
    slice-ring kernel: bit-identical to the gather kernel, <= 1e-4 from the checker; also as a float volume"""
    sc = make_scene("cfg1", n=32, size=72, steps=80, pose=pose)
    ref = sc.render()
    a, b = _both(R, sc)
    assert ref[..., 3].max() > 0.05
    assert np.array_equal(a, b) and np.abs(b - ref).max() <= TOL
    scf = make_scene("cfg1", n=32, size=72, steps=80, pose=pose)
    scf.data = (scf.data.astype(np.float32) / np.float32(255)).astype(np.float32)
    a, b = _both(R, scf)
    assert np.array_equal(a, b) and np.abs(b - scf.render()).max() <= TOL


def test_byte_offsets_beyond_4_gib_inside_a_slice_stride(gpu_renderer_factory):
    """S = y on a 512 x 512 x 1100 f32 volume: the window's V axis is z, whose stride is 4 MiB, so
    window origins above z = 1024 sit more than 4 GiB into the slab of one y-slice -- 32-bit
    byte arithmetic in the loaders passes every small-volume test and fails here.  No CPU
    reference at this size: the gather kernel (itself checked against the CPU at small sizes)
    is the reference, bit for bit."""
    import torch
    from _scenes import POSES, tf_cfg3, tf_h
    import oracle as O
    dims = (512, 512, 1100)
    r = gpu_renderer_factory()
    try:
        nx, ny, nz = dims
        scalar = torch.empty((nz, ny, nx), dtype=torch.uint8, device="cuda")
        r.synth_volume_device(0, 3, dims, scalar.data_ptr())
        vgh8 = torch.empty((nz, ny, nx, 3), dtype=torch.uint8, device="cuda")
        vghf = torch.empty((nz, ny, nx, 3), dtype=torch.float32, device="cuda")
        r.make_vgh_device(scalar.data_ptr(), 0, dims, 1, vgh8.data_ptr(), vghf.data_ptr())
        nrm = torch.empty((nz, ny, nx, 3), dtype=torch.uint8, device="cuda")
        r.normals_vgh_device(vgh8.data_ptr(), 3, dims, 0, nrm.data_ptr())
        del scalar, vgh8
        r.upload_volume_device(vghf.data_ptr(), dims, 3, 1, nrm.data_ptr())
        del vghf, nrm
        torch.cuda.empty_cache()
        sc = O.Scene(np.zeros((2, 2, 2, 3), np.float32))       # camera / shading parameters only
        sc.dims, sc.fsize = dims, tuple(np.float32(d / 1100.0) for d in dims)
        sc.xform = O.rotation(*POSES["y+"])
        sc.width = sc.height = 768          # ~1.4 voxels per pixel: windows must fit one DMA row
        sc.steps = 192
        r.set_option("tf_raw", 1)
        r.set_tf2d(tf_cfg3(), tf_h(0.5))
        r.set_camera(sc.mv(), sc.frustum, (sc.znear, 20.0), sc.width, sc.height)
        r.set_sampling(0.0, sc.steps, 1.0, 1)
        r.set_shading("r8k", sc.light_pos, sc.eye, sc.at, sc.xform, sc.intens)
        r.set_perturb(None, None, None)
        r.set_option("kernel", 1)
        a = r.render()
        r.set_option("kernel", 2)
        b = r.render()
        assert r.last_frame_info()[0] == 2
        assert a[..., 3].max() > 0.05
        assert np.array_equal(a, b), "slab and gather kernels differ: %g" % np.abs(a - b).max()
    finally:
        r.close()


def test_auto_mode_measures_both_kernels_and_settles(R):
    """kernel = 0: the first frames of a new configuration are trials -- six untimed slice-ring frames (its schedule
    settles on the workgroup times of earlier frames), an untimed gather frame, then a timed pair -- and the faster
    kernel is kept; every frame identical bit for bit."""
    sc = make_scene("cfg3", n=32, size=64, steps=64, pose="rot", f32=True, shade=1)
    push_scene(R, sc)
    R.set_option("kernel", 0)
    frames, kernels = [], []
    for _ in range(14):
        frames.append(R.render())
        kernels.append(R.last_frame_info()[0])
    assert kernels[:9] == [2, 2, 2, 2, 2, 2, 1, 2, 1]
    assert len(set(kernels[10:])) == 1 and kernels[10] in (1, 2)
    for f in frames[1:]:
        assert np.array_equal(f, frames[0])
    assert np.abs(frames[0] - sc.render()).max() <= TOL


def test_auto_mode_keeps_its_trial_timings_when_the_host_runs_far_ahead(gpu_renderer_factory):
    """150 frames enqueued without a synchronisation: the two timed trials' event pairs live in a ring of the last 64
    frames, and a decision taken after they had been recycled compared two later frames of the same kernel (a coin
    toss that could pin the slow kernel for 1024 frames).  The trials are waited for before their slots go."""
    import torch
    sc = make_scene("cfg3", n=128, size=512, steps=256, pose="rot", f32=True, shade=1)
    times = {}
    for forced in (1, 2):            # what each kernel really takes on this frame
        r = gpu_renderer_factory()
        push_scene(r, sc)
        r.set_option("kernel", forced)
        out = torch.zeros((sc.height * sc.width, 4), dtype=torch.float32, device="cuda")
        for _ in range(6):
            r.render_device(out.data_ptr())
        torch.cuda.synchronize()
        r.timing_reset()
        for _ in range(10):
            r.render_device(out.data_ptr())
        times[forced] = r.timing_read()[0]
        r.close()
    if max(times.values()) < 1.3 * min(times.values()):
        pytest.skip("the two kernels are too close on this box for the choice to be checkable: %r" % times)
    faster = min(times, key=times.get)
    for _ in range(3):               # (the coin toss came up wrong about every second time)
        r = gpu_renderer_factory()
        push_scene(r, sc)
        r.set_option("kernel", 0)
        out = torch.zeros((sc.height * sc.width, 4), dtype=torch.float32, device="cuda")
        for _ in range(150):
            r.render_device(out.data_ptr())
        torch.cuda.synchronize()
        for _ in range(3):
            r.render_device(out.data_ptr())
        torch.cuda.synchronize()
        assert r.last_frame_info()[0] == faster, times
        assert r.stat("slab_failures") == 0
        r.close()


@pytest.mark.parametrize("axis", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("pose", ["rot", "x+"])
def test_orthogonal_clip_plane(R, axis, pose):
    """gluvv.clip in its orthogonal mode: the volume ends at an axis-aligned plane through vpos
    (NV20VolRen3D.cpp:251-327) -- a smaller region for both kernels."""
    sc = make_scene("cfg4", n=32, size=64, steps=72, pose=pose, f32=True, shade=1)
    whole = sc.render()
    sc.clip = (axis, (0.37, 0.52, 0.61))
    ref = sc.render()
    assert np.abs(ref - whole).max() > 0.02           # the plane does cut something away
    a, b = _both(R, sc)
    assert np.array_equal(a, b) and np.abs(b - ref).max() <= TOL
    sc.clip = (axis, (-1.0, -1.0, -1.0) if axis % 2 else (2.0, 2.0, 2.0))   # everything clipped away
    a, b = _both(R, sc, upload=False)
    assert not a.any() and not b.any() and not sc.render().any()
    sc.clip = None
    a, b = _both(R, sc, upload=False)
    assert np.array_equal(a, b) and np.abs(b - whole).max() <= TOL


def test_clip_plane_outside_a_shard_leaves_it_empty(gpu_renderer_factory):
    """a clip plane on the far side of a rank's brick region: nothing of that rank is visible; the
    slice-ring kernel hands such frames to the gather kernel (its membership test needs lo <= hi)"""
    from simian_spacemonkey_amd import sortlast
    sc = make_scene("cfg2", dims=(70, 44, 56), shade=1, pose="rot")
    sc.clip = (5, (0.44, 0.18, 0.11))                 # keeps z below ~7.5 voxels
    r = gpu_renderer_factory()
    try:
        r.set_shard(4, 8)                              # owns z >= 28
        push_scene(r, sc)
        sc.region = sortlast.shard_region(sc.dims, 4, 8)
        assert not sc.render().any()
        r.set_option("kernel", 0)
        assert not r.render().any() and r.last_frame_info()[0] == 1
        r.set_option("kernel", 2)
        with pytest.raises(Exception, match="region is empty"):
            r.render()
    finally:
        r.close()


@pytest.mark.parametrize("pose", ["rot", "z-", "x+", "y-"])
@pytest.mark.parametrize("f32", [True, False])
def test_free_clip_plane_on_the_slice_ring_kernel(R, pose, f32):
    """gluvv.clip in its free mode = glClipPlane under the widget's matrix (NV20VolRen3D.cpp:346-357): a half-space in
    eye space.  Along a ray the plane's value is monotone in the plane index, so the kept samples are an interval: the
    slice-ring kernel folds it into each ray's plane range in its set-up (nothing in the marching loop) and renders the
    frame bit for bit like the gather kernel's per-sample test, with and without the brick flags, split in depth or not."""
    sc = make_scene("cfg2", n=32, size=64, steps=72, pose=pose, f32=f32, shade=1)
    whole = sc.render()
    n = np.array([0.35, -0.2, -0.9])
    n /= np.linalg.norm(n)
    mv = np.array(sc.mv(), np.float64).reshape(4, 4).T   # column-major -> rows
    centre = mv @ np.array([float(sc.fsize[0]) / 2, float(sc.fsize[1]) / 2, float(sc.fsize[2]) / 2, 1.0])
    for off, flip in ((0.03, 1.0), (-0.11, -1.0), (0.6, 1.0), (-0.6, 1.0)):     # through the middle, the other side, nothing cut, everything cut
        sc.clip_plane = tuple(flip * v for v in (n[0], n[1], n[2], -float(n @ centre[:3]) + off))
        ref = sc.render()
        a, b = _both(R, sc)
        assert np.array_equal(a, b), "slab and gather kernels differ: %g" % np.abs(a - b).max()
        assert np.abs(b - ref).max() <= TOL
        if abs(off) < 0.2:
            assert np.abs(ref - whole).max() > 0.02 and ref[..., 3].max() > 0.05
    R.set_option("slab_split", 3)
    try:
        a, b = _both(R, sc)
        assert np.abs(a - b).max() <= 2e-5
    finally:
        R.set_option("slab_split", 0)
    sc.clip_plane = None
    a, b = _both(R, sc, upload=False)
    assert np.array_equal(a, b) and np.abs(b - whole).max() <= TOL


def test_a_failed_slice_ring_frame_is_rendered_again_and_not_tried_twice(gpu_renderer_factory):
    """Safety net: should the slice-ring kernel ever report a time-out or a violated window bound,
    the synchronous entry re-renders that frame on the gather kernel in auto mode (the caller gets
    a valid frame), the configuration stays with the gather kernel, and a forced slice-ring frame
    reports the failure.  The status word is injected through a test hook."""
    import torch
    r = gpu_renderer_factory()
    try:
        sc = make_scene("cfg3", n=32, size=64, steps=64, pose="rot", f32=True, shade=1)
        ref = sc.render()
        push_scene(r, sc)
        r.set_option("kernel", 0)
        for _ in range(11):                                  # trials done, slice-ring kernel chosen
            img = r.render()
        assert r.last_frame_info()[0] == 2
        r.set_option("inject_slab_status", 2)
        img = r.render()                                     # fails behind the scenes, rendered again
        assert r.last_frame_info()[0] == 1 and np.abs(img - ref).max() <= TOL
        assert r.stat("slab_status") == 0
        img = r.render()
        assert r.last_frame_info()[0] == 1 and np.abs(img - ref).max() <= TOL   # not tried twice
        r.set_option("kernel", 2)
        r.set_option("inject_slab_status", 1)
        with pytest.raises(Exception, match="time-out"):
            r.render()
        assert np.abs(r.render() - ref).max() <= TOL         # the next forced frame is fine
        # asynchronous entry: the host asks about the frame (smk_frame_failed), after which auto mode uses the gather kernel
        sc.steps = 65                                        # a new configuration
        push_scene(r, sc, upload=False)
        r.set_option("kernel", 0)
        out = torch.zeros((64 * 64, 4), dtype=torch.float32, device="cuda")
        r.set_option("inject_slab_status", 2)
        r.render_device(out.data_ptr(), None, None)          # first trial frame = slice-ring kernel
        bad = r.last_frame_id()
        torch.cuda.synchronize()
        assert r.frame_failed(bad) == 1
        r.render_device(out.data_ptr(), None, None)
        torch.cuda.synchronize()
        assert r.last_frame_info()[0] == 1
        assert np.abs(out.cpu().numpy().reshape(64, 64, 4) - sc.render()).max() <= TOL
        # ... and a flagged frame nobody asks about fails the call that takes over its status slot, eight frames on
        r.set_option("kernel", 2)
        r.set_option("inject_slab_status", 2)
        r.render_device(out.data_ptr(), None, None)
        for _ in range(7):
            r.render_device(out.data_ptr(), None, None)
        torch.cuda.synchronize()
        with pytest.raises(Exception, match="window outside"):
            r.render_device(out.data_ptr(), None, None)
        r.render_device(out.data_ptr(), None, None)
        torch.cuda.synchronize()
    finally:
        r.close()


def test_frames_in_flight_report_their_own_status(gpu_renderer_factory):
    """smk_render_device only enqueues; each frame has its own status word.  A caller that keeps frames in
    flight asks about a frame after synchronising with it (smk_frame_failed), renders a flagged one again
    and goes on: no later call fails, and the counters a benchmark asserts on show what happened."""
    import torch
    r = gpu_renderer_factory()
    try:
        sc = make_scene("cfg3", n=32, size=64, steps=64, pose="rot", f32=True, shade=1)
        ref = sc.render()
        push_scene(r, sc)
        r.set_option("kernel", 2)
        out = torch.zeros((2, 64 * 64, 4), dtype=torch.float32, device="cuda")
        r.render_device(out[0].data_ptr(), None, None)
        a = r.last_frame_id()
        r.set_option("inject_slab_status", 1)
        r.render_device(out[1].data_ptr(), None, None)        # a second frame behind the first, flagged
        b = r.last_frame_id()
        assert b == a + 1
        torch.cuda.synchronize()
        assert r.stat("slab_failures") == 1                   # seen without being consumed
        assert r.frame_failed(a) == 0 and r.frame_failed(b) == 1
        assert r.frame_failed(b) == 0                         # asked and answered
        assert r.frame_failed(b + 5) == -1 and r.frame_failed(0) == -1   # never enqueued: unknown, not "valid"
        r.set_option("kernel", 1)
        r.render_device(out[1].data_ptr(), None, None)        # the repair: no error from the earlier frame
        torch.cuda.synchronize()
        assert np.abs(out[1].cpu().numpy().reshape(64, 64, 4) - ref).max() <= TOL
        assert r.stat("slab_failures") == 1 and r.stat("slab_retries") == 0
    finally:
        r.close()


def test_a_pose_the_slice_ring_kernel_declines_does_not_pin_later_frames(gpu_renderer_factory):
    """Auto mode plans every frame afresh: a wide-angle close-up (rays up to ~74 degrees off the principal
    axis: 'too oblique') runs on the gather kernel, and the very next ordinary frame is back on the
    slice-ring kernel -- a declined pose is not remembered against the configuration."""
    r = gpu_renderer_factory()
    try:
        sc = make_scene("cfg3", n=32, size=64, steps=64, pose="rot", f32=True, shade=1)
        push_scene(r, sc)
        r.set_option("kernel", 0)
        for _ in range(6):
            r.render()
        assert r.last_frame_info()[0] == 2
        wide = make_scene("cfg3", n=32, size=64, steps=64, pose="rot", f32=True, shade=1)
        wide.eye, wide.frustum = (0, 0, -1.1), (-3.5, 3.5, -3.5, 3.5)
        push_scene(r, wide, upload=False)
        img = r.render()
        assert r.last_frame_info()[0] == 1
        assert np.abs(img - wide.render()).max() <= TOL
        push_scene(r, sc, upload=False)
        for _ in range(6):                                    # (at most: the two kernels are timed again)
            img = r.render()
        assert r.last_frame_info()[0] == 2
        assert np.abs(img - sc.render()).max() <= TOL
    finally:
        r.close()


@pytest.mark.parametrize("k", [2, 3, 8])
@pytest.mark.parametrize("pose,f32,kind", [("rot", True, "cfg3"), ("z-", False, "cfg4"), ("x+", True, "cfg4"), ("y-", True, "tf3d_panes")])
def test_depth_segments(R, k, pose, f32, kind):
    """A tile rendered by k workgroups, each taking a run of its slice positions, the partial frames merged in marching
    order (option slab_split; automatic for tiles measured long): the same samples, the blend re-associated -- a few ulp per
    segment against the gather kernel (2e-5), the stated 1e-4 against the checker.  Every sample is taken exactly once:
    rays along segment seams, thin ranges (more segments than positions) and both marching directions."""
    sc = make_scene(kind, n=32, size=72, steps=80, pose=pose, f32=f32, shade=1)
    ref = sc.render()
    R.set_option("slab_split", k)
    try:
        a, b = _both(R, sc)
        assert R.stat("slab_split_tiles") > 0
    finally:
        R.set_option("slab_split", 0)
    assert np.abs(a - b).max() <= 2e-5
    assert np.abs(b - ref).max() <= TOL


def test_depth_segments_few_planes_and_max_blend(R):
    sc = make_scene("cfg3", n=32, size=64, steps=9, pose="rot", f32=True, shade=1)      # sample spacing of several slices
    ref = sc.render()
    R.set_option("slab_split", 5)
    try:
        a, b = _both(R, sc)
        assert np.abs(a - b).max() <= 2e-5 and np.abs(b - ref).max() <= TOL
        sc = make_scene("cfg3", n=32, size=64, steps=64, pose="side", f32=True, shade=1)
        push_scene(R, sc)
        R.set_blend(2)
        R.set_option("kernel", 1)
        a = R.render()
        R.set_option("kernel", 2)
        b = R.render()
        assert np.array_equal(a, b)      # a maximum does not care about association
    finally:
        R.set_blend(0)
        R.set_option("slab_split", 0)
        R.set_option("kernel", 0)


def test_a_pipelining_host_asks_about_frame_i_after_enqueuing_frame_i_plus_1(gpu_renderer_factory):
    """The sort-last pipeline's order of calls: render(i), render(i + 1), then frame_failed(i).  The second render call
    must not consume (or trip over) frame i's status word -- it only looks at the slot it takes over -- so the host still
    learns that frame i was flagged, repairs it locally, and no rank is left waiting in a collective."""
    import torch
    r = gpu_renderer_factory()
    try:
        sc = make_scene("cfg3", n=32, size=64, steps=64, pose="rot", f32=True, shade=1)
        ref = sc.render()
        push_scene(r, sc)
        r.set_option("kernel", 2)
        out = torch.zeros((2, 64 * 64, 4), dtype=torch.float32, device="cuda")
        r.set_option("inject_slab_status", 1)
        r.render_device(out[0].data_ptr(), None, None)        # frame i, flagged
        i = r.last_frame_id()
        torch.cuda.synchronize()                              # (its word has landed by the time the next call is made)
        r.render_device(out[1].data_ptr(), None, None)        # frame i + 1: must not raise
        assert r.last_frame_id() == i + 1
        torch.cuda.synchronize()
        assert r.frame_failed(i) == 1 and r.frame_failed(i + 1) == 0
        r.set_option("kernel", 1)
        r.render_device(out[0].data_ptr(), None, None)        # the repair
        torch.cuda.synchronize()
        for k in range(2):
            assert np.abs(out[k].cpu().numpy().reshape(64, 64, 4) - ref).max() <= TOL
        assert r.stat("slab_failures") == 1
    finally:
        r.close()
