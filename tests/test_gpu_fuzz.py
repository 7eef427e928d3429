"""Seeded random frames: ragged volume sizes, any rotation, off-centre and close-up eyes, wide
frusta, odd viewports, both voxel types, every classification / shading mode the slice-ring kernel
takes.  Per case: the slice-ring frame must equal the gather kernel's bit for bit (or the kernel
must decline with a reason, never fail), its status word must stay 0, and the frame must match the
CPU checker within the suite's tolerance.  SMK_FUZZ_CASES / SMK_FUZZ_SEED widen the run by hand."""
import os

import numpy as np
import pytest

import _scenes as S
from _scenes import O, push_scene

pytestmark = pytest.mark.gpu
TOL = 1e-4
NCASES = int(os.environ.get("SMK_FUZZ_CASES", "40"))
SEED = int(os.environ.get("SMK_FUZZ_SEED", "20240"))


def random_scene(rng):
    u = rng.random()
    hi = 150 if u < 0.08 else 72 if u < 0.3 else 40
    if os.environ.get("SMK_FUZZ_BIG") == "1" and u < 0.08:
        hi = 330                               # by hand: windows big enough for the 16-wave workgroups
    dims = tuple(int(rng.integers(2, hi + 1)) for _ in range(3))
    if rng.random() < 0.2:                      # a slab-shaped volume: one axis very thin
        a = int(rng.integers(0, 3))
        dims = tuple(int(rng.integers(2, 5)) if i == a else d for i, d in enumerate(dims))
    f32 = bool(rng.integers(0, 2))
    vgh8, vghf, nrm = S.ragged_vgh(dims, seed=int(rng.integers(1, 1000)))
    sc = O.Scene(vghf if f32 else vgh8, grad=nrm)
    kind = ["cfg2", "cfg3", "cfg4"][int(rng.integers(0, 3))]
    sc.tf_mode = 1
    if kind == "cfg2":
        sc.tf_vg, sc.tf_h = S.tf_cfg2()
    else:
        sc.tf_vg = S.tf_cfg3()
    if kind == "cfg4":
        sc.tf_h = S.tf_h(float(rng.uniform(0.2, 0.8)))
        sc.third_axis = 1
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis) + 1e-9
    sc.xform = O.rotation(tuple(float(a) for a in axis), float(rng.uniform(-180, 180)))
    wide = 420 if u < 0.08 else 150
    sc.width = int(rng.integers(9, wide))
    sc.height = int(rng.integers(9, wide))
    sc.steps = int(rng.integers(6, 2 * wide // 3 + 60))
    sc.shade_mode = int(rng.integers(0, 3))
    sc.use_spec = int(rng.integers(0, 2))
    r = rng.random()
    if r < 0.25:                                # close-up: strong perspective, part of the volume off screen
        sc.eye = (float(rng.uniform(-.3, .3)), float(rng.uniform(-.3, .3)), -float(rng.uniform(1.6, 2.5)))
        w = float(rng.uniform(0.15, 0.45))
        sc.frustum = (-w, w, -w, w)
    elif r < 0.5:                               # panned
        sc.trans = (float(rng.uniform(-.6, .6)), float(rng.uniform(-.6, .6)), float(rng.uniform(-1, 1)))
    elif r < 0.6:                               # asymmetric frustum
        sc.frustum = (-0.03, 0.11, -0.09, 0.05)
    if rng.random() < 0.15:
        sc.steps, sc.sample_rate = 0, float(rng.uniform(0.3, 2.5))
    sc.clip = None
    if rng.random() < 0.12:                    # the clip-plane widget in its orthogonal mode
        sc.clip = (int(rng.integers(1, 7)), tuple(float(rng.uniform(0.1, 0.9)) * float(f) for f in sc.fsize))
    sc.shard = None
    if rng.random() < 0.2 and min(dims) >= 4:  # one rank's brick region of a sort-last job
        world = int(rng.choice([2, 4, 8]))
        sc.shard = (int(rng.integers(0, world)), world)
    return sc, kind, f32, dims


def one_case(R, sc, tag):
    """returns None when both kernels rendered the frame, else the slice-ring kernel's reason for declining"""
    ref = sc.render()
    push_scene(R, sc)
    R.set_option("kernel", 1)
    a = R.render()
    assert np.abs(a - ref).max() <= TOL, tag + ": gather kernel vs CPU checker %g" % np.abs(a - ref).max()
    R.set_option("kernel", 2)
    try:
        b = R.render()
    except Exception as e:                 # forced slice-ring kernel on a frame it does not take
        assert "not applicable" in str(e), tag + ": " + str(e)
        assert R.stat("slab_status") == 0, tag
        return str(e).split("not applicable:")[-1].strip()[:80]
    finally:
        R.set_option("kernel", 0)
    assert R.stat("slab_status") == 0, tag
    assert np.array_equal(a, b), tag + ": slice-ring vs gather %g" % np.abs(a - b).max()
    return None


def test_random_frames(gpu_renderer_factory):
    rng = np.random.default_rng(SEED)
    from simian_spacemonkey_amd import sortlast
    R0 = gpu_renderer_factory()
    took = declined = 0
    reasons = {}
    try:
        for case in range(NCASES):
            sc, kind, f32, dims = random_scene(rng)
            tag = "case %d (seed %d): %s dims %s f32 %d %dx%d x%d shade %d" % (
                case, SEED, kind, dims, f32, sc.width, sc.height, sc.steps, sc.shade_mode)
            R = R0
            if sc.shard:
                R = gpu_renderer_factory()         # (a context is sharded before its first upload)
                R.set_shard(*sc.shard)
                sc.region = sortlast.shard_region(sc.dims, *sc.shard)
                tag += " shard %d/%d" % sc.shard
            try:
                why = one_case(R, sc, tag)
            finally:
                if R is not R0:
                    R.close()
            if why:
                declined += 1
                reasons[why] = reasons.get(why, 0) + 1
            else:
                took += 1
    finally:
        R0.close()
    print("slice-ring kernel took %d frames, declined %d: %s" % (took, declined, reasons))
    assert took >= NCASES // 2, "slice-ring kernel declined %d of %d frames" % (declined, NCASES)


def test_interactive_session_in_auto_mode(gpu_renderer_factory):
    """A camera wandering for 60 frames with the kernel choice left to the library: trial frames,
    measured tile weights and schedule refreshes all happen along the way, and every frame must be
    the frame the gather kernel renders for that pose (the two kernels are bit-identical)."""
    rng = np.random.default_rng(SEED + 1)
    vgh8, vghf, nrm = S.ragged_vgh((72, 64, 56), seed=9)
    sc = O.Scene(vghf, grad=nrm)
    sc.tf_mode, sc.tf_vg, sc.tf_h, sc.third_axis = 1, S.tf_cfg3(), S.tf_h(0.5), 1
    sc.width, sc.height, sc.steps, sc.shade_mode = 200, 168, 120, 1
    A, B = gpu_renderer_factory(), gpu_renderer_factory()
    try:
        axis = np.array([0.3, 1.0, 0.1])
        angle = 0.0
        kernels = []
        for frame in range(60):
            axis = axis + 0.15 * rng.normal(size=3)
            axis /= np.linalg.norm(axis)
            angle += float(rng.uniform(2, 9))          # walks through every principal axis
            sc.xform = O.rotation(tuple(float(a) for a in axis), angle)
            push_scene(A, sc, upload=frame == 0)
            push_scene(B, sc, upload=frame == 0)
            A.set_option("kernel", 0)
            B.set_option("kernel", 1)
            a, b = A.render(), B.render()
            kernels.append(A.last_frame_info()[0])
            assert A.stat("slab_status") == 0
            assert np.array_equal(a, b), "frame %d (kernel %d): %g" % (frame, kernels[-1], np.abs(a - b).max())
        assert 2 in kernels                              # the slice-ring kernel did take part
    finally:
        A.close()
        B.close()


def test_random_frames_gather_only_modes(gpu_renderer_factory):
    """The modes only the gather kernel takes -- 1-D TLUT on scalar data, dense 3-D transfer
    function, noise-perturbed fetches, first-hit depth, bricked uploads -- on random volumes and
    poses, against the CPU checker."""
    rng = np.random.default_rng(SEED + 2)
    R = gpu_renderer_factory()
    try:
        for case in range(max(12, NCASES // 2)):
            dims = tuple(int(rng.integers(5, 41)) for _ in range(3))
            mode = ["cfg1", "tf3d", "pert", "depth", "bricks"][case % 5]
            if mode == "bricks":        # MetaVolume::brick drops the remainder voxels of an odd size
                dims = tuple(max(8, d & ~1) for d in dims)   # (MetaVolume.cpp:1394-1396): keep it whole
            f32 = bool(rng.integers(0, 2))
            if mode == "cfg1":
                nz, ny, nx = dims[2], dims[1], dims[0]
                v = rng.integers(0, 256, size=(nz, ny, nx, 1), dtype=np.uint8)
                sc = O.Scene(v)
                sc.tf_mode, sc.tlut = 0, O.tlut_volumerenderable()
            else:
                vgh8, vghf, nrm = S.ragged_vgh(dims, seed=int(rng.integers(1, 1000)))
                sc = O.Scene(vghf if f32 else vgh8, grad=nrm)
                sc.tf_mode, sc.tf_vg = 1, S.tf_cfg3()
                if mode == "tf3d":
                    sc.tf_mode, sc.tf3d = 2, S.tf3d_dense()
                if mode == "pert":
                    sc.noise = O.noise_tex(32)
                    sc.pert_w = (float(rng.uniform(0, .3)), float(rng.uniform(0, .2)), 0, 0)
                    sc.pert_s = (.2, 2.1, 4.5, 8.7)
            axis = rng.normal(size=3)
            axis /= np.linalg.norm(axis) + 1e-9
            sc.xform = O.rotation(tuple(float(a) for a in axis), float(rng.uniform(-180, 180)))
            sc.width, sc.height = int(rng.integers(9, 120)), int(rng.integers(9, 120))
            sc.steps = int(rng.integers(6, 120))
            sc.shade_mode = 0 if mode == "cfg1" else int(rng.integers(0, 3))
            tag = "case %d (seed %d): %s dims %s f32 %d %dx%d x%d shade %d" % (
                case, SEED + 2, mode, dims, f32, sc.width, sc.height, sc.steps, sc.shade_mode)
            grid = (2, 2, 2) if mode == "bricks" else (1, 1, 1)
            push_scene(R, sc, grid)
            R.set_option("kernel", 0)
            if mode == "depth":
                ref, rd = sc.render(depth=True)
                img, dep = R.render(depth=True)
                fin = np.isfinite(rd)
                assert np.array_equal(fin, np.isfinite(dep)), tag
                assert not fin.any() or np.abs(rd[fin] - dep[fin]).max() <= 1e-4, tag
            else:
                ref, img = sc.render(), R.render()
            assert np.abs(img - ref).max() <= TOL, tag + ": %g" % np.abs(img - ref).max()
    finally:
        R.close()
