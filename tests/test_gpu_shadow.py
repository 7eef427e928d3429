"""GPU parity of the half-angle-slicing shadow mode (smk_set_shadow, smk_shadow.hip) against the CPU
checker: identical set-up coefficients, frames and light buffers within the renderer's tolerance
(max-abs <= 1e-4 on premultiplied fp32 RGBA; sample placement is bit-identical by construction)."""
import numpy as np
import pytest

from _scenes import make_scene, push_scene

pytestmark = pytest.mark.gpu
TOL = 1e-4

LIGHTS = {"eye_side": (0, 0, -5), "oblique": (3, 4, -3), "behind": (-2, 3, 4), "side": (5, 1, 0.5)}


@pytest.fixture(scope="module")
def R(gpu_renderer_factory):
    r = gpu_renderer_factory()
    yield r
    r.close()


def _fields(a):
    out = {}
    for f, _ in a._fields_:
        v = getattr(a, f)
        out[f] = list(v) if hasattr(v, "__len__") else v
    return out


def _check(R, sc, tol=TOL):
    push_scene(R, sc)
    assert _fields(R.shadowcoef()) == _fields(sc.shadowcoef())
    ref, refL = sc.render_shadow()
    got = R.render()
    gotL = R.light_buffer()
    assert R.last_frame_info()[0] in (1, 2)   # (the eye pass is a frame of the ray-marchers; 3 = a launch per slice)
    assert ref[..., 3].max() > 0.05 and refL[..., 3].max() > 0.05, "vacuous scene"
    assert gotL.shape == refL.shape
    eL, e = np.abs(gotL - refL).max(), np.abs(got - ref).max()
    assert eL <= tol, f"light buffer max abs err {eL}"
    assert e <= tol, f"frame max abs err {e}"
    return ref, got


@pytest.mark.parametrize("light", sorted(LIGHTS))
@pytest.mark.parametrize("f32", [False, True])
def test_cfg3_with_shadows(R, light, f32):
    sc = make_scene("cfg3", f32=f32, shade=1)
    sc.light_pos = LIGHTS[light]
    sc.shadow = (128, 0.5)
    ref, _ = _check(R, sc)
    plain = sc.render()
    assert (ref[..., :3].sum() < 0.97 * plain[..., :3].sum()) or light == "behind"   # the shadow term does something


@pytest.mark.parametrize("kind,shade,pose", [("cfg2", 0, "id"), ("cfg4", 1, "side"), ("tf3d", 0, "rot"), ("tf3d", 1, "back")])
def test_other_classifications_and_poses(R, kind, shade, pose):
    sc = make_scene(kind, shade=shade, pose=pose)
    sc.light_pos = LIGHTS["oblique"]
    sc.shadow = (96, 0.7)       # 67.2 -> 68 texels: a ragged light buffer
    _check(R, sc)


def test_ragged_volume_sample_rate_mode_odd_viewport(R):
    sc = make_scene("cfg3", dims=(40, 24, 18), shade=1, size=45)
    sc.steps, sc.sample_rate = 0, 1.5
    sc.light_pos = LIGHTS["side"]
    sc.shadow = (64, 1.0)
    _check(R, sc)


def test_larger_frame_many_slices(R):
    """64^3 f32 volume, 160^2 viewport, 200 slices, 256^2 light buffer: the recurrence over slices does not drift"""
    sc = make_scene("cfg4", n=64, size=160, steps=200, f32=True, shade=1, pose="diag")
    sc.light_pos = LIGHTS["oblique"]
    sc.shadow = (512, 0.5)
    _check(R, sc)


def test_shadows_off_restores_the_plain_frame(R):
    sc = make_scene("cfg3", shade=1)
    sc.shadow = (64, 0.5)
    push_scene(R, sc)
    R.render()
    sc.shadow = None
    push_scene(R, sc, upload=False)
    got = R.render()
    assert R.last_frame_info()[0] in (1, 2)
    assert np.abs(got - sc.render()).max() <= TOL


def test_unsupported_combinations_fail_with_the_reason(R, smk, gpu_renderer_factory):
    sc = make_scene("cfg1")
    sc.shadow = (64, 0.5)
    push_scene(R, sc)
    with pytest.raises(smk.SmkError, match="2-D or 3-D transfer function"):
        R.render()
    sc = make_scene("cfg3", shade=2)
    sc.shadow = (64, 0.5)
    push_scene(R, sc)
    with pytest.raises(smk.SmkError, match="NV20"):
        R.render()
    sc = make_scene("cfg3", pert=True)
    sc.shadow = (64, 0.5)
    push_scene(R, sc)
    with pytest.raises(smk.SmkError, match="perturbation"):
        R.render()
    sc = make_scene("cfg3")
    sc.shadow = (64, 0.5)
    sc.light_pos = (0, 4, 0)
    push_scene(R, sc)
    with pytest.raises(smk.SmkError, match="y axis"):
        R.render()
    with pytest.raises(smk.SmkError, match="quality"):
        R.set_shadow(1, 64, 0.0)
    r2 = gpu_renderer_factory()
    try:
        r2.set_shard(0, 2)
        sc = make_scene("cfg3")
        sc.shadow = (64, 0.5)
        push_scene(r2, sc)
        with pytest.raises(smk.SmkError, match="whole volume on one GPU"):
            r2.render()
    finally:
        r2.close()
    R.set_shadow(0)


def test_all_slices_in_one_cooperative_launch(R):
    """Option shadow_fused: the slices of a frame as ONE launch with a grid barrier between them, the frame and the light
    buffer accessed at device scope.  Slower than a launch per slice on MI355X, kept as an option: the same
    arithmetic, so frame and light buffer are bit-identical to the per-slice launches'."""
    sc = make_scene("cfg4", n=64, size=160, steps=200, f32=True, shade=1, pose="diag")
    sc.light_pos = LIGHTS["oblique"]
    sc.shadow = (512, 0.5)
    push_scene(R, sc)
    a = R.render()
    la = R.light_buffer()
    R.set_option("shadow_fused", 1)
    try:
        b = R.render()
        lb = R.light_buffer()
        assert R.last_frame_info()[0] == 3
    finally:
        R.set_option("shadow_fused", 0)
    assert np.array_equal(a, b) and np.array_equal(la, lb)


@pytest.mark.parametrize("light", sorted(LIGHTS))
@pytest.mark.parametrize("kind,f32,shade,pose", [("cfg3", True, 1, "rot"), ("cfg3", False, 1, "diag"), ("tf3d", False, 0, "back"),
                                                 ("cfg4", True, 1, "side")])
def test_two_marches_equal_a_launch_per_slice(R, light, kind, f32, shade, pose):
    """The default since round 3: one march per light-buffer texel (its value depends on itself alone from slice to slice)
    that keeps every slice's light buffer, then the eye pass as an ordinary frame of the ray-marchers over the half-angle
    slices, looking slice k's shading up in buffer k - 1.  The same samples and the same operations as a launch per slice
    (option shadow_march 0): light buffers bit-identical; frames bit-identical where the slices run away from the viewer
    (the blend keeps its order), and within the re-association of the blend where they run towards the viewer (the marchers
    composite front to back what the per-slice form blends back to front).  Both ray-marchers agree bit for bit."""
    sc = make_scene(kind, n=48, size=112, steps=150, f32=f32, shade=shade, pose=pose)
    sc.light_pos = LIGHTS[light]
    sc.shadow = (96, 0.7)
    push_scene(R, sc)
    f2b = R.shadowcoef().front_to_back
    out = {}
    try:
        for kern in (1, 2):
            R.set_option("kernel", kern)
            out[kern] = (R.render(), R.light_buffer())
            assert R.last_frame_info()[0] == kern
        R.set_option("kernel", 0)
        R.set_option("shadow_march", 0)
        b = R.render()
        lb = R.light_buffer()
        assert R.last_frame_info()[0] == 3
    finally:
        R.set_option("shadow_march", 1)
        R.set_option("kernel", 0)
    assert b[..., 3].max() > 0.05 and lb[..., 3].max() > 0.05, "vacuous scene"
    assert np.array_equal(out[1][0], out[2][0])
    for kern in (1, 2):
        assert np.array_equal(out[kern][1], lb)
        if f2b:
            assert np.array_equal(out[kern][0], b)
        else:
            assert np.abs(out[kern][0] - b).max() <= 2e-5


@pytest.mark.parametrize("light", ["oblique", "behind"])
@pytest.mark.parametrize("which", ["orthogonal", "free", "both"])
def test_shadows_with_clip_planes(R, which, light):
    """Round 3: frames with shadows take the clip-plane widget's planes.  The reference draws the same clipped slice
    polygons in both passes (volShadow slices the box setupClips left, NV20VolRen3D.cpp:251-327; glClipPlane stays
    enabled, :346-357), so neither the eye nor the light pass has a sample beyond a plane: frame and light buffer against
    the CPU checker, the two ray-marchers bit for bit, and the two marches against a launch per slice."""
    sc = make_scene("cfg3", n=48, size=112, steps=150, f32=True, shade=1, pose="rot")
    sc.light_pos = LIGHTS[light]
    sc.shadow = (96, 0.7)
    if which in ("orthogonal", "both"):
        sc.clip = (3, tuple(0.55 * float(f) for f in sc.fsize))      # Y+: what lies below y = .55 stays
    if which in ("free", "both"):
        n = np.array([0.35, -0.2, -0.9])
        n /= np.linalg.norm(n)
        mv = np.array(sc.mv(), np.float64).reshape(4, 4).T
        centre = mv @ np.array([float(sc.fsize[0]) / 2, float(sc.fsize[1]) / 2, float(sc.fsize[2]) / 2, 1.0])
        sc.clip_plane = (n[0], n[1], n[2], -float(n @ centre[:3]) + 0.03)
    try:
        ref, _ = _check(R, sc)
        unclipped = make_scene("cfg3", n=48, size=112, steps=150, f32=True, shade=1, pose="rot")
        unclipped.light_pos, unclipped.shadow = sc.light_pos, sc.shadow
        full, _ = unclipped.render_shadow()
        assert np.abs(full - ref).max() > 1e-2              # (the planes do cut something)
        f2b = R.shadowcoef().front_to_back
        out = {}
        for kern in (1, 2):
            R.set_option("kernel", kern)
            out[kern] = (R.render(), R.light_buffer())
            assert R.last_frame_info()[0] == kern
        R.set_option("kernel", 0)
        R.set_option("shadow_march", 0)
        b, lb = R.render(), R.light_buffer()
        assert np.array_equal(out[1][0], out[2][0]) and np.array_equal(out[1][1], lb) and np.array_equal(out[2][1], lb)
        assert np.array_equal(out[2][0], b) if f2b else np.abs(out[2][0] - b).max() <= 2e-5
    finally:
        R.set_option("shadow_march", 1)
        R.set_option("kernel", 0)
        sc.clip = None
        sc.clip_plane = None
        push_scene(R, sc)
