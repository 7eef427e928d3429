"""The C++ host-side mirror of the reference renderer interface
(simian-spacemonkey_amd/host/HipVolumeRenderer.{h,cpp}: createVolume / createTLUT / getColorMap /
renderVolume + a gluvvPrimitive with init()/draw()) driven exactly like Simian's main() does
(tests/host/adapter_main.cpp), compared with the CPU checker."""
import os
import subprocess

import numpy as np
import pytest

from _scenes import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host", "adapter_main")


def _run(tmp_path, sc, shade, rate, deptex, extra=()):
    vol = tmp_path / "vol.u8"
    sc.data.tofile(vol)
    grad = "-"
    if sc.grad is not None:
        grad = tmp_path / "grad.u8"
        sc.grad.tofile(grad)
    dep = "-"
    if deptex is not None:
        dep = tmp_path / "deptex.rgba"
        deptex.tofile(dep)
    out = tmp_path / "frame.f32"
    nx, ny, nz = sc.dims
    cmd = [EXE, str(vol), str(nx), str(ny), str(nz), str(sc.nelts), str(grad), str(dep),
           str(sc.width), str(sc.height), repr(rate), str(shade)] + [repr(float(v)) for v in sc.xform] + [str(out)] + list(extra)
    p = subprocess.run(cmd, capture_output=True, text=True)
    return p, out


def test_adapter_builds_and_refuses_to_run_without_a_gpu(tmp_path):
    import torch
    assert os.path.exists(EXE), "build with __graft_entry__.build()"
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = make_scene("cfg1", n=16, size=16)
    p, _ = _run(tmp_path, sc, 1, 1.0, None)
    assert p.returncode == 3 and "no HIP device" in p.stderr   # loud failure, no CPU path


@pytest.mark.gpu
def test_scalar_path_like_volumerenderable(tmp_path, O):
    sc = make_scene("cfg1", n=24, size=40, pose="rot")
    sc.steps, sc.sample_rate = 0, 1.5
    t = O.tlut("default", 256)
    t[:, 3] = (0.1 * np.arange(256) / 255).astype(np.float32)
    sc.tlut = O.tlut_scale_alpha(t, 1.0, 1.5)           # what draw() does before the frame
    p, out = _run(tmp_path, sc, 1, 1.5, None)
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, np.float32).reshape(sc.height, sc.width, 4)
    ref = sc.render()
    assert ref[..., 3].max() > 0.05 and np.abs(got - ref).max() <= 1e-4


@pytest.mark.gpu
def test_vgh_path_like_nv20volren3d(tmp_path, O):
    sc = make_scene("cfg3", n=24, size=40, pose="rot", shade=1)
    sc.steps, sc.sample_rate = 0, 2.5
    raw = sc.tf_vg
    sc.tf_vg = O.copy_scale(raw, 2.5)                    # renderVolume's copyScale(rate/gamma)
    p, out = _run(tmp_path, sc, 3, 2.5, raw)
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, np.float32).reshape(sc.height, sc.width, 4)
    ref = sc.render()
    assert ref[..., 3].max() > 0.05 and np.abs(got - ref).max() <= 1e-4


@pytest.mark.gpu
def test_cfg5_through_the_renderer_slot(tmp_path, O):
    """BASELINE config 5 through HipVolumeRenderable::init()/draw(): two merged fields (dmode V2G), a
    transfer-function table of 16 sheets (gluvv.tf.ptexsz[2] > 1: the dense 3-D table), gluvv.pert
    switched on -- the adapter makes createNoiseTex's texture itself (srand(1), libc rand) -- and the
    cube-map Phong.  The CPU checker gets the same state, its noise from its own rand() clone."""
    import _scenes as S
    nx, ny, nz = 40, 36, 32
    rng = np.random.default_rng(77)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    r2 = ((x - 20) ** 2 + (y - 18) ** 2 + (z - 16) ** 2) ** .5
    f = np.stack([np.clip(255 - r2 * 11 + rng.normal(0, 3, x.shape), 0, 255),
                  np.clip(np.sin(x * .35) * np.cos(y * .3) * 90 + 120 + rng.normal(0, 3, x.shape), 0, 255)], -1).astype(np.uint8)
    merged, nrm = O.merge_addg(f)               # MetaVolume::mergeMV + addG, as initData leaves gluvv.mv
    sc = O.Scene(merged, grad=nrm)
    sc.tf_mode, sc.tf3d = 2, S.tf3d_dense()
    sc.xform = O.rotation((1, 1, 0), 30)
    sc.width = sc.height = 56
    sc.steps, sc.sample_rate, sc.shade_mode = 0, 2.5, 1
    sc.noise, sc.pert_w, sc.pert_s = O.noise_tex(32), (.05, .03, 0, 0), (.2, 2.1, 4.5, 8.7)
    p, out = _run(tmp_path, sc, 3, 2.5, sc.tf3d, extra=["dmode=4", "tfsize=16,16,16", "pert=0.05,0.03,0.2,2.1"])
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, np.float32).reshape(sc.height, sc.width, 4)
    ref = sc.render()
    assert ref[..., 3].max() > 0.05 and np.abs(got - ref).max() <= 1e-4
    # and it is the perturbation that was rendered: the unperturbed frame differs visibly
    sc.noise = None
    assert np.abs(sc.render() - ref).max() > 1e-2


@pytest.mark.gpu
def test_nv20_platform_selects_the_register_combiner_phong(tmp_path, O):
    """gluvv.plat = GPNV20: the renderer Simian would have picked is NV20VolRen3D (gluvv.cpp:141-199),
    whose Phong is the register-combiner one; third-axis table deptex2 forwarded as well."""
    sc = make_scene("cfg4", n=24, size=40, pose="rot", shade=2)
    sc.steps, sc.sample_rate = 0, 2.5
    raw = sc.tf_vg
    sc.tf_vg = O.copy_scale(raw, 2.5)
    d2 = tmp_path / "deptex2.rgba"
    sc.tf_h.tofile(d2)
    p, out = _run(tmp_path, sc, 3, 2.5, raw, extra=["plat=5", "deptex2=%s" % d2])
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, np.float32).reshape(sc.height, sc.width, 4)
    ref = sc.render()
    assert ref[..., 3].max() > 0.05 and np.abs(got - ref).max() <= 1e-4
    sc.shade_mode = 1
    assert np.abs(sc.render() - ref).max() > 1e-3   # (the two Phong variants are told apart)


@pytest.mark.gpu
def test_shadow_check_box_through_the_renderer_slot(tmp_path, O):
    """gluvv.light.shadow = 1 with the good sampling rate in force: the adapter hands buffsz and gShadowQual
    to smk_set_shadow (R8kVolRen3D::setupPBuff's choice, R8kVolRen3D.cpp:1114-1123) and the frame is the
    half-angle-slicing one"""
    sc = make_scene("cfg3", n=24, size=40, pose="rot", shade=1)
    sc.steps, sc.sample_rate = 0, 2.5
    raw = sc.tf_vg
    sc.tf_vg = O.copy_scale(raw, 2.5)
    sc.light_pos = (3, 4, -3)
    sc.shadow = (96, 0.5)
    p, out = _run(tmp_path, sc, 3, 2.5, raw, extra=["light=3,4,-3", "shadow=96,0.5"])
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, np.float32).reshape(sc.height, sc.width, 4)
    ref, _ = sc.render_shadow()
    assert ref[..., 3].max() > 0.05 and np.abs(got - ref).max() <= 1e-4
    assert np.abs(sc.render() - ref).max() > 1e-2   # (not the unshadowed frame)


@pytest.mark.gpu
def test_dataset_loaded_from_trex_files_renders_like_the_checker(tmp_path, O):
    """The whole load path of `gluvv data.trex`: a big-endian float data set in two pre-bricked raw
    files + its .trex description -> VolumeFiles (parse, read, byte-swap, min/max quantise) ->
    the adapter's init()/draw().  The CPU checker renders the same quantised voxels."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import volume_files as VF
    rng = np.random.default_rng(4)
    nx, ny, nz = 24, 20, 16
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    native = (np.sin(x * .4) * np.cos(y * .3) * 40 + z * 3 + rng.normal(0, 2, x.shape)).astype(np.float32)
    q = VF.quantize(native)                      # one min/max per FILE in the reference: brick by brick
    base = tmp_path / "met"
    halves = [native[:, :, :nx // 2], native[:, :, nx // 2:]]
    for b, h in enumerate(halves):
        np.ascontiguousarray(h).astype(">f4").tofile("%s.0000.%02d" % (base, b))
    trex = tmp_path / "met.trex"
    trex.write_text("Data Set Name: met\nData Set Files: %s\nNumber of Time Steps: 1, 0, 0\nData Type: float\nEndian: big\n"
                    "Volume Size int: %d, %d, %d\nVolume Size float: 1, %.9g, %.9g\nNumber of Sub Volumes: 2\n"
                    "SubVolume {\n Size int: %d, %d, %d\n Size float: .5, %.9g, %.9g\n Pos int: 0, 0, 0\n Pos float: 0, 0, 0\n}\n"
                    "SubVolume {\n Size int: %d, %d, %d\n Size float: .5, %.9g, %.9g\n Pos int: %d, 0, 0\n Pos float: .5, 0, 0\n}\n"
                    % (base, nx, ny, nz, ny / nx, nz / nx, nx // 2, ny, nz, ny / nx, nz / nx,
                       nx // 2, ny, nz, ny / nx, nz / nx, nx // 2))
    whole = np.concatenate([VF.quantize(h) for h in halves], axis=2)
    assert whole.shape == q.shape
    sc = O.Scene(np.ascontiguousarray(whole)[..., None])
    sc.tf_mode = 0
    sc.xform = O.rotation((1, 1, 0), 30)
    sc.width = sc.height = 40
    sc.steps, sc.sample_rate = 0, 1.5
    t = O.tlut("default", 256)
    t[:, 3] = (0.1 * np.arange(256) / 255).astype(np.float32)
    sc.tlut = O.tlut_scale_alpha(t, 1.0, 1.5)
    out = tmp_path / "frame.f32"
    cmd = [EXE, str(trex), "0", "0", "0", "1", "-", "-", str(sc.width), str(sc.height), repr(1.5), "1"] + \
          [repr(float(v)) for v in sc.xform] + [str(out)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, np.float32).reshape(sc.height, sc.width, 4)
    ref = sc.render()
    assert ref[..., 3].max() > 0.05 and np.abs(got - ref).max() <= 1e-4


@pytest.mark.gpu
def test_inner_interface_sub_box_and_slice_quad(tmp_path, O):
    """The rest of VolumeRenderer's inner interface (VolumeRenderer.h:103-123) through the adapter:
    renderVolume(rate, mv, xext, yext, zext) draws the axis-aligned sub-box only -- against the CPU checker's region render --
    and renderSlice(quad, alpha) blends one volume-textured quad over that frame -- against a float64 restatement of
    render3dSliceEXT (oracle/gl_slices.py; pixels on the quad's outline may fall either way)."""
    import gl_slices
    sc = make_scene("cfg1", n=32, size=56, pose="rot")
    sc.steps, sc.sample_rate = 0, 1.5
    t = O.tlut("default", 256)
    t[:, 3] = (0.1 * np.arange(256) / 255).astype(np.float32)
    sc.tlut = O.tlut_scale_alpha(t, 1.0, 1.5)
    g0, g1 = (8, 6, 4), (24, 20, 28)
    fs = [float(f) for f in sc.fsize]
    ext = []
    for a in range(3):
        ext += [g0[a] / sc.dims[a] * fs[a], g1[a] / sc.dims[a] * fs[a]]
    quad = [[0.1, 0.15, 0.3], [0.9, 0.1, 0.35], [0.85, 0.9, 0.6], [0.15, 0.8, 0.55]]   # (a planar-ish quad through the volume)
    # make it exactly planar: the fourth vertex from the other three
    q = np.array(quad)
    q[3] = q[0] + (q[2] - q[1])
    alpha = 0.6
    p, out = _run(tmp_path, sc, 1, 1.5, None, extra=["subbox=" + ",".join(repr(v) for v in ext)])
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, np.float32).reshape(sc.height, sc.width, 4)
    sc.region = (g0, g1)
    ref = sc.render()
    assert ref[..., 3].max() > 0.02 and np.abs(got - ref).max() <= 1e-4
    whole = make_scene("cfg1", n=32, size=56, pose="rot")
    whole.steps, whole.sample_rate, whole.tlut = 0, 1.5, sc.tlut
    assert np.abs(whole.render() - ref).max() > 1e-3        # (the sub-box IS a different frame)
    p, out = _run(tmp_path, sc, 1, 1.5, None, extra=["subbox=" + ",".join(repr(v) for v in ext),
                                                      "slice=" + ",".join(repr(float(v)) for v in [alpha] + list(q.reshape(-1)))])
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, np.float32).reshape(sc.height, sc.width, 4)
    want = ref.astype(np.float64).copy()
    edge = gl_slices.render_quad_slice(want, sc.data[..., 0].astype(np.float64) / 255.0, sc.fsize, sc.mv(), sc.frustum, sc.znear, 20.0, q, alpha)
    inner = ~edge
    assert np.abs(want - ref).max() > 0.05                   # the quad shows
    assert np.abs(got[inner] - want[inner]).max() <= 1e-4
    assert edge.sum() < 0.1 * inner.sum()
