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


def _run(tmp_path, sc, shade, rate, deptex):
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
           str(sc.width), str(sc.height), repr(rate), str(shade)] + [repr(float(v)) for v in sc.xform] + [str(out)]
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
