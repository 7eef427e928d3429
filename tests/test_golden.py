"""The CPU checker against the committed fixtures (tests/golden/, written by make_fixtures.py):
pins the checker -- and, on the GPU box, everything compared with it -- against drift."""
import os

import numpy as np
import pytest

from _scenes import make_scene

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "cfg1": dict(kind="cfg1", n=24, size=32, steps=32, pose="rot"),
    "cfg2_u8": dict(kind="cfg2", n=24, size=32, steps=32, pose="rot"),
    "cfg3_f32_r8k": dict(kind="cfg3", n=24, size=32, steps=32, pose="rot", f32=True, shade=1),
    "cfg3_u8_nv20": dict(kind="cfg3", n=24, size=32, steps=32, pose="id", shade=2),
    "cfg4_f32": dict(kind="cfg4", n=24, size=32, steps=32, pose="back", f32=True, shade=1),
    "pert": dict(kind="cfg3", n=24, size=32, steps=32, pose="rot", shade=1, pert=True),
}


def test_generators_match_fixtures(O):
    assert np.array_equal(O.genvol_spheres(24, seed=1), np.load(os.path.join(G, "genvol_spheres24.npy")))
    assert np.array_equal(O.make_vgh(O.genvol_spheres(24, seed=1)), np.load(os.path.join(G, "vgh24.npy")))
    assert np.array_equal(O.noise_tex(32), np.load(os.path.join(G, "noise32.npy")))
    assert np.array_equal(O.tlut_volumerenderable(), np.load(os.path.join(G, "tlut_cfg1.npy")))


def test_tf_tables_match_fixtures(O):
    from _scenes import tf_cfg2, tf_cfg3, tf_h
    assert np.array_equal(tf_cfg2()[0], np.load(os.path.join(G, "tf_cfg2_deptex.npy")))
    assert np.array_equal(tf_cfg3(), np.load(os.path.join(G, "tf_cfg3_levwidget.npy")))
    assert np.array_equal(tf_h(0.5), np.load(os.path.join(G, "tf_h_slider05.npy")))


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_frames(O, name):
    gold = np.load(os.path.join(G, "golden_frames.npz"))[name]
    img = make_scene(**CASES[name]).render()
    assert gold[..., 3].max() > 0.05
    assert np.abs(img - gold).max() <= 1e-6     # same code, same machine class: only libm noise


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_frames_gpu(gpu_renderer_factory, name):
    """the HIP product against the COMMITTED frames (not a freshly computed checker frame)"""
    from _scenes import push_scene
    gold = np.load(os.path.join(G, "golden_frames.npz"))[name]
    r = gpu_renderer_factory()
    try:
        push_scene(r, make_scene(**CASES[name]))
        assert np.abs(r.render() - gold).max() <= 1e-4
    finally:
        r.close()
