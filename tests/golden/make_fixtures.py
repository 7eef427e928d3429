"""Regenerates the committed fixtures in tests/golden/ from the CPU checker (oracle/).

Run here (the container), never on the GPU box:  python tests/golden/make_fixtures.py
Fixtures are DATA (inputs + expected outputs); nothing from the reference tree is copied.
Because the reference has no renderer tests, the golden framebuffers are produced by this
repo's own CPU ray-marcher and pin it against regressions; the Perlin
probes are additionally checked against the reference's own genvol/perlin.c when
oracle/_ref/libperlin_ref.so is available (tests/test_perlin_ref.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
from _scenes import make_scene, tf_cfg2, tf_cfg3, tf_h  # noqa: E402


def save(name, arr):
    np.save(os.path.join(HERE, name), arr, allow_pickle=False)
    print("wrote", name, arr.shape, arr.dtype)


def main():
    # classification tables used by bench.py (data files; bench never imports the checker for
    # its GPU workload)
    d1, d2 = tf_cfg2()
    save("tf_cfg2_deptex.npy", d1)
    save("tf_cfg3_levwidget.npy", tf_cfg3())
    save("tf_h_slider05.npy", tf_h(0.5))
    save("tlut_cfg1.npy", O.tlut_volumerenderable())
    save("noise32.npy", O.noise_tex(32))

    # Perlin probes: srand(1), main's init() + first-call re-init, 64 points (SURVEY KAT 11)
    L = O.lib()
    L.orc_srand(1)
    L.orc_perlin_reset()
    L.orc_perlin_init()
    rng = np.random.default_rng(2001)
    pts = rng.uniform(0, 3, (64, 3))
    vals = np.array([[L.orc_perlin3d(*p, 2.0, 2.0, 10), L.orc_perlin3d_abs(*p, 2.0, 2.0, 10)] for p in pts])
    save("perlin_probes.npy", np.concatenate([pts, vals], axis=1))

    # golden framebuffers, tiny (SURVEY KAT 12): inputs are regenerated from seeds by _scenes
    gold = {}
    for name, kw in [
        ("cfg1", dict(kind="cfg1", n=24, size=32, steps=32, pose="rot")),
        ("cfg2_u8", dict(kind="cfg2", n=24, size=32, steps=32, pose="rot")),
        ("cfg3_f32_r8k", dict(kind="cfg3", n=24, size=32, steps=32, pose="rot", f32=True, shade=1)),
        ("cfg3_u8_nv20", dict(kind="cfg3", n=24, size=32, steps=32, pose="id", shade=2)),
        ("cfg4_f32", dict(kind="cfg4", n=24, size=32, steps=32, pose="back", f32=True, shade=1)),
        ("pert", dict(kind="cfg3", n=24, size=32, steps=32, pose="rot", shade=1, pert=True)),
    ]:
        gold[name] = make_scene(**kw).render()
    np.savez_compressed(os.path.join(HERE, "golden_frames.npz"), **gold)
    print("wrote golden_frames.npz", list(gold))
    # the 24^3 input volume of those frames, so a drift of the generators is told apart from a
    # drift of the renderer
    save("genvol_spheres24.npy", O.genvol_spheres(24, seed=1))
    save("vgh24.npy", O.make_vgh(O.genvol_spheres(24, seed=1)))


if __name__ == "__main__":
    main()
