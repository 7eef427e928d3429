"""the classification widgets' rasterisers as headless host functions
(simian-spacemonkey_amd/host/TransferFunctions.cpp, driven through tests/host/tf_main): every
LevWidget shape -- triangle, ellipse, 1-D style, default style -- painted over empty and over
already painted tables, one and four sheets, with and without faux shading, and the third-axis
ramp, byte for byte against the CPU restatement in oracle/ (itself pinned by the known-answer
tests of tests/test_oracle_kat.py).  No GPU involved."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host", "tf_main")
KINDS = ["triangle", "ellipse", "1d", "default"]


def paint(tmp_path, kind, tex, faux, b, l, r, tw, th, hsl, alpha, be):
    sh = 1 if tex.ndim == 3 else tex.shape[0]
    sg, sv = tex.shape[-3], tex.shape[-2]
    src = tmp_path / "in.tex"
    tex.tofile(src)
    out = tmp_path / "out.tex"
    cmd = [EXE, "lev", KINDS.index(kind), int(faux), sv, sg, sh, *b, *l, *r, tw, th, *hsl, alpha, be, src, out]
    p = subprocess.run([str(c) if not isinstance(c, float) else repr(c) for c in cmd], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return np.fromfile(out, np.uint8).reshape(tex.shape)


def test_driver_is_built():
    assert os.path.exists(EXE), "build with __graft_entry__.build()"


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("faux", [False, True])
def test_default_widget_on_an_empty_table(tmp_path, O, kind, faux):
    tex = np.zeros((256, 256, 4), np.uint8)
    args = dict(b=(.5, 0.0), l=(.3, .7), r=(.7, .7), tw=-10.0, th=-10.0, hsl=(0.0, 1.0, .5), alpha=.5, be=1.0)
    got = paint(tmp_path, kind, tex, faux, **args)
    ref = O.lev_rasterize(O.lev_widget(kind, faux=faux, **args), tex.copy())
    assert ref[..., 3].max() > 50 and np.count_nonzero(ref[..., 3]) > 3000
    assert np.array_equal(got, ref)


def test_random_widgets_layered_over_each_other(tmp_path, O):
    """what a session does: several widgets of every shape painted one after the other into the
    same four-sheet table (colour = alpha-weighted average with what is there, alpha max / over)"""
    rng = np.random.default_rng(12)
    tex_p = np.zeros((4, 64, 96, 4), np.uint8)
    tex_o = tex_p.copy()
    for n in range(16):
        kind = KINDS[n % 4]
        bx = float(rng.uniform(.2, .8))
        by = float(rng.uniform(0, .3))
        ly = float(rng.uniform(by + .2, 1.0))
        lx = float(rng.uniform(0.02, bx - .05))
        rx = float(rng.uniform(bx + .05, .98))
        args = dict(b=(bx, by), l=(lx, ly), r=(rx, ly),
                    tw=-10.0 if n % 3 else float(rng.uniform(lx, rx)), th=-10.0 if n % 2 else float(rng.uniform(by, ly)),
                    hsl=(float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(.2, .8))),
                    alpha=float(rng.uniform(.1, 1)), be=float(rng.uniform(.2, 1)))
        faux = bool(n & 1)
        tex_p = paint(tmp_path, kind, tex_p, faux, **args)
        tex_o = O.lev_rasterize(O.lev_widget(kind, faux=faux, **args), tex_o)
        assert np.array_equal(tex_p, tex_o), "widget %d (%s)" % (n, kind)
    assert np.count_nonzero(tex_o[..., 3]) > 4000


def test_hue_circle_of_the_default_shape(tmp_path, O):
    """the default widget walks once around the hue circle across its width, backwards: red (hue
    just below 1), magenta, blue, cyan half way, green, yellow, red again"""
    tex = np.zeros((256, 256, 4), np.uint8)
    got = paint(tmp_path, "default", tex, False, (.5, 0.0), (.1, .9), (.9, .9), -10.0, -10.0, (0.0, 1.0, .5), 1.0, 1.0)
    row = got[200]
    painted = np.nonzero(row[:, 3])[0]
    assert painted.min() == 25 and painted.max() == 229            # (int)(.1*256) .. (int)(.9*256)-1
    first, mid, last = row[painted[2]], row[painted[len(painted) // 2]], row[painted[-3]]
    assert first[0] > 200 and first[1] < 60 and first[2] < 80      # hue just below 1: red, a touch of blue
    assert mid[1] > 200 and mid[0] < 60                            # green / cyan half way
    assert last[0] > 200 and last[1] < 80 and last[2] < 80          # back at red


@pytest.mark.parametrize("slider", [0.0, 0.5, 0.93, 1.0])
def test_third_axis_ramp(tmp_path, O, slider):
    out = tmp_path / "vgh.tex"
    p = subprocess.run([EXE, "vgh", "256", "4", repr(slider), str(out)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = np.fromfile(out, np.uint8).reshape(4, 256, 4)
    ref = O.rasterize_vgh(np.zeros((4, 256, 4), np.uint8), slider)
    assert np.array_equal(got, ref)
    assert got[0, 85, 3] == 0 and got[0, 171:, 3].max() == 0       # column 85 itself and the top third are never written
    assert got[0, 84, 3] >= got[0, 0, 3] and got[0, 86, 3] >= got[0, 170, 3]   # rises towards the zero crossing, falls after it
