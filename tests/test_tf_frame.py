"""SURVEY 8(f3), second half: the transfer-function window's brush / paint layer and the dual-domain data
probe as headless host functions (simian-spacemonkey_amd/host/TransferFunctions.{h,cpp}: TFFrame,
probe_sample, place_brush, probe_world_to_volume), driven through tests/host/tf_main `session` and
compared with the restatement in oracle/tf_frame.py: whole sessions of random strokes give
byte-identical tables; the probe's values and cell are equal to the last bit; closed-form known
answers for the probe and for the widget-tip -> volume mapping.  No GPU involved."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host", "tf_main")


def run_session(tmp_path, lines):
    script = tmp_path / "session.txt"
    script.write_text("\n".join(lines) + "\n")
    p = subprocess.run([EXE, "session", str(script)], capture_output=True, text=True)
    assert p.returncode == 0, (p.returncode, p.stderr)
    return [ln.split() for ln in p.stdout.splitlines()]


def r9(x):
    return "%.9g" % float(x)


def test_probe_known_answers(tmp_path, O):
    """a volume whose first byte ramps with x and second with y: the probe's transfer-function position IS
    (x, y) of the voxel grid (TFWidgetRen1.cpp:372-470, triLerpV3 :600-621); outside the volume nothing is sampled"""
    import tf_frame as TF
    sx, sy, sz = 20, 16, 12
    z, y, x = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
    vol = np.stack([x * 10, y * 12, (z * 15) % 256], -1).astype(np.uint8)
    f = tmp_path / "vol.u8"
    vol.tofile(f)
    pts = [(0.5, 0.5, 0.5), (0.27, 0.81, 0.33), (0.02, 0.5, 0.5), (0.5, 0.5, 0.97)]
    out = run_session(tmp_path, ["frame 64 64 1 9 1", "volume %s %d %d %d 3" % (f, sx, sy, sz), "brush 1 0"] +
                      ["probe %s %s %s 0.5" % tuple(r9(v) for v in p) for p in pts])
    for p, got in zip(pts, out):
        inside, cell, corners, val = TF.probe_sample(vol, 9, p)
        assert int(got[0]) == int(inside) and tuple(int(v) for v in got[1:4]) == cell
        assert [np.float32(v) for v in got[4:7]] == [v for v in val]      # equal to the last bit
    # closed form at (0.27, 0.81): voxel coordinate = vpos * size, value = coordinate * step / 255
    v = [float(g) for g in out[1][4:7]]
    assert abs(v[0] - 0.27 * sx * 10 / 255) < 1e-5 and abs(v[1] - 0.81 * sy * 12 / 255) < 1e-5
    assert out[2][0] == "0" and out[3][0] == "0"          # cells 0 and size-1 are outside (1 .. size-2 only)


def test_widget_tip_to_volume_coordinates(tmp_path):
    """DPWidgetRen::update_pos (DPWidgetRen.cpp:278-317): vpos = (T(trans) R S(scale) T(-size/2) S(size))^-1 pos"""
    ident = "1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1"
    out = run_session(tmp_path, ["world 0 0 0  0 0 0  1  1 1 1  " + ident,          # the volume's centre
                                 "world 0.5 0.25 -0.5  0 0 0  1  1 0.5 1  " + ident,  # a corner of a half-height volume
                                 "world 0.3 0.1 0.2  0.1 -0.2 0.05  2  1 1 1  " + ident])
    assert np.allclose([float(v) for v in out[0]], [.5, .5, .5], atol=1e-6)
    assert np.allclose([float(v) for v in out[1]], [1.0, 1.0, 0.0], atol=1e-6)
    assert np.allclose([float(v) for v in out[2]], [(0.3 - 0.1) / 2 + .5, (0.1 + 0.2) / 2 + .5, (0.2 - 0.05) / 2 + .5], atol=1e-6)
    # a rotation: the mapping inverts what the renderer's model matrix does to a volume-space point
    c, s = np.cos(0.7), np.sin(0.7)
    R = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    vp = np.array([0.2, 0.7, 0.4, 1])
    T = np.eye(4); T[:3, 3] = (0.1, 0.2, -0.3)
    C_ = np.eye(4); C_[:3, 3] = (-.5, -.4, -.5)
    S = np.diag([1, .8, 1, 1])
    pos = T @ R @ C_ @ S @ vp
    out = run_session(tmp_path, ["world %s %s %s  0.1 0.2 -0.3  1  1 0.8 1  %s" % (r9(pos[0]), r9(pos[1]), r9(pos[2]),
                                                                                  " ".join(r9(v) for v in R.T.reshape(-1)))])
    assert np.allclose([float(v) for v in out[0]], vp[:3], atol=1e-5)


@pytest.mark.parametrize("dmode,sh,seed", [(9, 1, 3), (9, 4, 4), (0, 1, 5), (4, 1, 6)])
def test_sessions_of_random_strokes(tmp_path, O, dmode, sh, seed):
    """what a user does in the TF window: widgets created, the probe dragged through the volume with each
    brush in turn, strokes painted, brushes dropped as widgets, the paint layer cleared, the table
    regenerated in between (TFWidgetRen1.cpp:194-242) -- tables byte for byte as the restatement's"""
    import tf_frame as TF
    rng = np.random.default_rng(seed)
    sx, sy, sz, ne = 24, 20, 16, (1 if dmode == 0 else 3)
    vol = rng.integers(0, 256, (sz, sy, sx, ne), dtype=np.uint8)
    z, y, x = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
    vol[..., 0] = (x * 9 + rng.integers(0, 12, x.shape)).astype(np.uint8)     # smooth-ish first channel
    f = tmp_path / "vol.u8"
    vol.tofile(f)
    sv, sg = 96, 64
    fr = TF.Frame(sv, sg, sh, dmode, faux=bool(seed & 1))
    lines = ["frame %d %d %d %d %d" % (sv, sg, sh, dmode, seed & 1), "volume %s %d %d %d %d" % (f, sx, sy, sz, ne)]
    expected = []
    for n in range(40):
        op = rng.integers(0, 10)
        if op < 2:       # a widget of a random shape appears
            kind = int(rng.integers(0, 4))
            bx, by = float(rng.uniform(.2, .8)), float(rng.uniform(0, .3))
            ly = float(rng.uniform(by + .2, 1.0))
            lx, rx = float(rng.uniform(0.02, bx - .05)), float(rng.uniform(bx + .05, .98))
            hsl = tuple(float(v) for v in (rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(.2, .8)))
            alpha, be = float(rng.uniform(.1, 1)), float(rng.uniform(.2, 1))
            # (parameters travel as text: both sides see the float32 the driver parses)
            vals = [np.float32(v) for v in (bx, by, lx, ly, rx, ly, -10.0, -10.0, *hsl, alpha, be)]
            lines.append("widget %d " % kind + " ".join(r9(v) for v in vals))
            v = [float(x) for x in vals]
            fr.widgets.insert(0, TF.Widget(TF.KINDS[kind], b=(v[0], v[1]), l=(v[2], v[3]), r=(v[4], v[5]), tw=-10.0, th=-10.0,
                                            hsl=(v[8], v[9], v[10]), alpha=v[11], be=v[12]))
        elif op < 6:     # the probe moves with some brush switched on
            kind = int(rng.integers(1, 6))
            on = int(rng.integers(0, 4) > 0)
            vpos = [np.float32(v) for v in rng.uniform(-0.05, 1.05, 3)]
            slider = np.float32(rng.uniform(0, 1))
            lines += ["brush %d %d" % (kind, on), "probe %s %s %s %s" % (r9(vpos[0]), r9(vpos[1]), r9(vpos[2]), r9(slider))]
            fr.brush_kind, fr.brushon = kind, bool(on)
            inside, cell, corners, val = TF.probe_sample(vol, dmode, vpos)
            if fr.brushon or not inside:
                TF.place_brush(fr, inside, corners, val, slider)
        elif op == 6:
            lines.append("paint")
            fr.paint()
        elif op == 7:
            lines.append("drop")
            fr.drop()
        elif op == 8 and n % 3 == 0:
            lines.append("clear")
            fr.clear_paint()
        else:
            out = tmp_path / ("tf%02d.tex" % n)
            lines.append("regen %s" % out)
            expected.append((out, fr.regenerate()))
    out = tmp_path / "final.tex"
    lines.append("regen %s" % out)
    expected.append((out, fr.regenerate()))
    run_session(tmp_path, lines)
    painted = 0
    for path, ref in expected:
        got = np.fromfile(path, np.uint8).reshape(ref.shape)
        assert np.array_equal(got, ref), "%s differs in %d bytes" % (path.name, int((got != ref).sum()))
        painted = max(painted, int(np.count_nonzero(ref[..., 3])))
    assert painted > 500
