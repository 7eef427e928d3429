"""The ray-marching CPU checker against a structurally different restatement of the same renderer: the
OpenGL slice pipeline of VolumeRenderer::render3DVA emulated step by step in float64 (oracle/gl_slices.py:
plane set from the farthest vertex, polygons from edge intersections, rasterisation with perspective-correct
texture coordinates, post-filter colour table, back-to-front framebuffer blend).  What the two share is only
the reading of the per-texel formulas; plane placement, plane count, the inside rule and the sample positions
are derived twice.  Runs without a GPU."""
import numpy as np
import pytest

from _scenes import make_scene


@pytest.mark.parametrize("pose,steps,rate", [("id", 14, 0.0), ("rot", 16, 0.0), ("side", 0, 0.6), ("back", 12, 0.0)])
def test_ray_marcher_equals_the_slice_pipeline(O, pose, steps, rate):
    import gl_slices
    sc = make_scene("cfg1", n=16, size=20, steps=steps, pose=pose)
    sc.sample_rate = rate
    ref = sc.render(blend=1)                       # VolumeRenderer's own order: far plane first, GL_ONE, GL_ONE_MINUS_SRC_ALPHA
    got = gl_slices.render_scalar_slices(sc.data[..., 0], sc.fsize, sc.mv(), sc.frustum, sc.znear, 20.0,
                                         sc.width, sc.height, sc.tlut, sample_rate=rate or None, steps=steps or None)
    assert ref[..., 3].max() > 0.05
    d = np.abs(got - ref).max(axis=2)
    # float64 slices vs fp32 fma chains: a pixel centre within ~1e-6 of a polygon edge, or a sample within ~1e-6 of a
    # colour-table bin boundary, may fall on the other side; everything else agrees to rounding
    assert (d <= 2e-5).mean() >= 0.97, f"{(d > 2e-5).sum()} of {d.size} pixels differ"
    assert d.max() <= 0.12                          # (never by more than one slice's contribution: alpha ramp tops at .1)
    assert np.abs(got[..., 3].sum() - ref[..., 3].sum()) <= 0.01 * ref[..., 3].sum()
