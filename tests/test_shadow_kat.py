"""Known-answer tests of the CPU checker's half-angle-slicing shadows (oracle/smk_oracle.c:
orc_shadow_setup / orc_render_shadow).  The reference's shadow mode cannot be run (pbuffers, ATI
fragment shaders), so each case is a closed-form consequence of the cited lines of
R8kVolRen3D.cpp / LTWidgetRen.cpp.  Runs without a GPU."""
import numpy as np
import pytest


def _uniform(O, n=24, size=24, steps=30, alpha_u8=26, rgb=(204, 102, 51)):
    sc = O.Scene(np.full((n, n, n, 2), 128, np.uint8))
    tf = np.zeros((16, 16, 4), np.uint8)
    tf[..., 0], tf[..., 1], tf[..., 2], tf[..., 3] = rgb[0], rgb[1], rgb[2], alpha_u8
    sc.tf_mode, sc.tf_vg = 1, tf
    sc.width = sc.height = size
    sc.steps = steps
    sc.shadow = (64, 0.5)
    return sc


def test_light_at_the_eye_side_closed_form(O):
    """v = l: the half-way vector is the view axis, slices run away from eye and light (front to back,
    R8kVolRen3D.cpp:305-314, 1441-1449).  In a homogeneous volume slice k sees the light buffer after
    k-1 slices, L_k = a c + (1-a) L_{k-1} (LERP, :3150-3156), and contributes a c (1 - L_{k-1}) (:2928-2934)."""
    sc = _uniform(O)
    img, L = sc.render_shadow()
    c = sc.shadowcoef()
    assert c.front_to_back == 1 and c.nslices == 30 and c.LB == 32
    a = np.float32(26 / 255)
    col = np.array([204, 102, 51], np.float32) / 255
    C, Lb = np.zeros(4), np.zeros(3)
    for _ in range(30):
        src = np.append(col * (1 - Lb) * a, a)
        C = C + (1 - C[3]) * src
        Lb = a * col + (1 - a) * Lb
    h = sc.height // 2
    assert np.allclose(img[h, h], C, atol=2e-6)
    assert np.allclose(L[16, 16, :3], Lb, atol=2e-6)
    assert np.isclose(L[16, 16, 3], 1 - (1 - a) ** 30, atol=2e-6)   # alpha = sat((1-a) L.a + a), :3158-3162
    # opacity is untouched by the shadow term: same planes as the unshadowed renderer when l == v
    ref = sc.render()
    inner = (slice(6, 18), slice(6, 18))
    assert np.allclose(ref[inner][..., 3], img[inner][..., 3], atol=1e-6)
    assert (img[..., :3] <= ref[..., :3] + 1e-6).all() and img[h, h, 0] < 0.7 * ref[h, h, 0]


def test_light_behind_the_volume_blends_back_to_front(O):
    """v.l <= 0: the view direction is negated before halving (:307-311) and the slices are blended with
    GL_ONE, GL_ONE_MINUS_SRC_ALPHA (:1436-1440): the frame equals the front-to-back composite of the same
    per-slice contributions taken in the opposite order."""
    sc = _uniform(O)
    sc.light_pos = (0, 0, 5)
    img, L = sc.render_shadow()
    c = sc.shadowcoef()
    assert c.front_to_back == 0
    a = np.float32(26 / 255)
    col = np.array([204, 102, 51], np.float32) / 255
    C, Lb = np.zeros(4), np.zeros(3)
    for _ in range(30):   # slices march away from the light = towards the eye: each new one goes OVER
        src = np.append(col * (1 - Lb) * a, a)
        C = src + (1 - a) * C
        Lb = a * col + (1 - a) * Lb
    h = sc.height // 2
    assert np.allclose(img[h, h], C, atol=2e-6)
    # the slice nearest the light is the brightest and ends up at the BACK: the frame is darker than with the light in front
    front, _ = _uniform(O).render_shadow()
    assert img[h, h, 0] < front[h, h, 0]


def test_transparent_volume_leaves_no_shadow_and_black_colours_do_not_darken(O):
    sc = _uniform(O, alpha_u8=0)
    img, L = sc.render_shadow()
    assert not img.any() and not L.any()
    sc = _uniform(O, rgb=(0, 0, 0))
    img, L = sc.render_shadow()
    assert not L[..., :3].any() and L[..., 3].max() > 0.9 and not img[..., :3].any() and img[..., 3].max() > 0.9


def test_light_buffer_footprint_follows_the_light_projection(O):
    """lc = (x'/w * .85 + .5) * quality (R8kVolRen3D.cpp:1673-1674) with w = 1 + z'/|light.pos| and z' = 1 - F.q
    (LTWidgetRen::genXForm: gluLookAt from -norm(light.pos), z translation negated, pj[11] = 1/d0).  For the
    light on the view axis the unit cube's near face (F.q = +1/2) projects to |x'/w| <= .5 / (1 + .5/5)."""
    sc = _uniform(O, alpha_u8=255)
    _, L = sc.render_shadow()
    lit = L[16, :, 3] > 0
    half = 0.5 / (1 + 0.5 / 5) * 0.85 * 32     # texels from the centre
    assert abs(lit.sum() / 2 - half) <= 1.0
    assert lit[16 - int(half) + 1] and not lit[16 - int(half) - 2]


def test_oblique_light_casts_the_shadow_to_the_far_side(O):
    """an opaque blob in a thin medium: with the light up and to the left the medium behind-right of the blob is
    darker than the medium on the light's side"""
    n = 32
    z, y, x = np.meshgrid(*(np.arange(n),) * 3, indexing="ij")
    blob = ((x - 10) ** 2 + (y - 20) ** 2 + (z - 16) ** 2) < 25
    data = np.zeros((n, n, n, 2), np.uint8)
    data[..., 0] = np.where(blob, 255, 60)
    data[..., 1] = 128
    sc = O.Scene(data)
    tf = np.zeros((16, 16, 4), np.uint8)
    tf[..., :3] = 255
    tf[..., 3] = 6
    tf[:, 12:, 3] = 255                       # the blob's value is opaque
    sc.tf_mode, sc.tf_vg = 1, tf
    sc.width = sc.height = 32
    sc.steps = 48
    sc.shadow = (128, 0.5)
    sc.light_pos = (5, 5, -3)                 # gluvv's camera sits at -z looking at +z; +x is image left (gluLookAt from -z)
    img, L = sc.render_shadow()
    plain = sc.render()
    ratio = img[..., 0] / np.maximum(plain[..., 0], 1e-6)
    # image columns: world +x maps to decreasing i; the blob sits at world x < centre => image right half
    dark = np.unravel_index(np.argmin(np.where(plain[..., 3] > 0.2, ratio, 9)), ratio.shape)
    assert ratio[dark] < 0.5
    assert L[..., 0].max() > 0.95
