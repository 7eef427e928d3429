"""A second, structurally different restatement of the scalar renderer: the OpenGL slice pipeline itself.

TEST INFRASTRUCTURE ONLY (imported by tests/ alone).  oracle/smk_oracle.c marches RAYS and places samples with
fp32 fma chains that the HIP kernels share; a misreading of the reference's geometry would be shared too.
This module instead follows VolumeRenderer::render3DVA (VolumeRenderer.cpp:507-741) step by step, in float64
numpy, the way the 2001 pipeline ran it:

  * slice planes: normal sn = the view axis taken to model space and normalised (:539-548), first plane point =
    the box vertex farthest from the eye, advanced by del = sn * dis BEFORE every slice (:596-606),
    samples = (int)(dist / dis) with dist = the box's view-depth extent (:551-562, :598);
  * each plane is cut against the 12 box edges (`intersect`, :745-780) -> a convex polygon with texture
    coordinates in [0,1]^3 (edge to edge);
  * the polygon's vertices go through modelview, glFrustum and the viewport (gluvv.cpp:531-552); it is
    rasterised at pixel centres as a triangle fan with perspective-correct interpolation of the texture
    coordinates (what GL does with glTexCoord3fv per vertex);
  * GL_LINEAR / clamp-to-edge 3-D texture fetch, the SGI texture colour table (post-filter lookup, nearest
    entry, TLUT.cpp:65-80 premultiplied), and the framebuffer blend GL_ONE, GL_ONE_MINUS_SRC_ALPHA (:589-590).

tests/test_gl_slices.py compares its frames with the ray-marching checker's.
"""
import numpy as np

# box vertices as VolumeRenderer::renderVolume builds them (x fastest: bit 0 = x, 1 = y, 2 = z) and the 12 edges
# of render3DVA's intersect() calls (:610-645)
EDGES = [(0, 1), (0, 2), (1, 3), (4, 0), (1, 5), (2, 3), (4, 5), (4, 6), (5, 7), (6, 7), (2, 6), (3, 7)]


def _frustum(l, r, b, t, n, f):
    return np.array([[2 * n / (r - l), 0, (r + l) / (r - l), 0],
                     [0, 2 * n / (t - b), (t + b) / (t - b), 0],
                     [0, 0, -(f + n) / (f - n), -2 * f * n / (f - n)],
                     [0, 0, -1, 0]], np.float64)


def _tex3d_linear(vol, s, t, r):
    """GL_LINEAR, GL_CLAMP_TO_EDGE; vol[z][y][x] scalar in [0,1]; s,t,r texture coordinates"""
    nz, ny, nx = vol.shape
    out = 0.0
    cs = []
    for c, n in ((s, nx), (t, ny), (r, nz)):
        u = min(max(c * n - 0.5, 0.0), n - 1.0)
        i0 = int(np.floor(u))
        i0 = min(i0, max(n - 2, 0))
        cs.append((i0, min(i0 + 1, n - 1), u - i0))
    (x0, x1, fx), (y0, y1, fy), (z0, z1, fz) = cs
    for zi, wz in ((z0, 1 - fz), (z1, fz)):
        for yi, wy in ((y0, 1 - fy), (y1, fy)):
            for xi, wx in ((x0, 1 - fx), (x1, fx)):
                out += wz * wy * wx * vol[zi, yi, xi]
    return out


def render_scalar_slices(vol_u8, fsize, mv, frustum, znear, zfar, width, height, tlut, sample_rate=None, steps=None):
    """vol_u8 [z][y][x]; mv column-major 16 (GL); tlut [size][4] straight colours.  Returns [height][width][4]
    premultiplied float64, row 0 = bottom."""
    vol = vol_u8.astype(np.float64) / 255.0
    M = np.array(mv, np.float64).reshape(4, 4).T          # column-major -> matrix
    Minv = np.linalg.inv(M)
    fx, fy, fz = (float(v) for v in fsize)
    vo = np.array([[(i & 1) * fx, ((i >> 1) & 1) * fy, ((i >> 2) & 1) * fz] for i in range(8)])
    tx = np.array([[(i & 1), ((i >> 1) & 1), ((i >> 2) & 1)] for i in range(8)], np.float64)
    rv = (M[:3, :3] @ vo.T).T + M[:3, 3]
    minvert = int(np.argmin(rv[:, 2]))                    # farthest from the eye (the eye looks down -z)
    zmin, zmax = rv[:, 2].min(), rv[:, 2].max()
    sn = Minv[:3, :3] @ np.array([0.0, 0.0, 1.0])         # translateV3(sn, mvinv, vpn): the view axis in model space
    sn /= np.linalg.norm(sn)
    # distance to be sampled: |mvinv (0,0,zmax) - mvinv (0,0,zmin)| (:551-562)
    dist = np.linalg.norm(Minv[:3, :3] @ np.array([0.0, 0.0, zmax - zmin]))
    if steps:
        dis = dist / steps
        samples = steps
    else:
        dis = np.float32(fx) / (np.float32(vol.shape[2]) * np.float32(sample_rate))   # float, as :595
        samples = int(dist / dis)
    sp = vo[minvert].copy()
    P = _frustum(frustum[0], frustum[1], frustum[2], frustum[3], znear, zfar)
    size = tlut.shape[0]
    table = np.concatenate([tlut[:, :3] * tlut[:, 3:4], tlut[:, 3:4]], axis=1).astype(np.float64)   # TLUT.cpp:65-71
    img = np.zeros((height, width, 4), np.float64)
    for _ in range(samples):
        sp = sp + sn * dis
        poly, tcs = [], []
        for a, b in EDGES:                                 # `intersect`: t = sn.(sp - p0) / sn.(p1 - p0), kept if 0 <= t <= 1
            den = sn @ (vo[b] - vo[a])
            if den == 0:
                continue
            t = (sn @ (sp - vo[a])) / den
            if 0 <= t <= 1:
                poly.append(vo[a] + t * (vo[b] - vo[a]))
                tcs.append(tx[a] + t * (tx[b] - tx[a]))
        if len(poly) < 3:
            continue
        poly, tcs = np.array(poly), np.array(tcs)
        eye = (M[:3, :3] @ poly.T).T + M[:3, 3]
        # sort around the centre (the reference's angle sort, :668-720; any consistent convex order rasterises the same)
        cen = eye[:, :2].mean(axis=0)
        order = np.argsort(np.arctan2(eye[:, 1] - cen[1], eye[:, 0] - cen[0]))
        eye, tcs = eye[order], tcs[order]
        clip = (P @ np.concatenate([eye, np.ones((len(eye), 1))], axis=1).T).T
        w = clip[:, 3]
        ndc = clip[:, :2] / w[:, None]
        win = np.stack([(ndc[:, 0] * .5 + .5) * width, (ndc[:, 1] * .5 + .5) * height], axis=1)
        x0, x1 = int(np.floor(win[:, 0].min())), int(np.ceil(win[:, 0].max()))
        y0, y1 = int(np.floor(win[:, 1].min())), int(np.ceil(win[:, 1].max()))
        for j in range(max(y0, 0), min(y1 + 1, height)):
            for i in range(max(x0, 0), min(x1 + 1, width)):
                p = np.array([i + .5, j + .5])
                # triangle fan (v0, vk, vk+1), perspective-correct barycentric interpolation of the texture coordinates
                for k in range(1, len(win) - 1):
                    a, b, c = win[0], win[k], win[k + 1]
                    den = (b[1] - c[1]) * (a[0] - c[0]) + (c[0] - b[0]) * (a[1] - c[1])
                    if den == 0:
                        continue
                    l0 = ((b[1] - c[1]) * (p[0] - c[0]) + (c[0] - b[0]) * (p[1] - c[1])) / den
                    l1 = ((c[1] - a[1]) * (p[0] - c[0]) + (a[0] - c[0]) * (p[1] - c[1])) / den
                    l2 = 1 - l0 - l1
                    if l0 < 0 or l1 < 0 or l2 < 0:
                        continue
                    iw = np.array([l0 / w[0], l1 / w[k], l2 / w[k + 1]])
                    tc = (iw[0] * tcs[0] + iw[1] * tcs[k] + iw[2] * tcs[k + 1]) / iw.sum()
                    v = _tex3d_linear(vol, tc[0], tc[1], tc[2])
                    idx = min(max(int(np.floor(v * (size - 1) + 0.5)), 0), size - 1)
                    src = table[idx]
                    img[j, i] = src + (1 - src[3]) * img[j, i]
                    break
    return img


def render_quad_slice(img, vol01, fsize, mv, frustum, znear, zfar, quad, alpha):
    """VolumeRenderer::render3dSliceEXT (VolumeRenderer.cpp:762-807) in float64: ONE quad in model space, vertices issued
    1, 0, 2, 3, textured with the scalar volume (GL_INTENSITY8: (I, I, I, I), no colour table), modulated by
    glColor4f(1, 1, 1, alpha), blended GL_ONE, GL_ONE_MINUS_SRC_ALPHA into img [H][W][4] (in place; returns the pixels
    the quad's EDGE passes within a quarter pixel of, which a different rasteriser may decide the other way)."""
    height, width = img.shape[:2]
    M = np.array(mv, np.float64).reshape(4, 4).T
    P = _frustum(frustum[0], frustum[1], frustum[2], frustum[3], znear, zfar)
    q = np.array(quad, np.float64)[[1, 0, 2, 3]]
    tcs = q / np.array([float(f) for f in fsize])
    eye = (M[:3, :3] @ q.T).T + M[:3, 3]
    clip = (P @ np.concatenate([eye, np.ones((4, 1))], axis=1).T).T
    w = clip[:, 3]
    ndc = clip[:, :2] / w[:, None]
    win = np.stack([(ndc[:, 0] * .5 + .5) * width, (ndc[:, 1] * .5 + .5) * height], axis=1)
    edge = np.zeros((height, width), bool)
    x0, x1 = int(np.floor(win[:, 0].min())), int(np.ceil(win[:, 0].max()))
    y0, y1 = int(np.floor(win[:, 1].min())), int(np.ceil(win[:, 1].max()))
    for j in range(max(y0, 0), min(y1 + 1, height)):
        for i in range(max(x0, 0), min(x1 + 1, width)):
            p = np.array([i + .5, j + .5])
            for k in (1, 2):
                a, b, c = win[0], win[k], win[k + 1]
                den = (b[1] - c[1]) * (a[0] - c[0]) + (c[0] - b[0]) * (a[1] - c[1])
                if den == 0:
                    continue
                l0 = ((b[1] - c[1]) * (p[0] - c[0]) + (c[0] - b[0]) * (p[1] - c[1])) / den
                l1 = ((c[1] - a[1]) * (p[0] - c[0]) + (a[0] - c[0]) * (p[1] - c[1])) / den
                l2 = 1 - l0 - l1
                scale = np.sqrt(abs(den))                    # ~ the triangle's size in pixels
                outer = (min(l0, l2) if k == 1 else min(l1, l0)) * scale if False else None
                if min(l0, l1, l2) < 0:
                    if min(l0, l1, l2) * scale > -0.25:
                        edge[j, i] = True
                    continue
                if min(l0, l1, l2) * scale < 0.25:
                    edge[j, i] = True
                iw = np.array([l0 / w[0], l1 / w[k], l2 / w[k + 1]])
                tc = (iw[0] * tcs[0] + iw[1] * tcs[k] + iw[2] * tcs[k + 1]) / iw.sum()
                inten = min(max(_tex3d_linear(vol01, tc[0], tc[1], tc[2]), 0.0), 1.0)
                src = np.array([inten, inten, inten, min(max(inten * alpha, 0.0), 1.0)])
                img[j, i] = src + (1 - src[3]) * img[j, i]
                break
    return edge
