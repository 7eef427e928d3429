"""ctypes/numpy front end of oracle/liboracle.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (simian-spacemonkey_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("smk_oracle.c", "smk_prep.c", "smk_oracle.h")]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _proto(_LIB)
    return _LIB


class Volume(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("nelts", C.c_int),
                ("dtype", C.c_int), ("data", C.c_void_p), ("grad", C.c_void_p),
                ("fx", C.c_float), ("fy", C.c_float), ("fz", C.c_float),
                ("g0", C.c_int * 3), ("g1", C.c_int * 3),
                ("clip_axis", C.c_int), ("clip_vpos", C.c_float * 3),
                ("cplane_on", C.c_int), ("cplane", C.c_float * 4)]


class Classify(C.Structure):
    _fields_ = [("mode", C.c_int), ("tlut", C.c_void_p), ("tlut_size", C.c_int),
                ("tf_vg", C.c_void_p), ("sv", C.c_int), ("sg", C.c_int),
                ("tf_h", C.c_void_p), ("third_axis", C.c_int),
                ("tf3d", C.c_void_p), ("s3v", C.c_int), ("s3g", C.c_int), ("s3h", C.c_int)]


class Camera(C.Structure):
    _fields_ = [("mv", C.c_double * 16), ("frustum", C.c_float * 4), ("znear", C.c_float),
                ("width", C.c_int), ("height", C.c_int), ("sample_rate", C.c_float),
                ("steps", C.c_int)]


class Shade(C.Structure):
    _fields_ = [("mode", C.c_int), ("L", C.c_float * 3), ("Hv", C.c_float * 3),
                ("xform", C.c_float * 16), ("intens", C.c_float), ("use_spec", C.c_int)]


class Perturb(C.Structure):
    _fields_ = [("on", C.c_int), ("noise", C.c_void_p), ("n", C.c_int),
                ("w", C.c_float * 4), ("s", C.c_float * 4)]


class RayCoef(C.Structure):
    _fields_ = [("pxs", C.c_float), ("pxl", C.c_float), ("pys", C.c_float), ("pyl", C.c_float),
                ("Ac", C.c_float * 3), ("Ax", C.c_float * 3), ("Ay", C.c_float * 3),
                ("Bc", C.c_float * 3), ("Bx", C.c_float * 3), ("By", C.c_float * 3),
                ("nplanes", C.c_int), ("tau0", C.c_float), ("dtau", C.c_float),
                ("zmin", C.c_float), ("zmax", C.c_float), ("dis", C.c_float)]


class ShadowCoef(C.Structure):
    _fields_ = [("pxs", C.c_float), ("pxl", C.c_float), ("pys", C.c_float), ("pyl", C.c_float),
                ("Ec", C.c_float * 3), ("Dc", C.c_float * 3), ("Dx", C.c_float * 3), ("Dy", C.c_float * 3),
                ("nDc", C.c_float), ("nDx", C.c_float), ("nDy", C.c_float), ("num0", C.c_float), ("dnum", C.c_float),
                ("las", C.c_float), ("lal", C.c_float), ("Lc", C.c_float * 3),
                ("Gc", C.c_float * 3), ("Gx", C.c_float * 3), ("Gy", C.c_float * 3),
                ("nGc", C.c_float), ("nGx", C.c_float), ("nGy", C.c_float), ("lnum0", C.c_float), ("ldnum", C.c_float),
                ("Xm", C.c_float * 4), ("Ym", C.c_float * 4), ("Wm", C.c_float * 4),
                ("lscale", C.c_float), ("lbias", C.c_float),
                ("nslices", C.c_int), ("LB", C.c_int), ("front_to_back", C.c_int)]


class LevWidget(C.Structure):
    _fields_ = [("type", C.c_int), ("verts", (C.c_float * 2) * 3), ("thresh", C.c_float * 2),
                ("color", C.c_float * 3), ("alpha", C.c_float), ("be", C.c_float),
                ("faux", C.c_int)]


def _proto(L):
    P = C.POINTER
    L.orc_render.restype = C.c_int
    L.orc_render.argtypes = [P(Volume), P(Classify), P(Camera), P(Shade), P(Perturb), C.c_int,
                             C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_render_pixels.restype = C.c_int
    L.orc_render_pixels.argtypes = [P(Volume), P(Classify), P(Camera), P(Shade), P(Perturb),
                                    C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_ray_setup.argtypes = [P(Volume), P(Camera), P(RayCoef)]
    L.orc_shadow_setup.restype = C.c_int
    L.orc_shadow_setup.argtypes = [P(Volume), P(Camera), P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float),
                                   C.c_int, C.c_float, P(ShadowCoef)]
    L.orc_render_shadow.restype = C.c_int
    L.orc_render_shadow.argtypes = [P(Volume), P(Classify), P(Camera), P(Shade), P(ShadowCoef), C.c_void_p, C.c_void_p, C.c_int]
    L.orc_last_inside_samples.restype = C.c_longlong
    L.orc_composite_over.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_shade_setup.argtypes = [C.c_int, C.c_int, P(C.c_float), P(C.c_float), P(C.c_float),
                                  P(C.c_float), C.c_float, P(Shade)]
    L.orc_modelview.argtypes = [P(C.c_float)] * 6 + [P(C.c_double)]
    L.orc_srand.argtypes = [C.c_uint]
    L.orc_rand.restype = C.c_int
    L.orc_noise3.restype = C.c_double
    L.orc_noise3.argtypes = [P(C.c_double)]
    for f in (L.orc_perlin3d, L.orc_perlin3d_abs):
        f.restype = C.c_double
        f.argtypes = [C.c_double] * 5 + [C.c_int]
    L.orc_genvol_spheres.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_double, P(C.c_float), C.c_float,
                                     C.c_float]
    L.orc_genvol_perl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  P(C.c_float), C.c_float, C.c_float]
    L.orc_genvol_blur.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, P(C.c_float)]
    L.orc_make_vgh.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p]
    L.orc_normals_vgh.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p]
    L.orc_hist2d.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]
    L.orc_hist2d.restype = C.c_int
    L.orc_merge_addg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p]
    L.orc_brick_grid.argtypes = [C.c_int] * 4 + [P(C.c_int)]
    for name in ("orc_tlut_default", "orc_tlut_spectral", "orc_tlut_blackbody",
                 "orc_tlut_cyanmagenta"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int]
    L.orc_tlut_channel_ramp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]
    L.orc_tlut_scale_alpha.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
    L.orc_tlut_premultiply.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_deptex_default.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.orc_copy_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.orc_lev_setpos.argtypes = [P(LevWidget), P(C.c_float), P(C.c_float), P(C.c_float),
                                 C.c_float, C.c_float]
    L.orc_lev_rasterize.argtypes = [P(LevWidget), C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_hsl_color.argtypes = [C.c_float, C.c_float, C.c_float, P(C.c_float)]
    L.orc_rasterize_vgh.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.orc_noise_tex.argtypes = [C.c_void_p, C.c_int]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f3(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


# ----------------------------------------------------------------------------- scene wrapper

IDENTITY = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]


def rotation(axis, deg):
    """column-major 4x4 rotation (what Trackball hands to gluvv.rinfo.xform)"""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    t = np.deg2rad(deg)
    c, s = np.cos(t), np.sin(t)
    x, y, z = a
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    M = np.eye(4)
    M[:3, :3] = R
    return [float(v) for v in M.T.reshape(-1)]  # column-major


def modelview(eye, at, up, trans, xform, fsize):
    mv = (C.c_double * 16)()
    lib().orc_modelview(_f3(eye), _f3(at), _f3(up), _f3(trans), _f3(xform), _f3(fsize), mv)
    return list(mv)


class Scene:
    """Plain description of one frame; the same object drives the oracle (here) and the HIP
    product (tests/_smk.py) so both see identical inputs."""

    def __init__(self, data, fsize=None, grad=None):
        data = np.ascontiguousarray(data)
        assert data.ndim == 4 and data.dtype in (np.uint8, np.float32)
        self.data, self.grad = data, (np.ascontiguousarray(grad) if grad is not None else None)
        nz, ny, nx, ne = data.shape
        self.dims = (nx, ny, nz)
        self.nelts = ne
        m = float(max(nx, ny, nz))
        # whole volume normalised so the largest dimension is 1 (MetaVolume.cpp:1047-1054)
        self.fsize = tuple(np.float32(v) for v in (fsize or (nx / m, ny / m, nz / m)))
        self.region = ((0, 0, 0), (nx, ny, nz))
        self.tf_mode = 1
        self.tlut = None
        self.tf_vg = self.tf_h = self.tf3d = None
        self.third_axis = 0
        self.eye, self.at, self.up = (0, 0, -7), (0, 0, 0), (0, 1, 0)   # gluvv.cpp:263-271
        self.trans = (0, 0, 0)
        self.xform = list(IDENTITY)
        self.frustum = (-0.5 / 7, 0.5 / 7, -0.5 / 7, 0.5 / 7)
        self.znear = 1.0
        self.width = self.height = 64
        self.sample_rate, self.steps = 0.0, 64
        self.shade_mode, self.use_spec = 0, 1
        self.light_pos, self.intens = (0, 0, -5), 0.75                  # gluvv.cpp:293-305
        self.noise = None
        self.pert_w, self.pert_s = (0, 0, 0, 0), (0.2, 2.1, 4.5, 8.7)   # gluvvui.cpp:213-267
        self.mv_override = None
        self.clip_plane = None  # free clip plane, eye space (what glClipPlane stores): keep plane . (x_eye,1) >= 0
        self.clip = None        # (axis 1..6 = X+ X- Y+ Y- Z+ Z-, vpos[3] in volume space): gluvv.clip, ortho mode
        self.shadow = None      # (buffer_px, quality): gluvv.light.shadow with buffsz / g|iShadowQual (gluvv.cpp:287-300)

    def mv(self):
        if self.mv_override is not None:
            return list(self.mv_override)
        return modelview(self.eye, self.at, self.up, self.trans, self.xform, self.fsize)

    # -- ctypes views
    def c_volume(self):
        v = Volume()
        v.nx, v.ny, v.nz = self.dims
        v.nelts = self.nelts
        v.dtype = 0 if self.data.dtype == np.uint8 else 1
        v.data = _p(self.data)
        v.grad = _p(self.grad)
        v.fx, v.fy, v.fz = self.fsize
        v.g0[:] = self.region[0]
        v.g1[:] = self.region[1]
        v.clip_axis = self.clip[0] if self.clip else 0
        v.clip_vpos[:] = self.clip[1] if self.clip else (0, 0, 0)
        v.cplane_on = 1 if self.clip_plane is not None else 0
        if self.clip_plane is not None:
            out = (C.c_float * 4)()
            lib().orc_clip_plane_voxel((C.c_double * 4)(*[float(x) for x in self.clip_plane]), (C.c_double * 16)(*self.mv()),
                                       (C.c_float * 3)(*[float(f) for f in self.fsize]), (C.c_int * 3)(*self.dims), out)
            v.cplane[:] = list(out)
        return v

    def c_classify(self):
        t = Classify()
        t.mode = self.tf_mode
        if self.tlut is not None:
            t.tlut, t.tlut_size = _p(self.tlut), self.tlut.shape[0]
        if self.tf_vg is not None:
            t.tf_vg, t.sg, t.sv = _p(self.tf_vg), self.tf_vg.shape[0], self.tf_vg.shape[1]
        t.tf_h = _p(self.tf_h)
        t.third_axis = self.third_axis
        if self.tf3d is not None:
            t.tf3d = _p(self.tf3d)
            t.s3h, t.s3g, t.s3v = self.tf3d.shape[:3]
        return t

    def c_camera(self):
        c = Camera()
        c.mv[:] = self.mv()
        c.frustum[:] = self.frustum
        c.znear = self.znear
        c.width, c.height = self.width, self.height
        c.sample_rate, c.steps = self.sample_rate, self.steps
        return c

    def c_shade(self):
        s = Shade()
        lib().orc_shade_setup(self.shade_mode, self.use_spec, _f3(self.light_pos), _f3(self.eye),
                              _f3(self.at), _f3(self.xform), self.intens, C.byref(s))
        return s

    def c_perturb(self):
        p = Perturb()
        p.on = 1 if (self.noise is not None and any(self.pert_w)) else 0
        if self.noise is not None:
            p.noise, p.n = _p(self.noise), self.noise.shape[0]
        p.w[:] = self.pert_w
        p.s[:] = self.pert_s
        return p

    # -- oracle entry points
    def raycoef(self):
        rc = RayCoef()
        v, c = self.c_volume(), self.c_camera()
        lib().orc_ray_setup(C.byref(v), C.byref(c), C.byref(rc))
        return rc

    def render(self, blend=0, depth=False, nthreads=0, rows=None):
        out = np.zeros((self.height, self.width, 4), np.float32)
        dep = np.zeros((self.height, self.width), np.float32) if depth else None
        v, t, c, s, p = (self.c_volume(), self.c_classify(), self.c_camera(), self.c_shade(),
                         self.c_perturb())
        r0, r1 = rows if rows else (0, self.height)
        rc = lib().orc_render(C.byref(v), C.byref(t), C.byref(c), C.byref(s), C.byref(p), blend,
                              _p(out), _p(dep), r0, r1, nthreads)
        assert rc == 0
        return (out, dep) if depth else out

    def shadowcoef(self):
        sc = ShadowCoef()
        v, c = self.c_volume(), self.c_camera()
        rc = lib().orc_shadow_setup(C.byref(v), C.byref(c), _f3(self.light_pos), _f3(self.eye), _f3(self.at),
                                    _f3(self.xform), int(self.shadow[0]), float(self.shadow[1]), C.byref(sc))
        assert rc == 0, "orc_shadow_setup: degenerate light"
        return sc

    def render_shadow(self, nthreads=0):
        """half-angle-slicing frame: (rgba [H][W][4], light buffer [LB][LB][4])"""
        sc = self.shadowcoef()
        out = np.zeros((self.height, self.width, 4), np.float32)
        light = np.zeros((sc.LB, sc.LB, 4), np.float32)
        v, t, c, s = self.c_volume(), self.c_classify(), self.c_camera(), self.c_shade()
        rc = lib().orc_render_shadow(C.byref(v), C.byref(t), C.byref(c), C.byref(s), C.byref(sc), _p(out), _p(light), nthreads)
        assert rc == 0, rc
        return out, light

    def render_pixels(self, pix, blend=0):
        pix = np.ascontiguousarray(pix, np.int32)
        out = np.zeros((pix.shape[0], 4), np.float32)
        v, t, c, s, p = (self.c_volume(), self.c_classify(), self.c_camera(), self.c_shade(),
                         self.c_perturb())
        rc = lib().orc_render_pixels(C.byref(v), C.byref(t), C.byref(c), C.byref(s), C.byref(p),
                                     blend, _p(pix), pix.shape[0], _p(out))
        assert rc == 0
        return out


def inside_samples():
    return int(lib().orc_last_inside_samples())


def composite_over(layers):
    layers = np.ascontiguousarray(layers, np.float32)
    n, npix = layers.shape[0], int(np.prod(layers.shape[1:-1]))
    out = np.zeros(layers.shape[1:], np.float32)
    lib().orc_composite_over(_p(layers), n, npix, _p(out))
    return out


# ----------------------------------------------------------------------------- data prep

def genvol_spheres(n, seed=1, nspheres=4, pharm=10, pscale=0.7, pwrap=(3, 3, 3), pabs=True,
                   blur=True, bw=(1, 1, 1, .7), use_perl=True):
    """genvol -spheres 4 -p 10 -pscale .7 -pwrap 3 3 3 -pabs -blur -bw 1 1 1 .7
    (genvol/scripts/make64.bat:1), srand(seed) before the Perlin tables are drawn."""
    L = lib()
    sx, sy, sz = (n, n, n) if np.isscalar(n) else n
    d = np.zeros((sz, sy, sx), np.uint8)
    L.orc_srand(seed)
    L.orc_perlin_reset()
    if use_perl:
        L.orc_perlin_init()   # genvol/main.cpp:118-120; the first noise3 call re-inits (perlin.c:84-87)
    L.orc_genvol_spheres(_p(d), sx, sy, sz, nspheres, int(use_perl), 1 if pabs else 0, pharm,
                         pscale, _f3(pwrap), 2.0, 2.0)
    if blur:
        L.orc_genvol_blur(_p(d), sx, sy, sz, _f3(bw))
    return d


def genvol_perl(n, seed=1, param=3, pharm=4, pwrap=(3, 3, 3)):
    """genvol default PERLIN_VOL mode, -s n n n -p 4 -pwrap 3 3 3"""
    L = lib()
    sx, sy, sz = (n, n, n) if np.isscalar(n) else n
    d = np.zeros((sz, sy, sx), np.uint8)
    L.orc_srand(seed)
    L.orc_perlin_reset()
    L.orc_perlin_init()       # `-p 4` sets use_perl (genvol/main.cpp:510-513) => main's init()
    L.orc_genvol_perl(_p(d), sx, sy, sz, param, pharm, _f3(pwrap), 2.0, 2.0)
    return d


def make_vgh(vol, compat=True, f32=False):
    vol = np.ascontiguousarray(vol)
    sz, sy, sx = vol.shape
    dt = 0 if vol.dtype == np.uint8 else 1
    if dt == 1:
        vol = vol.astype(np.float32)
    o8 = np.zeros((sz, sy, sx, 3), np.uint8)
    of = np.zeros((sz, sy, sx, 3), np.float32) if f32 else None
    lib().orc_make_vgh(_p(vol), dt, sx, sy, sz, int(compat), _p(o8), _p(of))
    return (o8, of) if f32 else o8


def normals_vgh(vgh_u8, blur=False):
    sz, sy, sx, ne = vgh_u8.shape
    out = np.zeros((sz, sy, sx, 3), np.uint8)
    lib().orc_normals_vgh(_p(np.ascontiguousarray(vgh_u8)), ne, sx, sy, sz, int(blur), _p(out))
    return out


def hist2d(vol_u8):
    """MetaVolume::hist2D: log-scaled joint (value, gradient) histogram [g][v] of a [z][y][x][nelts] volume"""
    v = np.ascontiguousarray(vol_u8)
    out = np.zeros((256, 256), np.uint8)
    ok = lib().orc_hist2d(_p(v), v.shape[-1], v.size // v.shape[-1], _p(out))
    return out if ok else None


def merge_addg(fields_u8):
    sz, sy, sx, nf = fields_u8.shape
    out = np.zeros((sz, sy, sx, nf + 1), np.uint8)
    grad = np.zeros((sz, sy, sx, 3), np.uint8)
    lib().orc_merge_addg(_p(np.ascontiguousarray(fields_u8)), nf, sx, sy, sz, _p(out), _p(grad))
    return out, grad


def brick_grid(sx, sy, sz, maxsz):
    d = (C.c_int * 3)()
    lib().orc_brick_grid(sx, sy, sz, maxsz, d)
    return tuple(d)


def tlut(kind="default", size=256):
    t = np.zeros((size, 4), np.float32)
    L = lib()
    L.orc_tlut_default(_p(t), size)
    if kind != "default":
        getattr(L, "orc_tlut_" + kind)(_p(t), size)
    return t


def tlut_ramp(t, ch, i0, i1, v0, v1):
    lib().orc_tlut_channel_ramp(_p(t), ch, i0, i1, v0, v1)
    return t


def tlut_scale_alpha(t, last, rate):
    lib().orc_tlut_scale_alpha(_p(t), t.shape[0], last, rate)
    return t


def tlut_premultiply(t):
    o = np.zeros_like(t)
    lib().orc_tlut_premultiply(_p(t), t.shape[0], _p(o))
    return o


def tlut_volumerenderable(size=256):
    """state after VolumeRenderable::init (VolumeRenderable.cpp:74-78)"""
    t = tlut("default", size)
    L = lib()
    L.orc_tlut_blackbody(_p(t), size)
    L.orc_tlut_cyanmagenta(_p(t), size)
    L.orc_tlut_spectral(_p(t), size)
    return tlut_ramp(t, 3, 0, size - 1, 0.0, 0.1)


def deptex_default(sx=256, sy=256):
    a = np.zeros((sy, sx, 4), np.uint8)
    b = np.zeros((sy, sx, 4), np.uint8)
    lib().orc_deptex_default(_p(a), _p(b), sx, sy)
    return a, b


def copy_scale(tex, sr):
    out = np.zeros_like(tex)
    lib().orc_copy_scale(_p(np.ascontiguousarray(tex)), _p(out), tex.shape[1], tex.shape[0], sr)
    return out


def lev_widget(kind, b=(.5, 0), l=(.3, .7), r=(.7, .7), tw=-10, th=-10, hsl=(0, 1, .5),
               alpha=.5, be=1.0, faux=False):
    """LevWidget with the constructor defaults (LevWidget.cpp:35-63) and TFWidgetRen::init's
    position (TFWidgetRen1.cpp:631-633); kind = 'triangle' | 'ellipse' | '1d' | 'default'"""
    w = LevWidget()
    w.type = {"triangle": 0, "ellipse": 1, "1d": 2, "default": 3}[kind]
    lib().orc_lev_setpos(C.byref(w), _f3(b), _f3(l), _f3(r), tw, th)
    col = (C.c_float * 3)()
    lib().orc_hsl_color(*hsl, col)
    w.color[:] = list(col)
    w.alpha, w.be, w.faux = alpha, be, int(faux)
    return w


def lev_rasterize(w, tex):
    sh = 1 if tex.ndim == 3 else tex.shape[0]
    sg, sv = tex.shape[-3], tex.shape[-2]
    lib().orc_lev_rasterize(C.byref(w), _p(tex), sv, sg, sh)
    return tex


def rasterize_vgh(tex, slider1hi):
    lib().orc_rasterize_vgh(_p(tex), tex.shape[1], tex.shape[0], slider1hi)
    return tex


def noise_tex(n=32):
    out = np.zeros((n, n, n, 4), np.uint8)
    lib().orc_noise_tex(_p(out), n)
    return out
