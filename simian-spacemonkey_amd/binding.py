"""ctypes binding of csrc/libsmk_hip.so (the C ABI in include/smk.h).

Mirrors how the reference drives a renderer (gluvvPrimitive::init()/draw(),
VolumeRenderable.cpp:36-82): upload once, then per frame set camera / sampling / TF and render.
There is no CPU path: if the library or a HIP device is missing, constructing a Renderer raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB = None

ABI_SYMBOLS = [
    "smk_create", "smk_destroy", "smk_last_error", "smk_upload_volume",
    "smk_upload_volume_device", "smk_set_shard", "smk_set_clip", "smk_set_clip_plane", "smk_hist2d", "smk_hist2d_device", "smk_merge_fields_device", "smk_shard_order", "smk_set_tlut1d",
    "smk_set_tf2d", "smk_set_tf3d", "smk_set_camera", "smk_set_shading", "smk_set_sampling",
    "smk_set_perturb", "smk_set_blend", "smk_set_shadow", "smk_get_shadowcoef", "smk_get_light_buffer", "smk_get_brick_flags", "smk_render", "smk_render_device", "smk_composite_over_device",
    "smk_make_vgh_device", "smk_normals_vgh_device", "smk_synth_volume_device",
    "smk_get_raycoef", "smk_set_option", "smk_last_frame_info", "smk_get_stat", "smk_get_trace", "smk_get_tf2d_effective",
    "smk_timing_reset", "smk_timing_read", "smk_last_frame_id", "smk_frame_failed",
    "smk_exchange_unique_id", "smk_exchange_create", "smk_exchange_connect_local", "smk_exchange_destroy",
    "smk_exchange_last_error", "smk_exchange_partial", "smk_exchange_acquire", "smk_exchange_rendered", "smk_exchange_frame",
    "smk_exchange_frame_local", "smk_exchange_wait", "smk_exchange_set_order",
    "smk_set_region", "smk_render_slice", "smk_render_slice_device", "smk_count_samples",
]

# gluvvDataMode order (gluvv.h:221-235)
GDM = {n: i for i, n in enumerate(
    ["V1", "V1G", "V1GH", "V2", "V2G", "V2GH", "V3", "V3G", "V4", "VGH", "VGH_VG", "VGH_V"])}
SHADE = {"none": 0, "r8k_diff": 1, "r8k": 2, "nv20_diff": 3, "nv20": 4}


class SmkError(RuntimeError):
    pass


class VolumeDesc(C.Structure):
    _fields_ = [("xiSize", C.c_int), ("yiSize", C.c_int), ("ziSize", C.c_int),
                ("xfSize", C.c_float), ("yfSize", C.c_float), ("zfSize", C.c_float),
                ("xiPos", C.c_int), ("yiPos", C.c_int), ("ziPos", C.c_int),
                ("xfPos", C.c_float), ("yfPos", C.c_float), ("zfPos", C.c_float),
                ("data", C.c_void_p), ("grad", C.c_void_p)]


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


def library_path():
    # SMK_LIB: developer override (kernel experiments build variant libraries side by side)
    return os.environ.get("SMK_LIB") or os.path.join(_CSRC, "libsmk_hip.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of the product library (cross-compiles without a GPU)."""
    so = library_path()
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "smk.h"))
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _CSRC, "-s", "-j4", "all"])
    return so


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = library_path()
    if not os.path.exists(so):
        raise SmkError("libsmk_hip.so is not built (run __graft_entry__.build()); "
                       "there is no CPU fallback")
    L = C.CDLL(so)
    if os.environ.get("SMK_LIB"):
        # developer override (an older or experimental build beside the product library): entry points
        # it lacks become stubs that fail when called, so the prototypes below can still be set
        class _Missing:
            def __init__(self, name):
                self.name = name

            def __call__(self, *a):
                raise SmkError("%s is not in %s" % (self.name, so))
        for name in ABI_SYMBOLS:
            try:
                getattr(L, name)
            except AttributeError:
                setattr(L, name, _Missing(name))
    P = C.POINTER
    L.smk_create.restype = C.c_void_p
    L.smk_create.argtypes = [C.c_int, P(C.c_int)]
    L.smk_destroy.argtypes = [C.c_void_p]
    L.smk_destroy.restype = None
    L.smk_last_error.restype = C.c_char_p
    L.smk_last_error.argtypes = [C.c_void_p]
    for n in ("smk_upload_volume", "smk_upload_volume_device"):
        getattr(L, n).argtypes = [C.c_void_p, P(VolumeDesc), C.c_int, C.c_int, C.c_int, C.c_int]
    L.smk_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.smk_set_clip.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.smk_set_clip_plane.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    L.smk_set_region.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.smk_render_slice.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float)]
    L.smk_render_slice_device.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_float, C.c_void_p, C.c_void_p]
    L.smk_shard_order.argtypes = [C.c_void_p, P(C.c_int)]
    L.smk_set_tlut1d.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.smk_set_tf2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.smk_set_tf3d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.smk_set_camera.argtypes = [C.c_void_p, P(C.c_double), P(C.c_float), P(C.c_float), C.c_int,
                                 C.c_int]
    L.smk_set_shading.argtypes = [C.c_void_p, C.c_int, P(C.c_float), P(C.c_float), P(C.c_float),
                                  P(C.c_float), C.c_float, C.c_float]
    L.smk_set_sampling.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_float, C.c_int]
    L.smk_set_perturb.argtypes = [C.c_void_p, C.c_void_p, C.c_int, P(C.c_float), P(C.c_float)]
    L.smk_set_blend.argtypes = [C.c_void_p, C.c_int]
    L.smk_set_shadow.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
    L.smk_get_shadowcoef.argtypes = [C.c_void_p, P(ShadowCoef)]
    L.smk_get_light_buffer.argtypes = [C.c_void_p, C.c_void_p, P(C.c_int)]
    L.smk_get_brick_flags.argtypes = [C.c_void_p, C.c_void_p, P(C.c_int), P(C.c_int)]
    L.smk_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.smk_render_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.smk_composite_over_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, P(C.c_int), C.c_int,
                                            C.c_void_p, C.c_void_p]
    L.smk_make_vgh_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p, C.c_void_p]
    L.smk_normals_vgh_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p]
    L.smk_merge_fields_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.smk_hist2d.argtypes = [C.c_void_p, C.POINTER(VolumeDesc), C.c_int, C.c_int, C.c_void_p]
    L.smk_hist2d_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.smk_synth_volume_device.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]
    L.smk_get_raycoef.argtypes = [C.c_void_p, P(RayCoef)]
    L.smk_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.smk_last_frame_info.argtypes = [C.c_void_p, P(C.c_int), P(C.c_float), P(C.c_double)]
    L.smk_get_stat.argtypes = [C.c_void_p, C.c_char_p, P(C.c_double)]
    L.smk_get_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int, P(C.c_int)]
    L.smk_get_tf2d_effective.argtypes = [C.c_void_p, C.c_void_p, P(C.c_float)]
    L.smk_last_frame_id.argtypes = [C.c_void_p]
    L.smk_last_frame_id.restype = C.c_longlong
    L.smk_frame_failed.argtypes = [C.c_void_p, C.c_longlong]
    L.smk_exchange_unique_id.argtypes = [C.c_void_p]
    L.smk_exchange_create.restype = C.c_void_p
    L.smk_exchange_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, P(C.c_int)]
    L.smk_exchange_connect_local.argtypes = [P(C.c_void_p), C.c_int]
    L.smk_exchange_destroy.argtypes = [C.c_void_p]
    L.smk_exchange_destroy.restype = None
    L.smk_exchange_last_error.restype = C.c_char_p
    L.smk_exchange_last_error.argtypes = [C.c_void_p]
    L.smk_exchange_partial.restype = C.c_void_p
    L.smk_exchange_partial.argtypes = [C.c_void_p, C.c_int]
    L.smk_exchange_acquire.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.smk_exchange_rendered.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.smk_exchange_set_order.argtypes = [C.c_void_p, C.c_int, P(C.c_int)]
    L.smk_exchange_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.smk_exchange_frame_local.argtypes = [P(C.c_void_p), C.c_int, C.c_int, C.c_void_p]
    L.smk_exchange_wait.argtypes = [C.c_void_p, C.c_void_p]
    L.smk_timing_reset.argtypes = [C.c_void_p]
    L.smk_timing_read.argtypes = [C.c_void_p, P(C.c_float), P(C.c_int)]
    L.smk_count_samples.argtypes = [C.c_void_p, P(C.c_double)]
    _LIB = L
    return L


def _fa(v, n=None):
    v = [float(x) for x in v]
    return (C.c_float * (n or len(v)))(*v)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def split_bricks(dims, fsize, grid):
    """MetaVolume::brick geometry (MetaVolume.cpp:1394-1417): equal bricks, fPos = fSize*index."""
    nx, ny, nz = dims
    xd, yd, zd = grid
    bx, by, bz = nx // xd, ny // yd, nz // zd
    out = []
    for i in range(zd):
        for j in range(yd):
            for k in range(xd):
                out.append(dict(ipos=(k * bx, j * by, i * bz), isize=(bx, by, bz),
                                fsize=(fsize[0] * bx / nx, fsize[1] * by / ny, fsize[2] * bz / nz)))
    for b in out:
        idx = [b["ipos"][a] // b["isize"][a] for a in range(3)]
        b["fpos"] = tuple(b["fsize"][a] * idx[a] for a in range(3))
    return out


class Renderer:
    """One smk_ctx.  Usage mirrors VolumeRenderable: upload_volume() at init(), the set_*()
    calls + render() at draw()."""

    def __init__(self, device=0):
        self.L = load_library()
        err = C.c_int(0)
        self.ctx = self.L.smk_create(device, C.byref(err))
        if not self.ctx:
            raise SmkError(self.L.smk_last_error(None).decode())
        self.size = (0, 0)
        self._keep = []

    def close(self):
        if self.ctx:
            self.L.smk_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise SmkError(self.L.smk_last_error(self.ctx).decode())

    # -- volume
    def upload_volume(self, data, grad=None, fsize=None, grid=(1, 1, 1), dmode="VGH"):
        """data [nz][ny][nx][nelts] u8/f32 (numpy); split into `grid` bricks the way
        MetaVolume::brick does and handed over brick by brick."""
        data = np.ascontiguousarray(data)
        nz, ny, nx, ne = data.shape
        m = float(max(nx, ny, nz))
        fsize = fsize or (nx / m, ny / m, nz / m)
        bricks = split_bricks((nx, ny, nz), fsize, grid)
        descs = (VolumeDesc * len(bricks))()
        keep = []
        for d, b in zip(descs, bricks):
            (x0, y0, z0), (bx, by, bz) = b["ipos"], b["isize"]
            sub = np.ascontiguousarray(data[z0:z0 + bz, y0:y0 + by, x0:x0 + bx])
            keep.append(sub)
            d.xiSize, d.yiSize, d.ziSize = bx, by, bz
            d.xfSize, d.yfSize, d.zfSize = b["fsize"]
            d.xiPos, d.yiPos, d.ziPos = x0, y0, z0
            d.xfPos, d.yfPos, d.zfPos = b["fpos"]
            d.data = _ptr(sub)
            if grad is not None:
                g = np.ascontiguousarray(grad[z0:z0 + bz, y0:y0 + by, x0:x0 + bx])
                keep.append(g)
                d.grad = _ptr(g)
        dt = 0 if data.dtype == np.uint8 else 1
        self._ck(self.L.smk_upload_volume(self.ctx, descs, len(bricks), ne, dt, GDM[dmode]))

    def upload_volume_device(self, dptr, dims, nelts, dtype, grad_dptr=None, fsize=None,
                             dmode="VGH"):
        nx, ny, nz = dims
        m = float(max(nx, ny, nz))
        fsize = fsize or (nx / m, ny / m, nz / m)
        d = VolumeDesc()
        d.xiSize, d.yiSize, d.ziSize = nx, ny, nz
        d.xfSize, d.yfSize, d.zfSize = fsize
        d.data = dptr
        d.grad = grad_dptr
        self._ck(self.L.smk_upload_volume_device(self.ctx, C.byref(d), 1, nelts, dtype, GDM[dmode]))

    def set_shard(self, rank, nranks):
        self._ck(self.L.smk_set_shard(self.ctx, rank, nranks))

    def shard_order(self, nranks):
        o = (C.c_int * nranks)()
        self._ck(self.L.smk_shard_order(self.ctx, o))
        return list(o)

    # -- classification
    def set_tlut1d(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32)
        self._ck(self.L.smk_set_tlut1d(self.ctx, _ptr(rgba), rgba.shape[0]))

    def set_tf2d(self, deptex, deptex2=None):
        deptex = np.ascontiguousarray(deptex, np.uint8)
        d2 = np.ascontiguousarray(deptex2, np.uint8) if deptex2 is not None else None
        self._ck(self.L.smk_set_tf2d(self.ctx, _ptr(deptex), _ptr(d2), deptex.shape[1], deptex.shape[0]))

    def set_tf3d(self, ptex):
        ptex = np.ascontiguousarray(ptex, np.uint8)
        sh, sg, sv = ptex.shape[:3]
        self._ck(self.L.smk_set_tf3d(self.ctx, _ptr(ptex), sv, sg, sh))

    # -- per frame state
    def set_camera(self, mv, frustum, clip, width, height):
        self._ck(self.L.smk_set_camera(self.ctx, (C.c_double * 16)(*mv), _fa(frustum), _fa(clip),
                                       width, height))
        self.size = (width, height)

    def set_shading(self, mode, light_pos, eye, at, xform, intens=0.75, amb=0.05):
        self._ck(self.L.smk_set_shading(self.ctx, SHADE[mode] if isinstance(mode, str) else mode,
                                        _fa(light_pos), _fa(eye), _fa(at), _fa(xform), intens, amb))

    def set_sampling(self, sample_rate=0.0, steps=0, gamma=1.0, scale_alphas=1):
        self._ck(self.L.smk_set_sampling(self.ctx, sample_rate, steps, gamma, scale_alphas))

    def set_clip(self, axis, vpos):
        """orthogonal clip plane: axis 1..6 = X+ X- Y+ Y- Z+ Z- (VolRenMajorAxis), vpos in volume space;
        axis 0 / None switches it off"""
        if not axis:
            self._ck(self.L.smk_set_clip(self.ctx, 0, 0, None))
        else:
            self._ck(self.L.smk_set_clip(self.ctx, 1, int(axis), _fa(vpos)))

    def set_clip_plane(self, plane_eye):
        """free clip plane in eye space (what glClipPlane stores); None switches it off"""
        if plane_eye is None:
            self._ck(self.L.smk_set_clip_plane(self.ctx, 0, None))
        else:
            self._ck(self.L.smk_set_clip_plane(self.ctx, 1, (C.c_double * 4)(*[float(v) for v in plane_eye])))

    def set_region(self, lo=None, hi=None):
        """sub-box of the volume in volume space (renderVolume's x/y/zext); None = off"""
        if lo is None:
            self._ck(self.L.smk_set_region(self.ctx, 0, None, None))
        else:
            self._ck(self.L.smk_set_region(self.ctx, 1, _fa(lo, 3), _fa(hi, 3)))

    def render_slice(self, quad, alpha, rgba):
        """VolumeRenderer::renderSlice: one textured quad (4 x 3, model space) blended into rgba [H][W][4] (in place)"""
        q = np.ascontiguousarray(quad, np.float32).reshape(12)
        assert rgba.dtype == np.float32 and rgba.flags["C_CONTIGUOUS"]
        self._ck(self.L.smk_render_slice(self.ctx, q.ctypes.data_as(C.POINTER(C.c_float)), float(alpha), rgba.ctypes.data_as(C.POINTER(C.c_float))))
        return rgba

    def set_perturb(self, noise, w, s):
        if noise is None:
            self._ck(self.L.smk_set_perturb(self.ctx, None, 0, _fa((0, 0, 0, 0)), _fa((0, 0, 0, 0))))
            return
        noise = np.ascontiguousarray(noise, np.uint8)
        self._ck(self.L.smk_set_perturb(self.ctx, _ptr(noise), noise.shape[0], _fa(w, 4), _fa(s, 4)))

    def set_blend(self, mode):
        """0 / "ftb" front to back (default), 1 / "btf" back to front, 2 / "max" GL_MAX (MIP)"""
        mode = {"ftb": 0, "btf": 1, "max": 2}.get(mode, mode)
        self._ck(self.L.smk_set_blend(self.ctx, int(mode)))

    def set_shadow(self, on, buffer_px=1024, quality=0.5):
        """gluvv.light.shadow with gluvv.light.buffsz[0] and g/iShadowQual (half-angle slicing)"""
        self._ck(self.L.smk_set_shadow(self.ctx, int(bool(on)), int(buffer_px), float(quality)))

    def shadowcoef(self):
        sc = ShadowCoef()
        self._ck(self.L.smk_get_shadowcoef(self.ctx, C.byref(sc)))
        return sc

    def light_buffer(self):
        """the light buffer as the last frame with shadows left it: [LB][LB][4] float32"""
        lb = C.c_int(0)
        self._ck(self.L.smk_get_light_buffer(self.ctx, None, C.byref(lb)))
        out = np.zeros((lb.value, lb.value, 4), np.float32)
        self._ck(self.L.smk_get_light_buffer(self.ctx, out.ctypes.data_as(C.c_void_p), C.byref(lb)))
        return out

    def brick_flags(self):
        """(flags [nbz][nby][nbx] uint8, in_use) -- the empty-space flags the next frame would use (smk_get_brick_flags)"""
        nb = (C.c_int * 3)()
        use = C.c_int(0)
        self._ck(self.L.smk_get_brick_flags(self.ctx, None, nb, C.byref(use)))
        out = np.zeros((nb[2], nb[1], nb[0]), np.uint8)
        self._ck(self.L.smk_get_brick_flags(self.ctx, out.ctypes.data_as(C.c_void_p), nb, C.byref(use)))
        return out, bool(use.value)

    def set_option(self, key, value):
        self._ck(self.L.smk_set_option(self.ctx, key.encode(), int(value)))

    # -- render
    def render(self, depth=False):
        w, h = self.size
        out = np.zeros((h, w, 4), np.float32)
        dep = np.zeros((h, w), np.float32) if depth else None
        self._ck(self.L.smk_render(self.ctx, _ptr(out), _ptr(dep)))
        return (out, dep) if depth else out

    def render_device(self, d_rgba, d_depth=None, stream=None):
        self._ck(self.L.smk_render_device(self.ctx, d_rgba, d_depth, stream))

    def composite_over_device(self, d_layers, nlayers, order, npix, d_out, stream=None):
        self._ck(self.L.smk_composite_over_device(self.ctx, d_layers, nlayers,
                                                  (C.c_int * nlayers)(*order), npix, d_out, stream))

    def last_frame_id(self):
        return int(self.L.smk_last_frame_id(self.ctx))

    def frame_failed(self, frame_id):
        """after synchronising with that frame: 1 = a streaming kernel flagged it, render it again; 0 = valid; -1 = unknown
        (never enqueued, or older than the status ring)"""
        return int(self.L.smk_frame_failed(self.ctx, int(frame_id)))

    # -- introspection
    def raycoef(self):
        rc = RayCoef()
        self._ck(self.L.smk_get_raycoef(self.ctx, C.byref(rc)))
        return rc

    def count_samples(self):
        """samples of the current frame set-up that lie inside the volume (smk_count_samples)"""
        v = C.c_double(0)
        self._ck(self.L.smk_count_samples(self.ctx, C.byref(v)))
        return int(v.value)

    def last_frame_info(self):
        k, ms, b = C.c_int(0), C.c_float(0), C.c_double(0)
        self._ck(self.L.smk_last_frame_info(self.ctx, C.byref(k), C.byref(ms), C.byref(b)))
        return k.value, ms.value, b.value

    def stat(self, name):
        v = C.c_double(0)
        self._ck(self.L.smk_get_stat(self.ctx, name.encode(), C.byref(v)))
        return v.value

    def trace(self):
        n = C.c_int(0)
        self._ck(self.L.smk_get_trace(self.ctx, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 0), 8), dtype=np.uint32)
        if n.value > 0:
            self._ck(self.L.smk_get_trace(self.ctx, out.ctypes.data, n.value, C.byref(n)))
        return out

    def timing_reset(self):
        self._ck(self.L.smk_timing_reset(self.ctx))

    def timing_read(self):
        ms, n = C.c_float(0), C.c_int(0)
        self._ck(self.L.smk_timing_read(self.ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def tf2d_effective(self, sv, sg):
        out = np.zeros((sg, sv, 4), np.uint8)
        r = C.c_float(0)
        self._ck(self.L.smk_get_tf2d_effective(self.ctx, _ptr(out), C.byref(r)))
        return out, r.value

    # -- GPU data prep
    def make_vgh_device(self, d_scalar, dtype, dims, compat, d_vgh_u8=None, d_vgh_f32=None):
        sx, sy, sz = dims
        self._ck(self.L.smk_make_vgh_device(self.ctx, d_scalar, dtype, sx, sy, sz, int(compat),
                                            d_vgh_u8, d_vgh_f32))

    def normals_vgh_device(self, d_vgh_u8, nelts, dims, blur, d_normals):
        sx, sy, sz = dims
        self._ck(self.L.smk_normals_vgh_device(self.ctx, d_vgh_u8, nelts, sx, sy, sz, int(blur),
                                               d_normals))

    def merge_fields_device(self, d_fields, nf, dims, d_out, d_normals=None):
        sx, sy, sz = dims
        self._ck(self.L.smk_merge_fields_device(self.ctx, d_fields, nf, sx, sy, sz, d_out, d_normals))

    def hist2d_device(self, d_vol_u8, nelts, dims):
        """log-scaled joint (value, gradient) histogram [g][v] of a device-resident u8 volume"""
        sx, sy, sz = dims
        out = np.zeros((256, 256), np.uint8)
        self._ck(self.L.smk_hist2d_device(self.ctx, d_vol_u8, nelts, sx, sy, sz, _ptr(out)))
        return out

    def hist2d(self, data, grid=(1, 1, 1)):
        """the same from host memory, brick by brick as MetaVolume holds it"""
        data = np.ascontiguousarray(data, np.uint8)
        nz, ny, nx, ne = data.shape
        bricks = split_bricks((nx, ny, nz), (1.0, 1.0, 1.0), grid)
        descs = (VolumeDesc * len(bricks))()
        keep = []
        for d, b in zip(descs, bricks):
            (x0, y0, z0), (bx, by, bz) = b["ipos"], b["isize"]
            sub = np.ascontiguousarray(data[z0:z0 + bz, y0:y0 + by, x0:x0 + bx])
            keep.append(sub)
            d.xiSize, d.yiSize, d.ziSize = bx, by, bz
            d.data = _ptr(sub)
        out = np.zeros((256, 256), np.uint8)
        self._ck(self.L.smk_hist2d(self.ctx, descs, len(bricks), ne, _ptr(out)))
        return out

    def synth_volume_device(self, kind, seed, dims, d_out):
        sx, sy, sz = dims
        self._ck(self.L.smk_synth_volume_device(self.ctx, kind, seed, sx, sy, sz, d_out))


def exchange_unique_id():
    """the 128-byte RCCL communicator id (rank 0 makes it, every other rank needs the same bytes)"""
    buf = (C.c_ubyte * 128)()
    if load_library().smk_exchange_unique_id(buf) != 0:
        raise SmkError("smk_exchange_unique_id failed (librccl.so.1 not loadable?)")
    return bytes(buf)


class Exchange:
    """One rank's end of the sort-last merge behind the C ABI (smk_exchange_*): direct send of tiles,
    ordered over, gather.  id = the RCCL communicator id (one process per GPU) or None (all ranks are
    contexts of this process: Exchange.connect_local + Exchange.frame_local)."""

    def __init__(self, renderer, rank, nranks, npix, id=None):
        self.L, self.r = renderer.L, renderer
        err = C.c_int(0)
        idbuf = (C.c_ubyte * 128).from_buffer_copy(id) if id is not None else None
        self.x = self.L.smk_exchange_create(renderer.ctx, rank, nranks, idbuf, npix, C.byref(err))
        if not self.x:
            raise SmkError(self.L.smk_exchange_last_error(None).decode())

    def _ck(self, rc):
        if rc != 0:
            raise SmkError(self.L.smk_exchange_last_error(self.x).decode())

    def close(self):
        if self.x:
            self.L.smk_exchange_destroy(self.x)
            self.x = None

    def partial(self, slot):
        return self.L.smk_exchange_partial(self.x, slot)

    def acquire(self, slot, render_stream=None):
        self._ck(self.L.smk_exchange_acquire(self.x, slot, render_stream))

    def rendered(self, slot, render_stream=None):
        self._ck(self.L.smk_exchange_rendered(self.x, slot, render_stream))

    def set_order(self, slot, order):
        arr = (C.c_int * len(order))(*[int(v) for v in order])
        self._ck(self.L.smk_exchange_set_order(self.x, slot, arr))

    def frame(self, slot, d_frame):
        self._ck(self.L.smk_exchange_frame(self.x, slot, d_frame))

    def wait(self, stream=None):
        self._ck(self.L.smk_exchange_wait(self.x, stream))

    @staticmethod
    def connect_local(xs):
        arr = (C.c_void_p * len(xs))(*[x.x for x in xs])
        xs[0]._ck(xs[0].L.smk_exchange_connect_local(arr, len(xs)))

    @staticmethod
    def frame_local(xs, slot, d_frame):
        arr = (C.c_void_p * len(xs))(*[x.x for x in xs])
        xs[0]._ck(xs[0].L.smk_exchange_frame_local(arr, len(xs), slot, d_frame))
