"""Seeded synthetic scenes shared by the CPU-only and the GPU parity tests (SURVEY 8d inputs at
sizes the CPU checker finishes in seconds), plus the glue that pushes one oracle.Scene into the
HIP product through its C ABI."""
import functools

import numpy as np

import oracle as O


@functools.lru_cache(maxsize=None)
def scalar_volume(n=32, seed=1, kind="spheres"):
    if kind == "spheres":
        return O.genvol_spheres(n, seed=seed)
    return O.genvol_perl(n, seed=seed)


@functools.lru_cache(maxsize=None)
def vgh_volume(n=32, seed=1):
    v = scalar_volume(n, seed)
    vgh8, vghf = O.make_vgh(v, compat=True, f32=True)
    nrm = O.normals_vgh(vgh8)
    return vgh8, vghf, nrm


def ragged_vgh(dims=(40, 24, 18), seed=3):
    rng = np.random.default_rng(seed)
    nx, ny, nz = dims
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    base = (np.sin(x * .31) + np.cos(y * .23) + np.sin(z * .41 + x * .07)) * 40 + 128
    v = np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)
    vgh8, vghf = O.make_vgh(v, compat=False, f32=True)
    return vgh8, vghf, O.normals_vgh(vgh8, blur=True)


def tf_cfg2():
    """reference default deptex ramp (NV20VolRen3D.cpp:1479-1486), opacity-corrected at rate 1"""
    d1, d2 = O.deptex_default()
    return O.copy_scale(d1, 1.0), d2


def tf_cfg3():
    """one default triangle + one ellipse LevWidget rasterised into a cleared 256^2"""
    tex = np.zeros((256, 256, 4), np.uint8)
    O.lev_rasterize(O.lev_widget("triangle"), tex)
    O.lev_rasterize(O.lev_widget("ellipse", b=(.1, .05), l=(.1, .5), r=(.45, .5), hsl=(.6, 1, .5), alpha=.6), tex)
    return tex


def tf_h(slider=0.5):
    _, d2 = O.deptex_default()
    return O.rasterize_vgh(d2.copy(), slider)


def tf3d_dense(n=16, seed=5):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, (n, n, n, 4), dtype=np.uint8)
    t[..., 3] = (t[..., 3].astype(np.float32) * 0.25).astype(np.uint8)
    return t


def tf3d_panes():
    """the reference's dense-table shape: 256 x 256 x 4 sheets (TFWidgetRen.cpp:98-100), here cfg 3's widgets in every
    sheet with the opacity scaled per sheet -- mostly transparent, which is what the kernels' occupancy shortcut lives on"""
    pane = tf_cfg3()
    t = np.stack([pane] * 4).copy()
    for h, be in enumerate((0.4, 1.0, 0.7, 0.4)):
        t[h, ..., 3] = (pane[..., 3].astype(np.float32) * be).astype(np.uint8)
    t[2, :40, :40, 3] = 0          # (a region only SOME sheets paint: the bit must stay set there)
    t[0, 200:, 200:, 3] = 90       # (... and one only the first sheet paints)
    t[3, :, 60:110, 3] = np.maximum(t[3, :, 60:110, 3], 70)   # (... and a band of values only the LAST sheet paints everywhere)
    return t


# principal axis / marching direction coverage for the slice-ring kernel: views roughly along
# +-z, +-y, +-x (the last needs the x-major copy), each a little off-axis
POSES = {
    "z+": ((0.3, 1, 0.2), 12), "z-": ((0.1, 1, 0.05), 171),
    "y+": ((1, 0.15, 0.1), 80), "y-": ((1, -0.1, 0.2), -97),
    "x+": ((0.1, 1, 0.2), 82), "x-": ((0.15, 1, -0.1), -95),
    "diag": ((1, 1, 1), 50),
}


def make_scene(kind, n=32, size=48, steps=48, pose="rot", f32=False, shade=0, third=False,
               pert=False, dims=None):
    """kind: 'cfg1' (u8 scalar, 1-D TLUT), 'cfg2' (VGH, 2-D TF), 'cfg3' (VGH, LevWidget TF),
    'cfg4' (VGH, separable (v,g)x(h) TF), 'tf3d' (dense 3-D TF)"""
    if kind == "cfg1":
        v = scalar_volume(n, 1, "perl")
        sc = O.Scene(v[..., None])
        sc.tf_mode = 0
        sc.tlut = O.tlut_volumerenderable()
    else:
        if dims:
            vgh8, vghf, nrm = ragged_vgh(dims)
        else:
            vgh8, vghf, nrm = vgh_volume(n)
        sc = O.Scene(vghf if f32 else vgh8, grad=nrm)
        sc.tf_mode = 1
        if kind == "cfg2":
            sc.tf_vg, sc.tf_h = tf_cfg2()
        elif kind == "cfg3":
            sc.tf_vg = tf_cfg3()
        elif kind == "cfg4":
            sc.tf_vg = tf_cfg3()
            sc.tf_h = tf_h(0.5)
            third = True
        elif kind == "tf3d":
            sc.tf_mode = 2
            sc.tf3d = tf3d_dense()
        elif kind == "tf3d_panes":
            sc.tf_mode = 2
            sc.tf3d = tf3d_panes()
        sc.third_axis = 1 if third else 0
        if third and sc.tf_h is None:
            sc.tf_h = tf_h(0.5)
    sc.width = sc.height = size
    sc.steps = steps
    if pose == "rot":
        sc.xform = O.rotation((1, 1, 0), 30)      # SURVEY 8d second pose
    elif pose == "back":
        sc.xform = O.rotation((0, 1, 0), 160)
    elif pose == "side":
        sc.xform = O.rotation((.2, 1, .1), 75)
    elif pose in POSES:
        sc.xform = O.rotation(*POSES[pose])
    sc.shade_mode = shade
    if pert:
        sc.noise = O.noise_tex(32)
        sc.pert_w = (.2, .1, 0, 0)
        sc.pert_s = (.2, 2.1, 4.5, 8.7)
    return sc


_DMODE = {1: "V1", 2: "V1G", 3: "VGH", 4: "V2GH"}


def push_scene(r, sc, grid=(1, 1, 1), upload=True):
    """drive the C ABI exactly as a gluvvPrimitive adapter would: init() part + draw() part"""
    if upload:
        r.upload_volume(sc.data, sc.grad, fsize=tuple(float(f) for f in sc.fsize), grid=grid,
                        dmode=_DMODE[sc.nelts])
    r.set_option("tf_raw", 1)      # scenes carry already opacity-corrected tables
    if sc.tf_mode == 0:
        r.set_tlut1d(sc.tlut)
    elif sc.tf_mode == 1:
        r.set_tf2d(sc.tf_vg, sc.tf_h if sc.third_axis else None)
    else:
        r.set_tf3d(sc.tf3d)
    r.set_clip(*(sc.clip if getattr(sc, "clip", None) else (0, None)))
    r.set_clip_plane(getattr(sc, "clip_plane", None))
    r.set_camera(sc.mv(), sc.frustum, (sc.znear, 20.0), sc.width, sc.height)
    r.set_sampling(sc.sample_rate, sc.steps, 1.0, 1)
    mode = {0: "none", 1: "r8k" if sc.use_spec else "r8k_diff", 2: "nv20" if sc.use_spec else "nv20_diff"}[sc.shade_mode]
    r.set_shading(mode, sc.light_pos, sc.eye, sc.at, sc.xform, sc.intens)
    if sc.noise is not None and any(sc.pert_w):
        r.set_perturb(sc.noise, sc.pert_w, sc.pert_s)
    else:
        r.set_perturb(None, None, None)
    if getattr(sc, "shadow", None):
        r.set_shadow(1, *sc.shadow)
    else:
        r.set_shadow(0)
