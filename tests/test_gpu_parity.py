"""GPU parity: the HIP product (through its C ABI) against the CPU checker on the same seeded
inputs.  Tolerance (stated): max-abs <= 1e-4 on premultiplied fp32 RGBA;
sample positions are bit-identical by construction, so the observed error is ~1e-6."""
import numpy as np
import pytest

from _scenes import make_scene, push_scene

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _check(r, sc, grid=(1, 1, 1), tol=TOL, depth=False):
    push_scene(r, sc, grid)
    ref = sc.render(depth=depth)
    got = r.render(depth=depth)
    if depth:
        (ref, rd), (got, gd) = ref, got
        fin = np.isfinite(rd)
        assert np.array_equal(fin, np.isfinite(gd))
        assert np.abs(rd[fin] - gd[fin]).max() <= 1e-4
    err = np.abs(ref - got).max()
    assert ref[..., 3].max() > 0.05, "scene renders nothing: test is vacuous"
    assert err <= tol, f"max abs err {err}"
    return err


@pytest.fixture(scope="module")
def R(gpu_renderer_factory):
    r = gpu_renderer_factory()
    yield r
    r.close()


def test_raycoef_bits(R):
    """host-side sample placement is bit-identical to the checker's"""
    sc = make_scene("cfg2")
    push_scene(R, sc)
    a, b = R.raycoef(), sc.raycoef()
    for f, _ in a._fields_:
        va, vb = getattr(a, f), getattr(b, f)
        va = list(va) if hasattr(va, "__len__") else [va]
        vb = list(vb) if hasattr(vb, "__len__") else [vb]
        assert va == vb, f


@pytest.mark.parametrize("pose", ["id", "rot", "back", "side"])
def test_cfg1_scalar_tlut(R, pose):
    _check(R, make_scene("cfg1", pose=pose))


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("pose", ["id", "rot"])
def test_cfg2_vgh_2dtf(R, f32, pose):
    _check(R, make_scene("cfg2", pose=pose, f32=f32), depth=True)


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("shade", [0, 1, 2])
def test_cfg3_phong_levwidget(R, f32, shade):
    _check(R, make_scene("cfg3", f32=f32, shade=shade))


@pytest.mark.parametrize("f32", [False, True])
def test_cfg4_third_axis(R, f32):
    _check(R, make_scene("cfg4", f32=f32, shade=1))


@pytest.mark.parametrize("f32", [False, True])
def test_dense_tf3d(R, f32):
    _check(R, make_scene("tf3d", f32=f32, shade=1))


@pytest.mark.parametrize("pert", [False, True])
def test_sparse_3d_table_of_the_references_shape(R, pert):
    """256 x 256 x 4 sheets, mostly transparent: the occupancy shortcut (a folded bit per (v, g) texel quad) skips
    lookups without changing a bit -- with the perturbed fetch this is BASELINE config 5's classification"""
    sc = make_scene("tf3d_panes", shade=1, pert=pert, third=True)
    _check(R, sc)
    assert (sc.render()[..., 3] > 0).mean() > 0.2


def test_perturbation(R):
    _check(R, make_scene("cfg3", pert=True, shade=1))


def test_sample_rate_mode(R):
    sc = make_scene("cfg2")
    sc.steps, sc.sample_rate = 0, 1.7
    _check(R, sc)


def test_ragged_dims_and_nonsquare_window(R):
    sc = make_scene("cfg2", dims=(40, 24, 18), shade=1, pose="side")
    sc.width, sc.height = 57, 33
    _check(R, sc)


@pytest.mark.parametrize("grid", [(2, 2, 2), (1, 2, 4)])
def test_bricked_equals_unbricked(R, grid):
    """bricks are re-assembled with global addressing: no seams"""
    _check(R, make_scene("cfg3", shade=1), grid=grid)


def test_empty_and_degenerate(R):
    sc = make_scene("cfg2")
    sc.trans = (5, 0, 0)           # volume outside the frustum: every ray misses
    push_scene(R, sc)
    assert np.abs(R.render()).max() == 0.0
    sc = make_scene("cfg2")
    sc.steps = 1
    _check(R, sc, tol=1e-6) if sc.render()[..., 3].max() > 0.05 else None


@pytest.mark.parametrize("world", [1, 8])
def test_cfg5_multi_field_dense_tf_and_perturbation(gpu_renderer_factory, O, world):
    """BASELINE config 5 in small: two co-registered fields merged on the GPU (value 1, value 2,
    summed-gradient magnitude + normals; MetaVolume::mergeMV), the dense 3-D transfer function over
    those three axes, the noise-perturbed fetch of testPert, R8k shading -- unsharded and as one of
    eight brick shards with the halo the perturbation needs."""
    import torch
    from simian_spacemonkey_amd import sortlast
    import _scenes as S
    nx, ny, nz = 40, 36, 32
    rng = np.random.default_rng(77)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    r2 = ((x - 20) ** 2 + (y - 18) ** 2 + (z - 16) ** 2) ** .5
    f = np.stack([np.clip(255 - r2 * 11 + rng.normal(0, 3, x.shape), 0, 255),
                  np.clip(np.sin(x * .35) * np.cos(y * .3) * 90 + 120 + rng.normal(0, 3, x.shape), 0, 255)], -1).astype(np.uint8)
    R = gpu_renderer_factory()
    try:
        d_f = torch.from_numpy(f).cuda()
        merged = torch.zeros((nz, ny, nx, 3), dtype=torch.uint8, device="cuda")
        nrm = torch.zeros((nz, ny, nx, 3), dtype=torch.uint8, device="cuda")
        R.merge_fields_device(d_f.data_ptr(), 2, (nx, ny, nz), merged.data_ptr(), nrm.data_ptr())
        ref_m, ref_n = O.merge_addg(f)
        assert np.array_equal(merged.cpu().numpy(), ref_m) and np.array_equal(nrm.cpu().numpy(), ref_n)
        sc = O.Scene(ref_m, grad=ref_n)
        sc.tf_mode, sc.tf3d = 2, S.tf3d_dense()
        sc.xform = O.rotation((1, 1, 0), 30)
        sc.width = sc.height = 72
        sc.steps, sc.shade_mode = 96, 1
        sc.noise, sc.pert_w, sc.pert_s = O.noise_tex(32), (.05, .03, 0, 0), (.2, 2.1, 4.5, 8.7)
        rank = 5
        if world > 1:
            R.set_option("halo", 6)
            R.set_shard(rank, world)
            sc.region = sortlast.shard_region(sc.dims, rank, world)
        # the volume goes from the merge kernel's output straight into the renderer: no host copy
        R.upload_volume_device(merged.data_ptr(), (nx, ny, nz), 3, 0, nrm.data_ptr(), dmode="V2G")
        S.push_scene(R, sc, upload=False)
        R.set_option("kernel", 0)
        img = R.render()
        ref = sc.render()
        assert ref[..., 3].max() > 0.05
        assert np.abs(img - ref).max() <= 1e-4
    finally:
        R.close()


@pytest.mark.parametrize("kind,f32,shade", [("cfg1", False, 0), ("cfg3", True, 1)])
def test_blend_orders_and_mip(gpu_renderer_factory, O, kind, f32, shade):
    """The reference's three framebuffer blends: front to back (R8kVolRen3D.cpp:1441-1449, the
    product's default), back to front (VolumeRenderer.cpp:590 -- the scalar path's actual order,
    NV20VolRen3D.cpp:930) and GL_MAX (gluvvShadeMIP, NV20VolRen3D.cpp:158-163).  Each against the CPU
    checker's own implementation of that blend (1e-4); and the two 'over' orders against each other: they
    are the same operator evaluated from opposite ends, equal up to fp32 rounding (stated: 2e-6)."""
    sc = make_scene(kind, n=32, size=56, steps=64, pose="rot", f32=f32, shade=shade)
    R = gpu_renderer_factory()
    try:
        push_scene(R, sc)
        R.set_option("kernel", 0)
        frames = {}
        for name, mode in (("ftb", 0), ("btf", 1), ("max", 2)):
            R.set_blend(name)
            img, dep = R.render(depth=True)
            ref, rdep = sc.render(blend=mode, depth=True)
            assert ref[..., 3].max() > 0.05
            assert np.abs(img - ref).max() <= 1e-4, name
            hit = np.isfinite(rdep)
            assert np.array_equal(np.isfinite(dep), hit) and np.abs(dep[hit] - rdep[hit]).max() <= 1e-4, name
            frames[name] = img
        assert np.abs(frames["ftb"] - frames["btf"]).max() <= 2e-6
        assert np.abs(frames["ftb"] - sc.render(blend=1)).max() <= 2e-6     # FTB frame vs the checker's back-to-front one
        assert (frames["max"][..., 3] <= frames["ftb"][..., 3] + 1e-6).all()  # the largest sample alpha never exceeds the accumulated one
        if kind == "cfg3":   # GL_MAX on the slice-ring kernel: bit-identical to the gather kernel's
            R.set_blend("max")
            R.set_option("kernel", 1)
            a = R.render()
            R.set_option("kernel", 2)
            b = R.render()
            assert R.last_frame_info()[0] == 2 and np.array_equal(a, b)
            # back to front: the slice-ring kernel composites such a frame FRONT TO BACK (its slices stream one way; "over" is
            # associative) -- the gather kernel's back-to-front walk up to the association of the blend
            R.set_blend("btf")
            b = R.render()
            assert R.last_frame_info()[0] == 2
            R.set_option("kernel", 1)
            a = R.render()
            assert np.abs(a - b).max() <= 2e-5
    finally:
        R.close()
