"""GPU data-prep kernels (VGH synthesis, normals) against the CPU checker: integer/byte outputs,
so the bar is bit-exact (genVGH/main.cpp:56-182, VectorMath.h:874-899, 1133-1148, 1217-1281)."""
import numpy as np
import pytest

from _scenes import scalar_volume

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R(gpu_renderer_factory):
    r = gpu_renderer_factory()
    yield r
    r.close()


def _dev(t):
    import torch
    return torch.from_numpy(np.ascontiguousarray(t)).cuda()


@pytest.mark.parametrize("compat", [1, 0])
@pytest.mark.parametrize("dims", [(32, 32, 32), (37, 21, 12)])
def test_make_vgh_u8_bit_exact(R, O, compat, dims):
    import torch
    nx, ny, nz = dims
    rng = np.random.default_rng(7)
    if dims == (32, 32, 32):
        v = scalar_volume(32, 1)
    else:
        z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        v = np.clip(np.sin(x * .3) * 60 + np.cos(y * .2 + z * .5) * 50 + 128 + rng.normal(0, 4, x.shape), 0, 255).astype(np.uint8)
        v[2:5, 3:9, 4:11] = 77     # flat patch: zero gradient -> NaN second derivative path
    ref8, reff = O.make_vgh(v, compat=bool(compat), f32=True)
    d = _dev(v)
    o8 = torch.zeros((nz, ny, nx, 3), dtype=torch.uint8, device="cuda")
    of = torch.zeros((nz, ny, nx, 3), dtype=torch.float32, device="cuda")
    R.make_vgh_device(d.data_ptr(), 0, dims, compat, o8.data_ptr(), of.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(o8.cpu().numpy(), ref8)
    assert np.array_equal(of.cpu().numpy(), reff)


def test_make_vgh_f32_input_bit_exact(R, O):
    import torch
    rng = np.random.default_rng(11)
    v = rng.random((14, 18, 22)).astype(np.float32)
    ref8 = O.make_vgh(v, compat=True)
    o8 = torch.zeros((14, 18, 22, 3), dtype=torch.uint8, device="cuda")
    R.make_vgh_device(_dev(v).data_ptr(), 1, (22, 18, 14), 1, o8.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.array_equal(o8.cpu().numpy(), ref8)


@pytest.mark.parametrize("blur", [0, 1])
def test_normals_bit_exact(R, O, blur):
    import torch
    vgh = O.make_vgh(scalar_volume(32, 1))
    ref = O.normals_vgh(vgh, blur=bool(blur))
    out = torch.zeros((32, 32, 32, 3), dtype=torch.uint8, device="cuda")
    R.normals_vgh_device(_dev(vgh).data_ptr(), 3, (32, 32, 32), blur, out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)


def test_synth_volume_is_deterministic_and_nontrivial(R):
    import torch
    a = torch.zeros((48, 40, 56), dtype=torch.uint8, device="cuda")
    b = torch.zeros_like(a)
    R.synth_volume_device(0, 1, (56, 40, 48), a.data_ptr())
    R.synth_volume_device(0, 1, (56, 40, 48), b.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    h = a.cpu().numpy()
    assert h.max() > 200 and h.min() == 0 and 10 < h.mean() < 200


def test_device_upload_matches_host_upload(R, O):
    """smk_upload_volume_device (no PCIe copy) packs the same HBM image as the host path"""
    import torch
    from _scenes import make_scene, push_scene
    sc = make_scene("cfg3", f32=True, shade=1)
    push_scene(R, sc)
    a = R.render()
    dv, dg = _dev(sc.data), _dev(sc.grad)
    R.upload_volume_device(dv.data_ptr(), sc.dims, 3, 1, dg.data_ptr(),
                           fsize=tuple(float(f) for f in sc.fsize))
    push_scene(R, sc, upload=False)
    b = R.render()
    assert np.array_equal(a, b)


def test_random_sizes_bit_exact(R, O):
    """ragged sizes down to the smallest the entry points take (3 voxels per axis: the derivative
    stencils need an interior): VGH synthesis and normals (with and without the 27-tap blur) stay
    bit-exact; smaller volumes are refused, not mis-computed"""
    import torch
    rng = np.random.default_rng(99)
    with pytest.raises(Exception, match="dims >= 3"):
        R.make_vgh_device(_dev(np.zeros((2, 5, 5), np.uint8)).data_ptr(), 0, (5, 5, 2), 1,
                          torch.zeros((2, 5, 5, 3), dtype=torch.uint8, device="cuda").data_ptr(), None)
    sizes = [(3, 3, 3), (3, 17, 4), (41, 3, 7), (5, 4, 3)]
    sizes += [tuple(int(rng.integers(3, 45)) for _ in range(3)) for _ in range(10)]
    for dims in sizes:
        nx, ny, nz = dims
        v = rng.integers(0, 256, size=(nz, ny, nx), dtype=np.uint8)
        if nx > 4 and ny > 4 and nz > 4:
            v[1:4, 1:4, 1:4] = 200     # flat patch: zero gradient
        for compat in (1, 0):
            ref8, reff = O.make_vgh(v, compat=bool(compat), f32=True)
            o8 = torch.zeros((nz, ny, nx, 3), dtype=torch.uint8, device="cuda")
            of = torch.zeros((nz, ny, nx, 3), dtype=torch.float32, device="cuda")
            R.make_vgh_device(_dev(v).data_ptr(), 0, dims, compat, o8.data_ptr(), of.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(o8.cpu().numpy(), ref8), (dims, compat)
            # (one interior voxel: the normalisation range is 0 and both sides emit the same NaNs)
            assert np.array_equal(of.cpu().numpy(), reff, equal_nan=True), (dims, compat)
        vgh = O.make_vgh(v)
        for blur in (0, 1):
            ref = O.normals_vgh(vgh, blur=bool(blur))
            out = torch.zeros((nz, ny, nx, 3), dtype=torch.uint8, device="cuda")
            R.normals_vgh_device(_dev(vgh).data_ptr(), 3, dims, blur, out.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), ref), (dims, blur)


def test_hist2d_matches_the_reference_arithmetic(R, O):
    """MetaVolume::hist2D on the GPU (16-bit LDS bins, flushed before they can overflow) against the
    CPU restatement that counts in float bins as the reference does: every byte equal."""
    import torch
    rng = np.random.default_rng(17)
    # a VGH-like volume: most voxels in a few bins (air), a tail spread over many
    nz, ny, nx = 41, 53, 67
    vgh = np.zeros((nz, ny, nx, 3), np.uint8)
    vgh[..., 0] = np.clip(rng.normal(20, 6, (nz, ny, nx)), 0, 255)
    vgh[..., 1] = np.clip(rng.exponential(9, (nz, ny, nx)), 0, 255)
    vgh[10:30, 10:40, 5:60, 0] = rng.integers(0, 256, (20, 30, 55))
    vgh[10:30, 10:40, 5:60, 1] = rng.integers(0, 256, (20, 30, 55))
    vgh[:8] = (3, 0, 85)                                   # a uniform slab: whole waves in one bin
    ref = O.hist2d(vgh)
    got = R.hist2d_device(_dev(vgh).data_ptr(), 3, (nx, ny, nz))
    assert ref.max() == 255 and np.count_nonzero(ref) > 1000
    assert np.array_equal(got, ref)
    assert np.array_equal(R.hist2d(vgh, grid=(1, 1, 1)), ref)
    # bricked host volume: MetaVolume::brick drops remainder voxels, so compare on what the bricks hold
    even = np.ascontiguousarray(vgh[:40, :52, :66])
    assert np.array_equal(R.hist2d(even, grid=(2, 2, 2)), O.hist2d(even))
    two = np.ascontiguousarray(vgh[..., :2])               # (value, gradient) pairs only
    assert np.array_equal(R.hist2d_device(_dev(two).data_ptr(), 2, (nx, ny, nz)), O.hist2d(two))
    with pytest.raises(Exception, match="not implemented"):
        R.hist2d_device(_dev(vgh[..., :1]).data_ptr(), 1, (nx, ny, nz))
    assert O.hist2d(np.ascontiguousarray(vgh[..., :1])) is None


def test_hist2d_float_bins_stop_at_two_to_the_24(R, O):
    """the reference counts in float: a bin holding more than 2^24 voxels reads 2^24.  A 272^3
    volume with 19 M voxels in one bin and 1.1 M in another pins that on both sides."""
    import torch
    n = 272
    vol = torch.zeros((n, n, n, 2), dtype=torch.uint8, device="cuda")
    vol[:15] = torch.tensor([7, 9], dtype=torch.uint8, device="cuda")
    got = R.hist2d_device(vol.data_ptr(), 2, (n, n, n))
    ref = O.hist2d(vol.cpu().numpy())
    assert np.array_equal(got, ref)
    big, small = (n - 15) * n * n, 15 * n * n
    assert big > 2 ** 24
    want_small = int(np.float32(np.float32(np.log(float(small))) / np.float32(np.log(float(2 ** 24))) * 255))
    assert got[0, 0] == 255 and got[9, 7] == want_small and np.count_nonzero(got) == 2


@pytest.mark.parametrize("nf,dims", [(2, (23, 17, 12)), (1, (9, 30, 7)), (3, (16, 16, 16))])
def test_merge_fields_bit_exact(R, O, nf, dims):
    """MetaVolume::mergeMV + addG on the GPU: interleaved fields, gradient magnitude of the summed
    per-field differences scaled by its maximum, and the normal bytes -- every byte as the CPU
    restatement computes it"""
    import torch
    nx, ny, nz = dims
    rng = np.random.default_rng(31 + nf)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    f = np.stack([np.clip(np.sin(x * (.2 + .1 * e)) * 70 + np.cos(y * .3 + z * (.2 + .05 * e)) * 50 + 128 +
                          rng.normal(0, 5, x.shape), 0, 255) for e in range(nf)], axis=-1).astype(np.uint8)
    ref, refn = O.merge_addg(f)
    out = torch.zeros((nz, ny, nx, nf + 1), dtype=torch.uint8, device="cuda")
    nrm = torch.zeros((nz, ny, nx, 3), dtype=torch.uint8, device="cuda")
    R.merge_fields_device(_dev(f).data_ptr(), nf, dims, out.data_ptr(), nrm.data_ptr())
    torch.cuda.synchronize()
    assert ref[..., nf].max() == 255
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(nrm.cpu().numpy(), refn)


@pytest.mark.parametrize("dims,seed", [((64, 64, 64), 1), ((40, 24, 33), 2)])
def test_genvol_spheres_on_the_gpu_bit_exact(R, O, dims, seed):
    """The product's test-volume generator (smk_synth_volume_device kind 1 = `genvol -spheres 4 -p 10
    -pscale .7 -pwrap 3 3 3 -pabs -blur -bw 1 1 1 .7`, genvol/main.cpp:212-256, 334-430 + perlin.c) against
    the CPU checker's restatement, whose Perlin noise is pinned against the reference's own perlin.c
    (tests/test_perlin_ref.py): every byte equal, double-precision noise and float blur sums included.
    This is the volume bench.py and the full-size tests render."""
    import torch
    sx, sy, sz = dims
    out = torch.zeros((sz, sy, sx), dtype=torch.uint8, device="cuda")
    R.synth_volume_device(1, seed, dims, out.data_ptr())
    ref = O.genvol_spheres(dims, seed=seed)
    got = out.cpu().numpy()
    assert ref.max() > 150 and len(np.unique(ref)) > 8
    assert np.array_equal(got, ref), "differs in %d voxels" % int((got != ref).sum())
