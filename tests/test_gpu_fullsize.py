"""Parity at BASELINE.json's full sizes, through size-independent properties:
  * the two ray-march kernels (independent implementations of the same fma chains) produce
    bit-identical 1024 x 1024 x 512-plane frames on the 512^3 (configs[1]) and 1024^3
    (north star) f32 VGH volumes;
  * the CPU checker agrees (<= 1e-4) on a random sample of rays of the full-size frame;
  * two brick shards rendered separately and composited in visibility order == the whole.
Volumes are synthesised on the GPU exactly as bench.py does (smk_prep.hip, itself bit-exact
against the checker in test_gpu_prep.py)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("smk_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _frame(r, size, kernel):
    import torch
    out = torch.zeros((size * size, 4), dtype=torch.float32, device="cuda")
    r.set_option("kernel", kernel)
    r.render_device(out.data_ptr(), None, None)
    torch.cuda.synchronize()
    assert r.stat("slab_status") == 0
    assert r.last_frame_info()[0] == kernel
    return out.cpu().numpy().reshape(size, size, 4)


@pytest.mark.parametrize("n,workload", [(512, "cfg3"), (1024, "cfg4")])
def test_full_size_frames(gpu_renderer_factory, O, n, workload):
    import torch
    b = _bench()
    size, planes = 1024, 512
    r = gpu_renderer_factory()
    try:
        vghf, nrm = b.make_volume(r, n)
        r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
        xform, mv = b.configure(r, workload, n, size, planes)
        g = _frame(r, size, 1)
        s = _frame(r, size, 2)
        assert g[..., 3].max() > 0.5
        assert np.array_equal(g, s), "gather and slice-ring kernels differ at full size: %g" % np.abs(g - s).max()
        assert r.stat("slab_failures") == 0 and r.stat("slab_retries") == 0
        # CPU checker on random rays of the same frame (same effective table, same matrix); the checker
        # works per ray, so the 1024^3 north-star frame costs it memory for the volume (13 GB as f32 on
        # the host) and a few seconds: 600 rays at 512^3, 300 at 1024^3
        tf_eff, _ = r.tf2d_effective(256, 256)
        sc = O.Scene(vghf.cpu().numpy(), grad=nrm.cpu().numpy())
        del vghf, nrm
        torch.cuda.empty_cache()
        sc.tf_mode, sc.tf_vg = 1, tf_eff
        if workload == "cfg4":
            sc.tf_h, sc.third_axis = np.load(os.path.join(ROOT, "tests", "golden", "tf_h_slider05.npy")), 1
        sc.width = sc.height = size
        sc.steps = planes
        sc.xform = [float(v) for v in xform.T.reshape(-1)]
        sc.mv_override = mv
        sc.shade_mode, sc.use_spec = 1, 1
        sc.frustum = b.FRUSTUM
        rng = np.random.default_rng(7)
        pix = rng.integers(0, size, size=(600 if n == 512 else 300, 2)).astype(np.int32)   # (i, j)
        ref = sc.render_pixels(pix)
        got = s[pix[:, 1], pix[:, 0]]
        assert ref[:, 3].max() > 0.5
        assert np.abs(got - ref).max() <= 1e-4
        vghf = nrm = None
        del vghf, nrm
        torch.cuda.empty_cache()
    finally:
        r.close()


def test_full_size_two_shards_composite_to_whole(gpu_renderer_factory):
    import torch
    b = _bench()
    n, size, planes = 512, 1024, 512
    whole = None
    rs = []
    try:
        layers = torch.zeros((2, size * size, 4), dtype=torch.float32, device="cuda")
        for rank in (None, 0, 1):
            r = gpu_renderer_factory()
            rs.append(r)
            if rank is not None:
                r.set_shard(rank, 2)
            vghf, nrm = b.make_volume(r, n)
            r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
            del vghf, nrm
            b.configure(r, "cfg3", n, size, planes)
            if rank is None:
                whole = _frame(r, size, 2)
            else:
                r.set_option("kernel", 0)
                r.render_device(layers[rank].data_ptr(), None, None)
        torch.cuda.synchronize()
        for r in rs:
            assert r.stat("slab_status") == 0
        order = rs[1].shard_order(2)
        out = torch.zeros((size * size, 4), dtype=torch.float32, device="cuda")
        rs[1].composite_over_device(layers.data_ptr(), 2, order, size * size, out.data_ptr(), None)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(size, size, 4)
        # "over" of two partial sums re-associates the front-to-back blend: equal up to fp32 rounding
        assert np.abs(got - whole).max() <= 2e-5
    finally:
        for r in rs:
            r.close()


@pytest.mark.parametrize("weights,halo", [((.2, .1, 0, 0), 80), ((.02, .01, 0, 0), 16)])
def test_full_size_cfg5_multi_field_perturbed(gpu_renderer_factory, O, weights, halo):
    """BASELINE config 5 at its full size: two 512^3 fields merged on the GPU (mergeMV + addG),
    dense 3-D transfer function, testPert's noise-perturbed fetch, 1024^2 x 1024 planes (gather
    kernel) -- at the weights SURVEY 8(d) states, (.2, .1): displacements of up to 77 voxels, and at a
    tenth of them.  The CPU checker agrees on 300 random rays of the frame, and the frame rendered as two
    brick shards (halo wide enough for the perturbation: 0.5 * 0.3 * 512 + the texel pair) composites back to it."""
    import torch
    import _scenes as S
    n, size, planes = 512, 1024, 1024
    b = _bench()
    rs = []
    try:
        r = gpu_renderer_factory()
        rs.append(r)
        fields = torch.empty((n, n, n, 2), dtype=torch.uint8, device="cuda")
        one = torch.empty((n, n, n), dtype=torch.uint8, device="cuda")
        for e, seed in enumerate((1, 2)):
            r.synth_volume_device(0, seed, (n, n, n), one.data_ptr())
            fields[..., e] = one if e == 0 else one.flip(2)       # the second field: another seed, mirrored
        del one
        merged = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
        nrm = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
        r.merge_fields_device(fields.data_ptr(), 2, (n, n, n), merged.data_ptr(), nrm.data_ptr())
        del fields
        xform = b.rotation((1, 1, 0), 30)
        mv = b.modelview(xform, (1.0, 1.0, 1.0))
        noise = O.noise_tex(32)
        tf3d = S.tf3d_dense()

        def setup(rr):
            rr.upload_volume_device(merged.data_ptr(), (n, n, n), 3, 0, nrm.data_ptr(), dmode="V2G")
            rr.set_option("tf_raw", 1)
            rr.set_tf3d(tf3d)
            rr.set_camera(mv, b.FRUSTUM, (1.0, 20.0), size, size)
            rr.set_sampling(0.0, planes, 1.0, 1)
            rr.set_shading("r8k", b.LIGHT, b.EYE, b.AT, [float(v) for v in xform.T.reshape(-1)], b.INTENS)
            rr.set_perturb(noise, weights, (.2, 2.1, 4.5, 8.7))
            rr.set_option("kernel", 0)

        setup(r)
        whole = torch.zeros((size * size, 4), dtype=torch.float32, device="cuda")
        r.render_device(whole.data_ptr(), None, None)
        torch.cuda.synchronize()
        assert r.last_frame_info()[0] == 1
        import time
        t0 = time.perf_counter()
        for _ in range(3):
            r.render_device(whole.data_ptr(), None, None)
        torch.cuda.synchronize()
        print("cfg5 at full size, weights %s, one GPU, gather kernel: %.2f ms per 1024 x 1024 x 1024-plane frame" % (weights[:2], (time.perf_counter() - t0) / 3 * 1e3))
        w = whole.cpu().numpy().reshape(size, size, 4)
        assert w[..., 3].max() > 0.5
        sc = O.Scene(merged.cpu().numpy(), grad=nrm.cpu().numpy())
        sc.tf_mode, sc.tf3d = 2, tf3d
        sc.width = sc.height = size
        sc.steps = planes
        sc.xform = [float(v) for v in xform.T.reshape(-1)]
        sc.mv_override = mv
        sc.shade_mode, sc.use_spec = 1, 1
        sc.frustum = b.FRUSTUM
        sc.noise, sc.pert_w, sc.pert_s = noise, weights, (.2, 2.1, 4.5, 8.7)
        rng = np.random.default_rng(8)
        pix = rng.integers(0, size, size=(300, 2)).astype(np.int32)
        ref = sc.render_pixels(pix)
        assert ref[:, 3].max() > 0.5
        assert np.abs(w[pix[:, 1], pix[:, 0]] - ref).max() <= 1e-4
        del sc
        layers = torch.zeros((2, size * size, 4), dtype=torch.float32, device="cuda")
        for rank in (0, 1):
            rr = gpu_renderer_factory()
            rs.append(rr)
            rr.set_option("halo", halo)       # half the summed weights of 512 voxels of displacement + the texel pair
            rr.set_shard(rank, 2)
            setup(rr)
            rr.render_device(layers[rank].data_ptr(), None, None)
        torch.cuda.synchronize()
        out = torch.zeros((size * size, 4), dtype=torch.float32, device="cuda")
        rs[1].composite_over_device(layers.data_ptr(), 2, rs[1].shard_order(2), size * size, out.data_ptr(), None)
        torch.cuda.synchronize()
        assert (out - whole).abs().max().item() <= 2e-5
    finally:
        for r in rs:
            r.close()


def test_full_size_shadows_two_marches_equal_a_launch_per_slice(gpu_renderer_factory):
    """BASELINE config 3 at full size with the shadow check box on (512 half-angle slices, 512^2 light buffer, the light
    up-left of the eye: the slices run away from the viewer): the light march + the eye pass on either ray-marcher give the
    frame and the light buffer of 512 per-slice launches, bit for bit (the same operations per pixel and texel in the same
    order; the small-scene tests compare both forms with the CPU checker)."""
    b = _bench()
    n, size, planes = 512, 1024, 512
    r = gpu_renderer_factory()
    try:
        vghf, nrm = b.make_volume(r, n)
        r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
        del vghf, nrm
        xform, _ = b.configure(r, "cfg3", n, size, planes)
        r.set_shading("r8k", (3.0, 4.0, -3.0), b.EYE, b.AT, [float(v) for v in xform.T.reshape(-1)], b.INTENS)
        r.set_shadow(1, 1024, 0.5)
        assert r.shadowcoef().front_to_back == 1
        s = _frame(r, size, 2)
        ls = r.light_buffer()
        g = _frame(r, size, 1)
        lg = r.light_buffer()
        r.set_option("shadow_march", 0)
        import torch
        out = torch.zeros((size * size, 4), dtype=torch.float32, device="cuda")
        r.set_option("kernel", 0)
        r.render_device(out.data_ptr(), None, None)
        torch.cuda.synchronize()
        assert r.last_frame_info()[0] == 3
        p = out.cpu().numpy().reshape(size, size, 4)
        lp = r.light_buffer()
        assert p[..., 3].max() > 0.5 and lp[..., 3].max() > 0.5
        assert np.array_equal(s, g) and np.array_equal(s, p)
        assert np.array_equal(ls, lp) and np.array_equal(lg, lp)
    finally:
        r.set_option("shadow_march", 1)
        r.set_shadow(0)
        r.close()
