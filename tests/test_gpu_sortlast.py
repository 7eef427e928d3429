"""Sort-last on the GPU: P shard contexts on one device render their regions, the HIP
compositor merges them in the order smk_shard_order returns; result == unsharded frame and
== the CPU checker.  (RCCL itself is exercised by bench.py --gpus N; the exchange plumbing by
tests/test_sortlast_cpu.py.)"""
import numpy as np
import pytest

from _scenes import make_scene, push_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("world,size", [(2, 48), (4, 48), (8, 48), (8, 45)])
@pytest.mark.parametrize("pose", ["rot", "back"])
def test_sharded_frames_composite_to_whole(gpu_renderer_factory, smk, world, size, pose, kernel):
    """(size 45, pose "back": the centre row of an odd viewport under a rotation about y runs exactly ALONG
    the shard boundary y = N/2 -- fma(m, B, A) rounds onto the face for every plane while (lo - A) / B says the
    ray leaves at once; both kernels bracketed that ray's plane range too tightly until round 2)"""
    import torch
    from simian_spacemonkey_amd import sortlast
    sc = make_scene("cfg3", n=32, size=size, steps=48, pose=pose, f32=True, shade=1)
    ref = sc.render()
    npix = sc.width * sc.height
    layers = torch.zeros((world, npix, 4), dtype=torch.float32, device="cuda")
    rs = []
    try:
        for r in range(world):
            R = gpu_renderer_factory()
            rs.append(R)
            R.set_shard(r, world)
            push_scene(R, sc)
            R.set_option("kernel", kernel)      # 0: auto (the first frame of a configuration = slice-ring kernel), 1: gather
            R.render_device(layers[r].data_ptr(), None, None)
        torch.cuda.synchronize()
        for R in rs:   # asynchronous frames report kernel-side failures through the status word
            assert R.stat("slab_status") == 0
        order = rs[0].shard_order(world)
        # the C++ BSP rule agrees with the Python mirror used by the CPU tests
        mv = np.array(sc.mv()).reshape(4, 4).T
        eye = np.linalg.inv(mv)[:3, 3]
        eye_vox = [eye[a] * sc.dims[a] / float(sc.fsize[a]) - 0.5 for a in range(3)]
        assert order == sortlast.front_to_back_order(eye_vox, sc.dims, world)
        out = torch.zeros((npix, 4), dtype=torch.float32, device="cuda")
        rs[0].composite_over_device(layers.data_ptr(), world, order, npix, out.data_ptr(), None)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(sc.height, sc.width, 4)
        assert np.abs(got - ref).max() <= 1e-4
        # each shard alone equals the checker restricted to that region
        sc.region = sortlast.shard_region(sc.dims, world - 1, world)
        part = layers[world - 1].cpu().numpy().reshape(sc.height, sc.width, 4)
        assert np.abs(part - sc.render()).max() <= 1e-4
    finally:
        for R in rs:
            R.close()


def test_perturbation_needs_halo_when_sharded(gpu_renderer_factory, smk):
    sc = make_scene("cfg3", n=32, pert=True, shade=1)
    R = gpu_renderer_factory()
    try:
        R.set_shard(0, 2)
        push_scene(R, sc)
        with pytest.raises(smk.SmkError, match="halo"):
            R.render()
    finally:
        R.close()
    R = gpu_renderer_factory()
    try:
        R.set_shard(1, 2)
        R.set_option("halo", 8)
        push_scene(R, sc)
        from simian_spacemonkey_amd import sortlast
        sc.region = sortlast.shard_region(sc.dims, 1, 2)
        assert np.abs(R.render() - sc.render()).max() <= 1e-4
    finally:
        R.close()


def test_bench_multi_rank_plumbing_rehearsal():
    """bench.py's N>1 path (shards, two frames in flight, exchange, ordered over, gather) run as
    two ranks that share this one GPU, the layers travelling over gloo through host memory: the
    merged frame must equal the unsharded frame bench.py renders beside it."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # plain `python bench.py --gpus 2`, as the driver runs it: bench.py starts its own two ranks and, on a
    # box with one GPU, labels the run a rehearsal
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SMK_BENCH_REHEARSE")}
    cmd = [sys.executable, os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--volume", "96", "--size", "192", "--planes", "96"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and "REHEARSAL" in out["data"]
    assert out["sortlast_check"]["alpha_mean"] > 0.01
    assert out["sortlast_check"]["max_abs_err_vs_unsharded_frame"] <= 2e-5
    assert out["slab_failures"] == 0 and out["frames_repaired"] == 0


@pytest.mark.parametrize("world", [2, 8])
def test_exchange_behind_the_c_abi_in_process(gpu_renderer_factory, smk, world):
    """smk_exchange_* with the in-process transport: `world` shard contexts of this process render two
    frames in flight into the exchange's own buffers; direct send of tiles, ordered over, gather -- all
    in C.  The merged frames equal the CPU checker's unsharded ones (and differ from each other: two
    poses, so a slot mix-up would show).  An odd pixel count leaves the last tile short."""
    import torch
    scs = [make_scene("cfg3", n=32, size=45, steps=48, pose=p, f32=True, shade=1) for p in ("rot", "back", "side")]
    npix = scs[0].width * scs[0].height
    rs, xs = [], []
    try:
        for r in range(world):
            R = gpu_renderer_factory()
            rs.append(R)
            R.set_shard(r, world)
            push_scene(R, scs[0])
            xs.append(smk.binding.Exchange(R, r, world, npix))
        smk.binding.Exchange.connect_local(xs)
        frames = torch.zeros((len(scs), npix, 4), dtype=torch.float32, device="cuda")
        for i, sc in enumerate(scs):
            slot = i & 1
            for R, x in zip(rs, xs):
                push_scene(R, sc, upload=False)
                x.acquire(slot)
                R.render_device(x.partial(slot), None, None)
                x.rendered(slot)
            smk.binding.Exchange.frame_local(xs, slot, frames[i].data_ptr())
        xs[0].wait(None)
        torch.cuda.synchronize()
        for R in rs:
            assert R.stat("slab_failures") == 0
        for i, sc in enumerate(scs):
            got = frames[i].cpu().numpy().reshape(sc.height, sc.width, 4)
            ref = sc.render()
            assert ref[..., 3].max() > 0.05 and np.abs(got - ref).max() <= 1e-4, i
    finally:
        for x in xs:
            x.close()
        for R in rs:
            R.close()


def test_exchange_rccl_transport_loads_and_runs_a_single_rank(gpu_renderer_factory, smk):
    """The RCCL transport on what one GPU allows: librccl opens at run time, the unique id is made,
    ncclCommInitRank succeeds for a world of one, and a frame passes through smk_exchange_frame (no
    peer to send to: the own tile is merged and delivered).  More ranks need more GPUs: bench.py --gpus N."""
    import torch
    sc = make_scene("cfg3", n=32, size=40, steps=48, pose="rot", f32=True, shade=1)
    npix = sc.width * sc.height
    R = gpu_renderer_factory()
    x = None
    try:
        R.set_shard(0, 1)
        push_scene(R, sc)
        uid = smk.binding.exchange_unique_id()
        assert len(uid) == 128 and any(uid)
        x = smk.binding.Exchange(R, 0, 1, npix, id=uid)
        out = torch.zeros((npix, 4), dtype=torch.float32, device="cuda")
        R.render_device(x.partial(0), None, None)
        x.rendered(0)
        x.frame(0, out.data_ptr())
        x.wait(None)
        torch.cuda.synchronize()
        assert np.abs(out.cpu().numpy().reshape(sc.height, sc.width, 4) - sc.render()).max() <= 1e-4
    finally:
        if x is not None:
            x.close()
        R.close()


def test_merge_uses_the_order_of_the_pose_the_frame_was_rendered_with(gpu_renderer_factory, smk):
    """Frame i's merge is enqueued after frame i + 1's camera has been set (two frames in flight).  When the eye crosses
    the shards' split plane between the two poses the visibility orders differ, and "over" is not commutative: the merge
    must use the order taken when frame i was rendered (smk_exchange_rendered), not the context's current camera."""
    import torch
    world = 2
    a = make_scene("cfg3", n=32, size=48, steps=48, pose="rot", f32=True, shade=1)
    b = make_scene("cfg3", n=32, size=48, steps=48, pose="rot", f32=True, shade=1)
    a.xform = __import__("oracle").rotation((0, 1, 0), 35)      # eye on one side of the x mid-plane ...
    b.xform = __import__("oracle").rotation((0, 1, 0), -35)     # ... and on the other
    npix = a.width * a.height
    rs, xs = [], []
    try:
        for r in range(world):
            R = gpu_renderer_factory()
            rs.append(R)
            R.set_shard(r, world)
            push_scene(R, a)
            xs.append(smk.binding.Exchange(R, r, world, npix))
        smk.binding.Exchange.connect_local(xs)
        orders = []
        for sc in (a, b):
            push_scene(rs[0], sc, upload=False)
            orders.append(rs[0].shard_order(world))
        assert orders[0] != orders[1], "the two poses must see the shards in different orders"
        frame = torch.zeros((npix, 4), dtype=torch.float32, device="cuda")
        for R, x in zip(rs, xs):
            push_scene(R, a, upload=False)
            x.acquire(0)
            R.render_device(x.partial(0), None, None)
            x.rendered(0)
            push_scene(R, b, upload=False)                        # the next frame's pose, before this frame's merge
        smk.binding.Exchange.frame_local(xs, 0, frame.data_ptr())
        xs[0].wait(None)
        torch.cuda.synchronize()
        got = frame.cpu().numpy().reshape(a.height, a.width, 4)
        assert np.abs(got - a.render()).max() <= 1e-4
    finally:
        for x in xs:
            x.close()
        for R in rs:
            R.close()
