"""Sort-last plumbing (simian-spacemonkey_amd/sortlast.py) on CPU: world_size-2 gloo ranks each
render their brick shard with the CPU checker, exchange 1/P tiles (all_to_all), composite in
BSP order and gather; the result must equal the unsharded frame.  The GPU path swaps the
shard renderer and the compositor for the HIP ones (tests/test_gpu_sortlast.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_over(layers, order):
    """checker compositor: C += (1-A)*src, front layer first"""
    acc = torch.zeros_like(layers[0])
    for l in order:
        acc = acc + (1.0 - acc[:, 3:4]) * layers[l]
    return acc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, pose, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from conftest import load_package
    load_package()
    from simian_spacemonkey_amd import sortlast
    from _scenes import make_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = make_scene("cfg3", n=24, size=40, steps=40, pose=pose, shade=1)
        sc.region = sortlast.shard_region(sc.dims, rank, world)
        part = torch.from_numpy(sc.render()).reshape(-1, 4)
        npix = part.shape[0]
        tp = sortlast.tile_pixels(npix, world)
        pad = torch.zeros((tp * world, 4))
        pad[:npix] = part
        # eye in voxel index space from the checker's ray coefficients: plane parameter -> inf
        import oracle as O
        mv = np.array(sc.mv()).reshape(4, 4).T
        eye_model = np.linalg.inv(mv)[:3, 3]
        eye_vox = [eye_model[a] * sc.dims[a] / float(sc.fsize[a]) - 0.5 for a in range(3)]
        order = sortlast.front_to_back_order(eye_vox, sc.dims, world)
        tile = sortlast.exchange_and_composite(pad, order, cpu_over)
        full = sortlast.gather_frame(tile, 0)
        if rank == 0:
            sc.region = ((0, 0, 0), sc.dims)
            whole = sc.render().reshape(-1, 4)
            q.put(float(np.abs(full[:npix].numpy() - whole).max()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pose", ["rot", "back"])
def test_two_rank_sort_last_equals_unsharded(pose):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pose, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) < 2e-6


def test_shard_regions_partition_the_volume(smk):
    from simian_spacemonkey_amd import sortlast
    dims = (20, 12, 16)
    for world in (1, 2, 4, 8):
        seen = np.zeros(dims[::-1], np.int32)
        for r in range(world):
            g0, g1 = sortlast.shard_region(dims, r, world)
            seen[g0[2]:g1[2], g0[1]:g1[1], g0[0]:g1[0]] += 1
        assert np.all(seen == 1)


def test_bsp_order_is_a_valid_visibility_order(smk):
    """front-to-back: a shard never precedes one that is nearer along every split axis"""
    from simian_spacemonkey_amd import sortlast
    dims = (16, 16, 16)
    for eye in [(-40, 5, 100), (30, 30, 30), (7.4, -9, 8.1), (100, 100, -100)]:
        for world in (2, 4, 8):
            order = sortlast.front_to_back_order(eye, dims, world)
            assert sorted(order) == list(range(world))
            centres = []
            for r in order:
                g0, g1 = sortlast.shard_region(dims, r, world)
                centres.append([(g0[a] + g1[a]) / 2 - 0.5 for a in range(3)])
            for i in range(world):
                for j in range(i + 1, world):
                    # j after i: j must not be strictly nearer than i on every differing axis
                    d = [(abs(centres[j][a] - eye[a]) < abs(centres[i][a] - eye[a])) for a in range(3)
                         if centres[i][a] != centres[j][a]]
                    assert not all(d)
