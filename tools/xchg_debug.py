#!/usr/bin/env python3
"""developer probe: in-process smk_exchange with N shard contexts, per-tile error of each frame"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
from _scenes import make_scene, push_scene  # noqa: E402

pkg = bench.load_package()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
poses = ("rot", "back", "side", "back", "rot")
scs = [make_scene("cfg3", n=32, size=45, steps=48, pose=p, f32=True, shade=1) for p in poses]
npix = 45 * 45
rs, xs = [], []
for r in range(world):
    R = pkg.Renderer(0)
    rs.append(R)
    R.set_shard(r, world)
    push_scene(R, scs[0])
    R.set_option("kernel", kernel)
    if "--nolock" in sys.argv:
        R.set_option("lockstep", 0)
    xs.append(pkg.Exchange(R, r, world, npix))
pkg.Exchange.connect_local(xs)
frames = torch.zeros((len(scs), npix, 4), dtype=torch.float32, device="cuda")
direct = torch.zeros((len(scs), world, npix, 4), dtype=torch.float32, device="cuda")
for i, sc in enumerate(scs):
    slot = i & 1
    for r, (R, x) in enumerate(zip(rs, xs)):
        push_scene(R, sc, upload=False)
        R.set_option("kernel", kernel)
        x.acquire(slot)
        R.render_device(x.partial(slot), None, None)
        x.rendered(slot)
        R.render_device(direct[i, r].data_ptr(), None, None)      # the same layer again, kept for the check
    pkg.Exchange.frame_local(xs, slot, frames[i].data_ptr())
    if "--sync" in sys.argv:
        torch.cuda.synchronize()
xs[0].wait(None)
torch.cuda.synchronize()
tp = (npix + world - 1) // world
for i, sc in enumerate(scs):
    ref = sc.render().reshape(-1, 4)
    got = frames[i].cpu().numpy()
    order = rs[0].shard_order(world) if False else None
    err = np.abs(got - ref).max(axis=1)
    per_tile = [float(err[t * tp:(t + 1) * tp].max()) if t * tp < npix else 0.0 for t in range(world)]
    # composite of the directly rendered layers with rank 0's order for THIS pose
    push_scene(rs[0], sc, upload=False)
    o = rs[0].shard_order(world)
    out = torch.zeros((npix, 4), dtype=torch.float32, device="cuda")
    rs[0].composite_over_device(direct[i].data_ptr(), world, o, npix, out.data_ptr(), None)
    torch.cuda.synchronize()
    e2 = float(np.abs(out.cpu().numpy() - ref).max())
    print("frame %d pose %-5s order %s  exchange max err per tile %s   direct composite err %.2e" %
          (i, poses[i], o, ["%.1e" % e for e in per_tile], e2), flush=True)
    if "--layers" in sys.argv and i < 2:
        from simian_spacemonkey_amd import sortlast
        for r in range(world):
            sc.region = sortlast.shard_region(sc.dims, r, world)
            lay = direct[i, r].cpu().numpy()
            refl = sc.render().reshape(-1, 4)
            if "--dump" in sys.argv and i == 1 and r in (0, 2):
                np.save(os.path.join(ROOT, "gpurun_out", "dbg_layer_r%d.npy" % r), lay.reshape(45, 45, 4))
                np.save(os.path.join(ROOT, "gpurun_out", "dbg_ref_r%d.npy" % r), refl.reshape(45, 45, 4))
            print("     rank %d region %s layer err %.2e (alpha max %.3f ref %.3f)" % (r, sc.region, float(np.abs(lay - refl).max()), lay[:, 3].max(), refl[:, 3].max()))
        sc.region = ((0, 0, 0), sc.dims)
for x in xs:
    x.close()
for R in rs:
    R.close()
