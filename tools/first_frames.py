#!/usr/bin/env python3
"""Developer probe: HIP-event kernel time of each of the first frames of a context (what auto mode's trial frames see).
    python tools/first_frames.py [nranks] [kernel] [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

nranks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 12
pkg = bench.load_package()
r = pkg.Renderer(0)
if nranks > 1:
    r.set_shard(nranks // 2, nranks)
n = 512
vghf, nrm = bench.make_volume(r, n)
r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
del vghf, nrm
bench.configure(r, "cfg3", n, 1024, 512)
r.set_option("kernel", kernel)
frame = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for f in range(frames):
    torch.cuda.synchronize()
    r.timing_reset()
    r.render_device(frame.data_ptr(), None, st)
    torch.cuda.synchronize()
    kms, _ = r.timing_read()
    print("frame %2d: kernel %d, %.3f ms; tiles cut %d, workgroups %d" % (f, r.last_frame_info()[0], kms, r.stat("slab_split_tiles"), r.stat("slab_workgroups")), flush=True)
r.close()
