#!/usr/bin/env python3
"""Soak run (developer tool, GPU box): a camera that moves every frame, thousands of frames in auto mode, and every
so often the same frame from both ray-marchers compared bit for bit; no slice-ring frame may be flagged.
    python tools/soak.py [volume] [frames] [check_every]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    vghf, nrm = bench.make_volume(r, n)
    r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    del vghf, nrm
    bench.configure(r, "cfg4" if n >= 1024 else "cfg3", n, 1024, 512)
    a = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    by_kernel = {1: 0, 2: 0, 3: 0}
    axes = [(1, 1, 0), (0, 1, 0), (1, 0, 0), (1, 1, 1), (0.2, 1, 0.1)]
    for f in range(frames):
        ax = axes[(f // 400) % len(axes)]
        xform = bench.rotation(ax, 30 + 0.21 * f)
        r.set_camera(bench.modelview(xform, (1.0, 1.0, 1.0)), bench.FRUSTUM, (1.0, 20.0), 1024, 1024)
        r.set_shading("r8k", bench.LIGHT, bench.EYE, bench.AT, [float(v) for v in xform.T.reshape(-1)], bench.INTENS)
        r.render_device(a.data_ptr(), None, st)
        if f % every == every - 1:
            torch.cuda.synchronize()
            k = r.last_frame_info()[0]
            by_kernel[k] = by_kernel.get(k, 0) + 1
            r.set_option("kernel", 1)
            r.set_option("bricks", 0)      # the reference frame: gather kernel, every sample fetched and classified
            r.render_device(b.data_ptr(), None, st)
            torch.cuda.synchronize()
            r.set_option("kernel", 0)
            r.set_option("bricks", 1)
            if not torch.equal(a, b):
                bad += 1
                print("frame %d (kernel %d): differs from the gather kernel by %g" % (f, k, float((a - b).abs().max())), flush=True)
    torch.cuda.synchronize()
    print("%d frames, %d compared (%d of them slice-ring frames), %d differing; slice-ring failures %d, retries %d"
          % (frames, frames // every, by_kernel[2], bad, r.stat("slab_failures"), r.stat("slab_retries")), flush=True)
    sys.exit(1 if bad or r.stat("slab_failures") else 0)


if __name__ == "__main__":
    main()
