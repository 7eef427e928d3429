#!/usr/bin/env python3
"""Wall time per frame when the camera moves every frame (developer tool): the host plans each frame afresh, so
this shows what planning costs beside the kernel.   python tools/moving_camera.py [volume] [frames] [nranks]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    nranks = int(sys.argv[3]) if len(sys.argv) > 3 else 1     # > 1: the shard of rank nranks // 2 (a shorter kernel beside the same planning)
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    if nranks > 1:
        r.set_shard(nranks // 2, nranks)
    vghf, nrm = bench.make_volume(r, n)
    r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    del vghf, nrm
    bench.configure(r, "cfg3", n, 1024, 512)
    r.set_option("kernel", int(os.environ.get("SMK_KERNEL", "0")))
    if os.environ.get("SMK_TILE"):    # (developer: force a slice-ring workgroup shape, option "tile")
        r.set_option("tile", int(os.environ["SMK_TILE"]))
    frame = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run(moving):
        for warm in (True, False):
            torch.cuda.synchronize()
            r.timing_reset()
            t0 = time.perf_counter()
            host = 0.0
            for f in range(frames):
                if moving:
                    xform = bench.rotation((1, 1, 0), 30 + 0.05 * f)
                    r.set_camera(bench.modelview(xform, (1.0, 1.0, 1.0)), bench.FRUSTUM, (1.0, 20.0), 1024, 1024)
                h0 = time.perf_counter()
                r.render_device(frame.data_ptr(), None, st)
                host += time.perf_counter() - h0
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / frames * 1e3
        kms, _ = r.timing_read()
        print("%-7s camera: %.3f ms/frame wall, kernel %.3f ms (kernel id %d), host time inside smk_render_device %.3f ms/frame; slice-ring failures %d retries %d"
              % ("moving" if moving else "static", t, kms, r.last_frame_info()[0], host / frames * 1e3, r.stat("slab_failures"), r.stat("slab_retries")), flush=True)

    run(False)
    run(True)


if __name__ == "__main__":
    main()
