#!/usr/bin/env python3
"""Frame time of the cfg 3 frame with shadows: the two marches (default) against a launch per slice (developer tool, GPU
box only).   python tools/shadow_time.py [volume] [light buffer px]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    lb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    vghf, nrm = bench.make_volume(r, n)
    r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    del vghf, nrm
    xform, _ = bench.configure(r, "cfg3", n, 1024, 512)
    r.set_shading("r8k", (3.0, 4.0, -3.0), bench.EYE, bench.AT, [float(v) for v in xform.T.reshape(-1)], bench.INTENS)
    r.set_shadow(1, lb, 0.5)
    frame = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    keep = {}
    for name, march, kernel in (("two marches, eye pass on the slice-ring kernel", 1, 2), ("two marches, eye pass on the gather kernel", 1, 1),
                                ("a launch per slice", 0, 0), ("auto", 1, 0)):
        r.set_option("shadow_march", march)
        r.set_option("kernel", kernel)
        for _ in range(40 if kernel != 1 else 3):
            r.render_device(frame.data_ptr(), None, st)
        torch.cuda.synchronize()
        r.timing_reset()
        for _ in range(10):
            r.render_device(frame.data_ptr(), None, st)
        torch.cuda.synchronize()
        kms, _ = r.timing_read()
        keep[name] = (frame.clone(), torch.from_numpy(r.light_buffer()))
        print("%-48s %.3f ms per frame (kernel id %d)" % (name, kms, r.last_frame_info()[0]), flush=True)
    names = list(keep)
    for n in names[1:]:
        print("%s vs %s: frames max |diff| %.3g, light buffers identical: %s" % (
            names[0], n, float((keep[names[0]][0] - keep[n][0]).abs().max()), bool((keep[names[0]][1] == keep[n][1]).all())), flush=True)
    print("max alpha %.3f" % float(keep[names[0]][0][:, 3].max()), flush=True)
    r.close()


if __name__ == "__main__":
    main()
