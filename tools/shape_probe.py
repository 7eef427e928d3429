#!/usr/bin/env python3
"""Developer tool (GPU box): the cfg 3 frame at several rotation angles, with each of the small workgroup shapes forced
(option tile 21 = 16x40, 20 = 40x16, 5 = 32x16) -- what the shape probe's ranking should reproduce.  With SMK_DEBUG=1 the
launcher prints each plan (window, chunks per slice, ring slots).   python tools/shape_probe.py [angles...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    angles = [float(a) for a in sys.argv[1:]] or [30, 33, 36, 40, 45]
    n = 512
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    vghf, nrm = bench.make_volume(r, n)
    r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    del vghf, nrm
    bench.configure(r, "cfg3", n, 1024, 512)
    r.set_option("kernel", 2)
    frame = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for ang in angles:
        xform = bench.rotation((1, 1, 0), ang)
        r.set_camera(bench.modelview(xform, (1.0, 1.0, 1.0)), bench.FRUSTUM, (1.0, 20.0), 1024, 1024)
        res = []
        for tile in (21, 20, 5, 0):
            r.set_option("tile", tile)
            sys.stderr.write("== angle %g tile %d\n" % (ang, tile))
            sys.stderr.flush()
            for _ in range(40):
                r.render_device(frame.data_ptr(), None, st)
            torch.cuda.synchronize()
            r.timing_reset()
            for _ in range(10):
                r.render_device(frame.data_ptr(), None, st)
            torch.cuda.synchronize()
            res.append("%s %.3f" % ({21: "16x40", 20: "40x16", 5: "32x16", 0: "probe"}[tile], r.timing_read()[0]))
        print("angle %5.1f: %s ms" % (ang, "  ".join(res)), flush=True)
    r.close()


if __name__ == "__main__":
    main()
