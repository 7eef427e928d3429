#!/usr/bin/env python3
"""Frame time of the cfg 3 frame with a first-hit depth output and with the back-to-front blend, gather kernel vs
slice-ring kernel (developer tool, GPU box only).   python tools/modes_time.py [volume]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    vghf, nrm = bench.make_volume(r, n)
    r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    del vghf, nrm
    bench.configure(r, "cfg3", n, 1024, 512)
    frame = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
    depth = torch.zeros((1024 * 1024,), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    keep = {}
    for name, blend, want_depth in (("plain", 0, False), ("first-hit depth", 0, True), ("back to front", 1, False), ("back to front + depth", 1, True), ("maximum", 2, False)):
        r.set_blend(blend)
        for kernel in (1, 2):
            r.set_option("kernel", kernel)
            for _ in range(60 if kernel == 2 else 3):
                r.render_device(frame.data_ptr(), depth.data_ptr() if want_depth else None, st)
            torch.cuda.synchronize()
            r.timing_reset()
            for _ in range(10):
                r.render_device(frame.data_ptr(), depth.data_ptr() if want_depth else None, st)
            torch.cuda.synchronize()
            kms, _ = r.timing_read()
            f = frame.clone()
            d = depth.clone()
            keep[(name, kernel)] = (f, d)
            print("%-24s kernel %d (ran %d): %.3f ms" % (name, kernel, r.last_frame_info()[0], kms), flush=True)
        a, b = keep[(name, 1)], keep[(name, 2)]
        msg = "   frames: max |gather - slice ring| %.3g" % float((a[0] - b[0]).abs().max())
        if want_depth:
            fa, fb = torch.isfinite(a[1]), torch.isfinite(b[1])
            msg += "; depth: same pixels hit %s, max difference %.3g" % (bool((fa == fb).all()), float((a[1][fa & fb] - b[1][fa & fb]).abs().max()))
        print(msg, flush=True)
    r.close()


if __name__ == "__main__":
    main()
