#!/usr/bin/env python3
"""Kernel experiment harness (developer tool, GPU box only): builds the north-star volume once,
then times render-kernel variants selected through smk_set_option and checks each frame against
the generic gather kernel's.

    python tools/kbench.py --volume 1024 --variants kernel=1 kernel=2 kernel=2,slab_T=8
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, default=1024)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--planes", type=int, default=512)
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--pose", default="rot")
    ap.add_argument("--u8", action="store_true")
    ap.add_argument("--variants", nargs="*", default=["kernel=1"])
    a = ap.parse_args()
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    n = a.volume
    vghf, nrm = bench.make_volume(r, n)
    if a.u8:
        v8 = (vghf * 255.0).to(torch.uint8)
        r.upload_volume_device(v8.data_ptr(), (n, n, n), 3, 0, nrm.data_ptr())
    else:
        r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    del vghf, nrm
    torch.cuda.empty_cache()
    bench.configure(r, a.workload, n, a.size, a.planes)
    if a.pose != "rot":
        ax, deg = {"id": ((0, 1, 0), 0), "x": ((0, 1, 0), 90), "y": ((1, 0, 0), 90), "back": ((0, 1, 0), 160),
                   "side": ((.2, 1, .1), 75), "r45": ((1, 1, 1), 45)}[a.pose]
        xform = bench.rotation(ax, deg)
        r.set_camera(bench.modelview(xform, (1.0, 1.0, 1.0)), bench.FRUSTUM, (1.0, 20.0), a.size, a.size)
        r.set_shading("r8k", bench.LIGHT, bench.EYE, bench.AT, [float(v) for v in xform.T.reshape(-1)], bench.INTENS)
    frame = torch.zeros((a.size * a.size, 4), dtype=torch.float32, device="cuda")
    base = None
    work = torch.cuda.Stream()
    for var in a.variants:
        for kv in var.split(","):
            k, v = kv.split("=")
            r.set_option(k, int(v))
        with torch.cuda.stream(work):
            t, kms, kn = bench.timed(r, a.frames, 2, frame, 1, None)
        kern, _, alg = r.last_frame_info()
        img = frame.cpu().numpy()
        if base is None:
            base = img
        err = float(np.abs(img - base).max())
        print("%-32s kernel=%d  %.3f ms/frame  (event avg %.3f ms)  %.1f GB/s alg  frac %.3f  maxdiff_vs_first %.2e  alpha_mean %.4f"
              % (var, kern, t / a.frames * 1e3, kms, alg / (kms * 1e-3) / 1e9, alg / (kms * 1e-3) / 1e9 / 8000, err, img[:, 3].mean()),
              flush=True)
        if "lockstep=6" in var or "lockstep=4" in var:
            t = img[::-1][:1024]
            t = t[t[:, 3] > 0]
            print("   loader cycles (mean over %d WGs): issue %.0f  wait %.0f  idle %.0f  total %.0f" % ((len(t),) + tuple(t.mean(0))))
    r.close()


if __name__ == "__main__":
    main()
