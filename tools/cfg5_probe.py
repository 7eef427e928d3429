#!/usr/bin/env python3
"""BASELINE config 5 on one GPU as bench.py's `cfg5` leg sets it up (developer tool, GPU box only; meant to run under
rocprofv3 --pmc): two 512^3 fields merged, dense 3-D table, R8k Phong, noise-perturbed fetch at SURVEY 8(d)'s weights.
    python tools/cfg5_probe.py [frames] [weight scale] [option=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    wscale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    n = 512
    fields = torch.empty((n, n, n, 2), dtype=torch.uint8, device="cuda")
    one = torch.empty((n, n, n), dtype=torch.uint8, device="cuda")
    for e, seed in enumerate((1, 2)):
        r.synth_volume_device(1, seed, (n, n, n), one.data_ptr())
        fields[..., e] = one
    del one
    merged = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    mnrm = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    r.merge_fields_device(fields.data_ptr(), 2, (n, n, n), merged.data_ptr(), mnrm.data_ptr())
    del fields
    r.upload_volume_device(merged.data_ptr(), (n, n, n), 3, 0, mnrm.data_ptr(), dmode="V2G")
    del merged, mnrm
    pane = np.load(os.path.join(bench.ROOT, "tests", "golden", "tf_cfg3_levwidget.npy"))
    t3 = np.stack([pane] * 4).copy()
    for h_, be in enumerate((0.4, 1.0, 0.7, 0.4)):
        t3[h_, ..., 3] = (pane[..., 3].astype(np.float32) * be).astype(np.uint8)
    r.set_option("tf_raw", 1)
    r.set_tf3d(t3)
    xform = bench.rotation((1, 1, 0), 30)
    xf = [float(v) for v in xform.T.reshape(-1)]
    r.set_camera(bench.modelview(xform, (1.0, 1.0, 1.0)), bench.FRUSTUM, (1.0, 20.0), 1024, 1024)
    r.set_sampling(0.0, 1024, 1.0, 1)
    r.set_shading("r8k", bench.LIGHT, bench.EYE, bench.AT, xf, bench.INTENS)
    if wscale > 0:
        r.set_perturb(bench.libc_noise_tex(32), (.2 * wscale, .1 * wscale, 0, 0), (.2, 2.1, 4.5, 8.7))
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        r.set_option(k, int(v))
    frame = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        r.render_device(frame.data_ptr(), None, st)
    torch.cuda.synchronize()
    r.timing_reset()
    for _ in range(frames):
        r.render_device(frame.data_ptr(), None, st)
    torch.cuda.synchronize()
    kms, _ = r.timing_read()
    inv = r.count_samples()
    print("cfg5 weights x %.2f: kernel %d, %.3f ms per frame; samples in volume %.4g of %.4g nominal; alpha mean %.4f"
          % (wscale, r.last_frame_info()[0], kms, inv, 1024.0 ** 3, float(frame[:, 3].mean())), flush=True)
    r.close()


if __name__ == "__main__":
    main()
