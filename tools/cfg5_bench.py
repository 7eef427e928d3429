#!/usr/bin/env python3
"""Developer tool (GPU box): BASELINE config 5 at full size -- two 512^3 fields merged on the GPU,
dense 3-D transfer function, noise-perturbed fetch, 1024^2 x 1024 planes -- timed on the gather
kernel under a few option settings.
    python tools/cfg5_bench.py [wave_w=8 blk_w=2 lockstep=1 ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import _scenes as S  # noqa: E402
import oracle as O  # noqa: E402


def main():
    n, size, planes = 512, 1024, 1024
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    fields = torch.empty((n, n, n, 2), dtype=torch.uint8, device="cuda")
    one = torch.empty((n, n, n), dtype=torch.uint8, device="cuda")
    for e, seed in enumerate((1, 2)):
        r.synth_volume_device(0, seed, (n, n, n), one.data_ptr())
        fields[..., e] = one if e == 0 else one.flip(2)
    merged = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    nrm = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    t0 = time.perf_counter()
    r.merge_fields_device(fields.data_ptr(), 2, (n, n, n), merged.data_ptr(), nrm.data_ptr())
    print("merge of two %d^3 fields (+ normals): %.2f ms" % (n, (time.perf_counter() - t0) * 1e3))
    del fields, one
    xform = bench.rotation((1, 1, 0), 30)
    mv = bench.modelview(xform, (1.0, 1.0, 1.0))
    r.upload_volume_device(merged.data_ptr(), (n, n, n), 3, 0, nrm.data_ptr(), dmode="V2G")
    r.set_option("tf_raw", 1)
    r.set_tf3d(S.tf3d_dense())
    r.set_camera(mv, bench.FRUSTUM, (1.0, 20.0), size, size)
    r.set_sampling(0.0, planes, 1.0, 1)
    r.set_shading("r8k", bench.LIGHT, bench.EYE, bench.AT, [float(v) for v in xform.T.reshape(-1)], bench.INTENS)
    full = "full" in sys.argv[1:]   # SURVEY 8(d)'s weights (.2, .1); default: a tenth of them
    if full:
        sys.argv.remove("full")
    r.set_perturb(O.noise_tex(32), (.2, .1, 0, 0) if full else (.02, .01, 0, 0), (.2, 2.1, 4.5, 8.7))
    out = torch.zeros((size * size, 4), dtype=torch.float32, device="cuda")
    base = None
    for var in [""] + sys.argv[1:]:
        for kv in var.split(","):
            if kv == "nopert":                # (how much of the frame is the perturbation?)
                r.set_perturb(None, None, None)
            elif kv:
                k, v = kv.split("=")
                r.set_option(k, int(v))
        r.render_device(out.data_ptr(), None, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            r.render_device(out.data_ptr(), None, None)
        torch.cuda.synchronize()
        img = out.cpu().numpy()
        if base is None:
            base = img
        print("%-28s %.2f ms/frame   maxdiff vs first %.1e" % (var or "(defaults)", (time.perf_counter() - t0) / 5 * 1e3, float(np.abs(img - base).max())), flush=True)
    r.close()


if __name__ == "__main__":
    main()
