#!/usr/bin/env python3
"""Developer probe (GPU box): the column-stream kernel against the gather kernel on small scenes, with its statistics."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
from conftest import load_package  # noqa: E402
from _scenes import make_scene, push_scene  # noqa: E402
import oracle  # noqa: E402

pkg = load_package()
R = pkg.Renderer(0)
cases = [("cfg3", "rot", True, 1, 64, 9), ("cfg3", "x-", False, 1, 90, 13), ("cfg3", "rot", True, 1, 64, 64), ("cfg4", "z-", False, 1, 72, 80), ("cfg4", "x+", True, 1, 72, 80), ("cfg2", "y-", True, 0, 96, 100),
         ("tf3d_panes", "diag", True, 1, 64, 64), ("cfg1", "diag", False, 0, 64, 64)]
for kind, pose, f32, shade, size, steps in cases:
    sc = make_scene(kind, n=32, size=size, steps=steps, pose=pose, f32=f32, shade=shade)
    ref = sc.render()
    want = oracle.inside_samples()
    push_scene(R, sc)
    R.set_option("kernel", 1)
    a = R.render()
    R.set_option("kernel", 3)
    R.set_option("cols_counts", 1)
    for chunk in (0, 8):
        R.set_option("cols_chunk", chunk)
        try:
            b = R.render()
        except Exception as ex:
            print(kind, pose, "chunk", chunk, "FAILED:", str(ex).splitlines()[-1][:200], flush=True)
            continue
        cfg = int(R.stat("cols_config"))
        print("%-10s %-4s f32=%d chunk=%3d kernel=%d  |C-G| %.2e  |C-ref| %.2e  samples %d / %d  visible %d  slices %d  segments %d  jobs %d  CWxCH %dx%d slots %d"
              % (kind, pose, f32, chunk, R.last_frame_info()[0], np.abs(a - b).max(), np.abs(b - ref).max(), R.stat("cols_samples"), want,
                 R.stat("cols_visible"), R.stat("cols_slices"), R.stat("cols_segments"), R.stat("cols_jobs"), cfg & 255, (cfg >> 8) & 255, (cfg >> 16) & 255), flush=True)
    R.set_option("cols_chunk", 0)
R.close()
