#!/usr/bin/env python3
"""Developer tool (GPU box): run tests/test_gpu_fuzz.py's random frames without stopping at the
first failure and list every failing case with its parameters."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import test_gpu_fuzz as F  # noqa: E402
from _scenes import push_scene  # noqa: E402
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    pkg = bench.load_package()
    R = pkg.Renderer(0)
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n):
        sc, kind, f32, dims = F.random_scene(rng)
        push_scene(R, sc)
        R.set_option("kernel", 1)
        a = R.render()
        R.set_option("kernel", 2)
        try:
            b = R.render()
        except Exception as e:
            if "not applicable" in str(e):
                continue
            bad += 1
            print("case %d: %s dims %s f32 %d %dx%d x%d shade %d eye %s trans %s frustum %s -> %s" % (
                case, kind, dims, f32, sc.width, sc.height, sc.steps, sc.shade_mode, sc.eye, sc.trans,
                tuple(round(float(x), 3) for x in sc.frustum), str(e)[:60]), flush=True)
            continue
        if not np.array_equal(a, b):
            bad += 1
            print("case %d: frames differ %g" % (case, np.abs(a - b).max()), flush=True)
    print("%d bad of %d" % (bad, n))
    R.close()


if __name__ == "__main__":
    main()
