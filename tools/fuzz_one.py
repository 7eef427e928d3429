#!/usr/bin/env python3
"""Developer tool (GPU box): re-run chosen cases of the random-frame fuzz with the slice-ring plan printed.
    SMK_DEBUG=1 python tools/fuzz_one.py SEED CASE [CASE ...]"""
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
    seed = int(sys.argv[1])
    want = set(int(a) for a in sys.argv[2:])
    pkg = bench.load_package()
    R = pkg.Renderer(0)
    rng = np.random.default_rng(seed)
    for case in range(max(want) + 1):
        sc, kind, f32, dims = F.random_scene(rng)
        if case not in want:
            continue
        print("== case %d: %s dims %s f32 %d %dx%d x%d shade %d" % (case, kind, dims, f32, sc.width, sc.height, sc.steps, sc.shade_mode), flush=True)
        push_scene(R, sc)
        R.set_option("kernel", 1)
        a = R.render()
        R.set_option("kernel", 2)
        for extra in ([], [("lockstep", 16)]):
            for k, v in extra:
                R.set_option(k, v)
            try:
                b = R.render()
                print("   ok, equal to gather: %s" % np.array_equal(a, b), flush=True)
            except Exception as e:
                print("   " + str(e)[:100], flush=True)
            R.set_option("lockstep", 0)
    R.close()


if __name__ == "__main__":
    main()
