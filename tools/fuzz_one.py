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
        r = R
        if sc.shard:
            from simian_spacemonkey_amd import sortlast
            r = pkg.Renderer(0)
            r.set_shard(*sc.shard)
            sc.region = sortlast.shard_region(sc.dims, *sc.shard)
        print("   rate %.3f eye %s trans %s frustum %s shard %s clip %s region %s" % (sc.sample_rate, sc.eye, sc.trans, sc.frustum, sc.shard, sc.clip, sc.region))
        try:
            print("   ->", F.one_case(r, sc, ""), flush=True)
        except Exception as e:
            print("   FAILED " + str(e)[-300:], flush=True)
        ref = sc.render()
        for k in (1, 2):
            r.set_option("kernel", k)
            try:
                img = r.render()
                d = np.abs(img - ref)
                j, i = np.unravel_index(d.max(axis=2).argmax(), d.shape[:2])
                print("   kernel %d vs CPU: max %g at pixel (%d,%d): gpu %s cpu %s" % (k, d.max(), i, j, img[j, i], ref[j, i]), flush=True)
            except Exception as e:
                print("   kernel %d: %s" % (k, str(e)[-200:]))
        r.set_option("kernel", 0)
        if r is not R:
            r.close()
    R.close()


if __name__ == "__main__":
    main()
