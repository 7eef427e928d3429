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


def gpu_only(R, sc, tag):
    """the two kernels against each other only (large cases: no CPU frame)"""
    push_scene(R, sc)
    R.set_option("kernel", 1)
    R.set_option("bricks", 0)      # the reference frame: gather kernel, every sample fetched and classified
    a = R.render()
    R.set_option("bricks", 1)
    R.set_option("kernel", 2)
    try:
        b = R.render()
    except Exception as e:
        if "not applicable" not in str(e):
            raise
        return str(e).split("not applicable:")[-1].strip()[:80]
    finally:
        R.set_option("kernel", 0)
    assert np.array_equal(a, b), "slice-ring vs gather %g" % np.abs(a - b).max()
    return None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    pkg = bench.load_package()
    R = pkg.Renderer(0)
    rng = np.random.default_rng(seed)
    bad = 0
    declined = {}
    check_cpu = os.environ.get("SMK_FUZZ_CPU", "1") == "1"
    from simian_spacemonkey_amd import sortlast
    for case in range(n):
        sc, kind, f32, dims = F.random_scene(rng)
        tag = "case %d: %s dims %s f32 %d %dx%d x%d rate %.2f shade %d eye %s trans %s frustum %s shard %s" % (
            case, kind, dims, f32, sc.width, sc.height, sc.steps, sc.sample_rate, sc.shade_mode, sc.eye, sc.trans,
            tuple(round(float(x), 3) for x in sc.frustum), (sc.shard, sc.clip))
        r = R
        if sc.shard:
            r = pkg.Renderer(0)
            r.set_shard(*sc.shard)
            sc.region = sortlast.shard_region(sc.dims, *sc.shard)
        try:
            why = F.one_case(r, sc, tag) if check_cpu else gpu_only(r, sc, tag)
            if why:
                declined[why] = declined.get(why, 0) + 1
        except Exception as e:
            bad += 1
            print(tag + " -> " + str(e)[:200], flush=True)
        finally:
            if r is not R:
                r.close()
    print("declined:", declined)
    print("%d bad of %d" % (bad, n))
    R.close()


if __name__ == "__main__":
    main()
