#!/usr/bin/env python3
"""Developer tool (GPU box): random frames WITH SHADOWS -- volume sizes, poses, cameras and light positions at random.
For each: the eye pass on the gather kernel against the CPU checker (1e-4), on the slice-ring kernel against the gather
kernel (bit-identical; or the reason it declines), and the two marches against a launch per slice (light buffers
bit-identical; frames bit-identical where the slices run away from the viewer, <= 2e-5 otherwise).
    python tools/fuzz_shadow.py [seed] [cases]"""
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
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    pkg = bench.load_package()
    R = pkg.Renderer(0)
    rng = np.random.default_rng(seed)
    took = declined = skipped = 0
    worst_cpu = worst_btf = 0.0
    reasons = {}
    for case in range(cases):
        sc, kind, f32, dims = F.random_scene(rng)
        sc.shard = None
        sc.clip = None
        if sc.shade_mode == 2:
            sc.shade_mode = 1          # (no shadow mode with the NV20 combiners)
        ang = rng.normal(size=3)
        ang /= np.linalg.norm(ang) + 1e-9
        sc.light_pos = tuple(float(v) for v in ang * float(rng.uniform(2.0, 8.0)))
        sc.shadow = (int(rng.integers(16, 200)), float(rng.uniform(0.3, 1.0)))
        tag = "case %d (%s dims %s f32 %d %dx%d x%d light %s)" % (case, kind, dims, f32, sc.width, sc.height, sc.steps, sc.light_pos)
        try:
            ref, refL = sc.render_shadow()
        except Exception as e:      # (the checker refuses what the product refuses: a light on the y axis, ...)
            skipped += 1
            continue
        try:
            push_scene(R, sc)
            f2b = R.shadowcoef().front_to_back
            R.set_option("kernel", 1)
            a = R.render()
            La = R.light_buffer()
        except Exception as e:
            print("%s: refused: %s" % (tag, str(e)[-160:]), flush=True)
            skipped += 1
            R.set_option("kernel", 0)
            continue
        e_cpu = float(np.abs(a - ref).max())
        e_L = float(np.abs(La - refL).max())
        worst_cpu = max(worst_cpu, e_cpu, e_L)
        if e_cpu > 1e-4 or e_L > 1e-4:
            print("%s: gather eye pass vs CPU checker %g, light buffer %g  <-- WRONG" % (tag, e_cpu, e_L), flush=True)
        R.set_option("kernel", 2)
        try:
            b = R.render()
            took += 1
            if not np.array_equal(a, b):
                print("%s: slice-ring vs gather max diff %g  <-- WRONG" % (tag, float(np.abs(a - b).max())), flush=True)
        except Exception as e:
            declined += 1
            why = str(e).split(":")[-1].strip()[:60]
            reasons[why] = reasons.get(why, 0) + 1
        R.set_option("kernel", 1)
        R.set_option("shadow_march", 0)
        try:
            c = R.render()
            Lc = R.light_buffer()
        finally:
            R.set_option("shadow_march", 1)
            R.set_option("kernel", 0)
        if not np.array_equal(La, Lc):
            print("%s: light buffers of the two forms differ by %g  <-- WRONG" % (tag, float(np.abs(La - Lc).max())), flush=True)
        d = float(np.abs(a - c).max())
        if f2b and d != 0.0:
            print("%s: front-to-back frame differs from the per-slice form by %g  <-- WRONG" % (tag, d), flush=True)
        if not f2b:
            worst_btf = max(worst_btf, d)
            if d > 2e-5:
                print("%s: back-to-front frame differs from the per-slice form by %g  <-- WRONG" % (tag, d), flush=True)
    print("%d cases: slice-ring kernel took %d, declined %d %s, skipped %d; worst vs CPU checker %.2e; worst re-association %.2e" % (
        cases, took, declined, reasons, skipped, worst_cpu, worst_btf), flush=True)
    print("slice-ring failures %d" % R.stat("slab_failures"), flush=True)
    R.close()


if __name__ == "__main__":
    main()
