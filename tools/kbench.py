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
    ap.add_argument("--synth", type=int, default=1, help="1 genvol spheres (bench input), 0 round 1's smooth shells")
    ap.add_argument("--variants", nargs="*", default=["kernel=1"])
    ap.add_argument("--tf", default="", help="'clear': an all-transparent table (what a frame costs when nothing is sampled); 'opaque'")
    ap.add_argument("--shadow", default="", help="BUFFER,QUALITY: half-angle-slicing shadows with the light at (3,4,-3)")
    a = ap.parse_args()
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    n = a.volume
    vghf, nrm = bench.make_volume(r, n, kind=a.synth)
    if a.u8:
        v8 = (vghf * 255.0).to(torch.uint8)
        r.upload_volume_device(v8.data_ptr(), (n, n, n), 3, 0, nrm.data_ptr())
    else:
        r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    del vghf, nrm
    torch.cuda.empty_cache()
    bench.configure(r, a.workload, n, a.size, a.planes)
    if a.tf:
        t = np.zeros((256, 256, 4), np.uint8)
        if a.tf == "opaque":
            t[:] = (200, 150, 100, 255)
        if a.tf == "ramp":    # the reference's default (value, gradient) ramp (NV20VolRen3D.cpp:1479-1486): dense
            t = np.load(os.path.join(ROOT, "tests", "golden", "tf_cfg2_deptex.npy"))
        r.set_tf2d(t, None)
    if a.pose.startswith("close"):
        # a close-up: strong perspective, rays far from the principal axis at the frame's edges
        dist = float(a.pose[5:] or 2.0)
        xform = bench.rotation((1, 1, 0), 30)
        M = bench.look_at((0.1, -0.05, -dist), bench.AT, bench.UP) @ xform @ bench.translate([-0.5, -0.5, -0.5])
        w = 0.42
        r.set_camera([float(v) for v in M.T.reshape(-1)], (-w, w, -w, w), (1.0, 20.0), a.size, a.size)
        r.set_shading("r8k", bench.LIGHT, (0.1, -0.05, -dist), bench.AT, [float(v) for v in xform.T.reshape(-1)], bench.INTENS)
    elif a.pose != "rot":
        ax, deg = {"id": ((0, 1, 0), 0), "x": ((0, 1, 0), 90), "y": ((1, 0, 0), 90), "back": ((0, 1, 0), 160),
                   "side": ((.2, 1, .1), 75), "r45": ((1, 1, 1), 45)}[a.pose]
        xform = bench.rotation(ax, deg)
        r.set_camera(bench.modelview(xform, (1.0, 1.0, 1.0)), bench.FRUSTUM, (1.0, 20.0), a.size, a.size)
        r.set_shading("r8k", bench.LIGHT, bench.EYE, bench.AT, [float(v) for v in xform.T.reshape(-1)], bench.INTENS)
    if a.shadow:
        buf, q = a.shadow.split(",")
        xform = bench.rotation((1, 1, 0), 30)
        r.set_shading("r8k", (3.0, 4.0, -3.0), bench.EYE, bench.AT, [float(v) for v in xform.T.reshape(-1)], bench.INTENS)
        r.set_shadow(1, int(buf), float(q))
    frame = torch.zeros((a.size * a.size, 4), dtype=torch.float32, device="cuda")
    base = None
    work = torch.cuda.Stream()
    for var in a.variants:
        for kv in var.split(","):
            k, v = kv.split("=")
            r.set_option(k, int(v))
        try:
            with torch.cuda.stream(work):
                t, kms, kn = bench.timed(r, a.frames, 2, frame, 1, None)
        except Exception as ex:  # (a forced configuration the launcher refuses: say so, go on with the next variant)
            print("%-32s refused: %s" % (var, str(ex).strip().splitlines()[-1][:150]), flush=True)
            try:
                r.render_device(frame.data_ptr())   # (a flagged frame makes the next call fail once: take that here)
            except Exception:
                pass
            continue
        kern, _, alg = r.last_frame_info()
        img = frame.cpu().numpy()
        st = r.stat("slab_status")
        if st:
            print("   !! slice-ring kernel status %d (1 = time-out, 2 = window outside its host bound): frame invalid" % st, flush=True)
        if base is None:
            base = img
        err = float(np.abs(img - base).max())
        print("%-32s kernel=%d  %.3f ms/frame  (event avg %.3f ms)  %.1f GB/s alg  frac %.3f  maxdiff_vs_first %.2e  alpha_mean %.4f"
              % (var, kern, t / a.frames * 1e3, kms, alg / (kms * 1e-3) / 1e9, alg / (kms * 1e-3) / 1e9 / 8000, err, img[:, 3].mean()),
              flush=True)
        if kern == 4:
            cfg = int(r.stat("cols_config"))
            sb = r.stat("cols_stream_bytes")
            print("   column-stream: CW x CH %d x %d, %d slots, shape %d, %d jobs; streamed %.2f GB = %.2f x alg -> %.0f GB/s; jobs: longest %.3f ms, sum %.1f ms (= %.3f ms per CU of 256), of which set-up %.1f ms; rays listed %.4g; layouts built %d" %
                  (cfg & 255, (cfg >> 8) & 255, (cfg >> 16) & 255, cfg >> 24, r.stat("cols_jobs"), sb / 1e9, sb / alg, sb / (kms * 1e-3) / 1e9,
                   r.stat("cols_job_ms_max"), r.stat("cols_job_ms_sum"), r.stat("cols_job_ms_sum") / 256, r.stat("cols_setup_ms_sum"), r.stat("cols_rays"), r.stat("cols_builds")), flush=True)
            if any(kv.split("=")[0] == "cols_counts" and int(kv.split("=")[1]) for kv in var.split(",")):
                it = r.stat("cols_iters") + 1e-9
                print("   samples taken %.4g, visible %.4g, slices streamed %.4g, segments %.4g; consumer wave-iterations %.4g: lanes with a sample %.1f%% of 64, taking one %.1f%%; iterations with a ray change %.1f%%, %.1f lanes each" %
                      (r.stat("cols_samples"), r.stat("cols_visible"), r.stat("cols_slices"), r.stat("cols_segments"), it, 100 * r.stat("cols_active_lanes") / (64 * it),
                       100 * r.stat("cols_samples") / (64 * it), 100 * r.stat("cols_switch_iters") / it, r.stat("cols_switch_lanes") / (r.stat("cols_switch_iters") + 1e-9)), flush=True)
        if kern == 2:
            print("   workgroups: longest %.3f ms, sum %.1f ms (= %.3f ms on every workgroup slot of 256 CUs x 2)" %
                  (r.stat("slab_tile_ms_max"), r.stat("slab_tile_ms_sum"), r.stat("slab_tile_ms_sum") / 512), flush=True)
        if any(kv.split("=")[0] == "lockstep" and int(kv.split("=")[1]) & 16 for kv in var.split(",")):
            it, act, ins, hit = (r.stat(k) for k in ("slab_iters", "slab_active_lanes", "slab_inside_lanes", "slab_hit_lanes"))
            print("   consumer wave-iterations %.4g: lanes active %.1f%%, inside %.1f%%, hit %.1f%% of 64; iterations with a hit %.1f%%" %
                  (it, 100 * act / (64 * it + 1e-9), 100 * ins / (64 * it + 1e-9), 100 * hit / (64 * it + 1e-9),
                   100 * r.stat("slab_iters_with_hit") / (it + 1e-9)), flush=True)
            print("   iterations that sample (some lane in a non-empty layer) %.1f%%; of those, with a lane whose OWN brick is flagged %.1f%%" %
                  (100 * r.stat("slab_iters_sampling") / (it + 1e-9), 100 * r.stat("slab_iters_own_brick") / (r.stat("slab_iters_sampling") + 1e-9)), flush=True)
            print("   mean (landed - slowest lane) at step time %.2f slices; waits %.3g; mean wstep %.2f" %
                  (r.stat("slab_lead_sum") / (it + 1e-9), r.stat("slab_waits"), r.stat("slab_wstep_sum") / (it + 1e-9)), flush=True)
            print("   mean share of its tile's slice range a wave no longer needs when it finishes: %.1f%%" %
                  (100 * r.stat("slab_dead_tail_sum") / (r.stat("slab_waves") + 1e-9)), flush=True)
            li, lw, lb, lt = (r.stat(k) for k in ("slab_loader_issue_kcyc", "slab_loader_wait_kcyc", "slab_loader_blocked_kcyc", "slab_loader_total_kcyc"))
            print("   loader 0 of every workgroup, sum of kilo-cycles: issue(+poll) %.4g  vmcnt wait %.4g  ring-blocked %.4g  total %.4g  (= %.3f ms per CU at 2.4 GHz if spread over 256 CUs)" %
                  (li, lw, lb, lt, lt * 1e3 / 256 / 2.4e9 * 1e3), flush=True)
        if any(kv.split("=")[0] == "lockstep" and int(kv.split("=")[1]) & 32 for kv in var.split(",")):
            tr = r.trace()
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            np.save(os.path.join(ROOT, 'gpurun_out', 'trace_%s.npy' % var.replace(',', '_').replace('=', '')), tr)
            tr = tr[tr[:, 1] != 0]
            t0 = tr[:, 0].astype(np.int64)
            t1 = tr[:, 1].astype(np.int64)
            base_t = t0.min()
            dur = (t1 - t0) / 100.0  # us
            cu = (tr[:, 2] >> 8) & 0xf
            se = (tr[:, 2] >> 13) & 0x7
            sh = (tr[:, 2] >> 12) & 1
            xcc = tr[:, 3] & 0xf
            key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
            if any(kv.split("=")[0] == "lockstep" and int(kv.split("=")[1]) & 128 for kv in var.split(",")):
                print("   set-up phases, mean us since workgroup start: slice range %.1f | windows tabled %.1f | flags + runs + last barrier %.1f | end %.1f" %
                      (tr[:, 4].mean() / 100.0, tr[:, 5].mean() / 100.0, tr[:, 6].mean() / 100.0, dur.mean()))
            print("   trace: %d workgroups on %d distinct (xcc,se,sh,cu); frame span %.1f us; WG duration min/mean/max %.1f/%.1f/%.1f us"
                  % (len(tr), len(np.unique(key)), (t1.max() - base_t) / 100.0, dur.min(), dur.mean(), dur.max()))
            busy = {}
            last = {}
            for k_, a_, b_ in zip(key, t0, t1):
                busy[k_] = busy.get(k_, 0) + (b_ - a_)
                last[k_] = max(last.get(k_, 0), b_)
            bz = np.array(list(busy.values())) / 100.0
            ends = (np.array(list(last.values())) - base_t) / 100.0
            print("   per-CU busy us min/mean/max %.1f/%.1f/%.1f ; last-end us min/mean/max %.1f/%.1f/%.1f" %
                  (bz.min(), bz.mean(), bz.max(), ends.min(), ends.mean(), ends.max()))
            for x in range(8):
                m = xcc == x
                if m.any():
                    print("   xcc %d: %4d WGs, sum busy %.0f us, first start %.1f last end %.1f, slices %d" %
                          (x, m.sum(), dur[m].sum(), (t0[m].min() - base_t) / 100.0, (t1[m].max() - base_t) / 100.0, int((tr[m, 3] >> 20).sum())))
    r.close()


if __name__ == "__main__":
    main()
