#!/usr/bin/env python3
"""Timeline of the PRODUCT slice-ring kernel's workgroups (developer tool, GPU box only): which share of the
workgroup slots is busy over the frame, when each XCD and CU runs dry.
    python tools/timeline.py [--volume 512] [--workload cfg3] [--frames 60] [--opts k=v,...]"""
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
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--opts", default="")
    ap.add_argument("--per-cu", type=int, default=2, help="workgroup slots per CU (2 small, 1 big)")
    a = ap.parse_args()
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    n = a.volume
    vghf, nrm = bench.make_volume(r, n)
    r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    del vghf, nrm
    bench.configure(r, a.workload, n, 1024, 512)
    r.set_option("kernel", 2)
    for kv in [x for x in a.opts.split(",") if x]:
        k, v = kv.split("=")
        r.set_option(k, int(v))
    frame = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(a.frames):
        r.render_device(frame.data_ptr(), None, st)
    torch.cuda.synchronize()
    r.timing_reset()
    for _ in range(10):
        r.render_device(frame.data_ptr(), None, st)
    torch.cuda.synchronize()
    kms, _ = r.timing_read()
    tr = r.trace()
    tr = tr[tr[:, 1] != 0]
    t0 = tr[:, 0].astype(np.int64)
    t1 = tr[:, 1].astype(np.int64)
    wrap = t1 < t0
    t1[wrap] += 1 << 32
    base = t0.min()
    s = (t0 - base) / 100.0
    e = (t1 - base) / 100.0
    span = e.max()
    cu = (tr[:, 2] >> 8) & 0xf
    sh = (tr[:, 2] >> 12) & 1
    se = (tr[:, 2] >> 13) & 0x7
    xcc = tr[:, 3] & 0xf
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ncu = len(np.unique(key))
    print("kernel %.3f ms (events); %d tiles with a record on %d CUs; span first start -> last end %.1f us; workgroup us min/mean/max %.1f/%.1f/%.1f; sum %.1f ms"
          % (kms, len(tr), ncu, span, (e - s).min(), (e - s).mean(), (e - s).max(), (e - s).sum() / 1e3))
    slots = ncu * a.per_cu
    print("mean load of a slot %.1f us (%d slots)" % ((e - s).sum() / slots, slots))
    # busy slots over time
    edges = np.linspace(0, span, 21)
    line = []
    for k in range(20):
        a0, a1 = edges[k], edges[k + 1]
        ov = np.clip(np.minimum(e, a1) - np.maximum(s, a0), 0, None).sum() / (a1 - a0)
        line.append("%3.0f" % (100 * ov / slots))
    print("busy slots, %% of %d, in 20 equal steps of the span: %s" % (slots, " ".join(line)))
    last_start = s.max()
    print("last workgroup starts at %.1f us; workgroups starting in the first 5 us: %d" % (last_start, int((s < 5).sum())))
    for x in range(8):
        m = xcc == x
        if m.any():
            cus = np.unique(key[m])
            ends = np.array([e[m & (key == c)].max() for c in cus])
            print("  xcc %d: %4d tiles, busy sum %.0f us, last start %.1f, CU run-dry times min/mean/max %.1f/%.1f/%.1f us"
                  % (x, m.sum(), (e - s)[m].sum(), s[m].max(), ends.min(), ends.mean(), ends.max()))
    # how long do the workgroups that start first take, and the ones that start last
    order = np.argsort(s)
    q = len(order) // 4
    print("duration of the first quarter of the workgroups to start %.1f us mean, of the last quarter %.1f us" % ((e - s)[order[:q]].mean(), (e - s)[order[-q:]].mean()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "timeline.npy"), tr)
    r.close()


if __name__ == "__main__":
    main()
