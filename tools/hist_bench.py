#!/usr/bin/env python3
"""Developer tool (GPU box): time smk_hist2d_device on an n^3 VGH volume (MetaVolume::hist2D's job)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    pkg = bench.load_package()
    r = pkg.Renderer(0)
    scalar = torch.empty((n, n, n), dtype=torch.uint8, device="cuda")
    r.synth_volume_device(0, 1, (n, n, n), scalar.data_ptr())
    vgh = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    r.make_vgh_device(scalar.data_ptr(), 0, (n, n, n), 1, vgh.data_ptr(), None)
    h = r.hist2d_device(vgh.data_ptr(), 3, (n, n, n))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        h = r.hist2d_device(vgh.data_ptr(), 3, (n, n, n))
    dt = (time.perf_counter() - t0) / 5
    print("%d^3 VGH: %.2f ms per histogram (incl. 256 KB read-back + log scaling on the host), %.0f GB/s of voxel bytes; %d non-empty bins"
          % (n, dt * 1e3, 3.0 * n ** 3 / dt / 1e9, int((h > 0).sum())))
    r.close()


if __name__ == "__main__":
    main()
