#!/usr/bin/env python3
"""Developer tool (GPU box): the two twelve-wave workgroup shapes of the slice-ring kernel (option tile 21 = 16x40 pixels,
20 = 40x16) in turn, five times each, on the cfg 3 frame at the given rotation angles -- same process, same box.
    python tools/shape_ab.py [angle ...]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
n = 512
pkg = bench.load_package()
r = pkg.Renderer(0)
vghf, nrm = bench.make_volume(r, n)
r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
del vghf, nrm
bench.configure(r, "cfg3", n, 1024, 512)
r.set_option("kernel", 2)
angles = [float(a) for a in sys.argv[1:]] or [30.0]
frame = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for ang in angles:
  xform = bench.rotation((1, 1, 0), ang)
  r.set_camera(bench.modelview(xform, (1.0, 1.0, 1.0)), bench.FRUSTUM, (1.0, 20.0), 1024, 1024)
  res = {21: [], 20: []}
  for rep in range(5):
      for tile in (21, 20):
          r.set_option("tile", tile)
          for _ in range(70):
              r.render_device(frame.data_ptr(), None, st)
          torch.cuda.synchronize()
          r.timing_reset()
          for _ in range(20):
              r.render_device(frame.data_ptr(), None, st)
          torch.cuda.synchronize()
          res[tile].append(r.timing_read()[0])
  print("angle %g  16x40:" % ang, " ".join("%.3f" % v for v in res[21]), " mean %.4f" % (sum(res[21]) / 5))
  print("angle %g  40x16:" % ang, " ".join("%.3f" % v for v in res[20]), " mean %.4f" % (sum(res[20]) / 5))
