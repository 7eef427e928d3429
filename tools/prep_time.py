import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
pkg = bench.load_package()
r = pkg.Renderer(0)
for n in (512, 1024):
    out = torch.empty((n, n, n), dtype=torch.uint8, device="cuda")
    r.synth_volume_device(1, 1, (n, n, n), out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.synth_volume_device(1, 1, (n, n, n), out.data_ptr())
    torch.cuda.synchronize()
    print("genvol spheres+blur %d^3: %.2f ms" % (n, (time.perf_counter() - t0) * 1e3), flush=True)
    del out
