import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
pkg = bench.load_package()
n = 512
RANKS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (1, 2, 4, 8)
SPLITS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0,)
for nranks in RANKS:
    for rank in range(0, nranks, max(1, nranks // 2)):
        r = pkg.Renderer(0)
        if nranks > 1:
            r.set_shard(rank, nranks)
        vghf, nrm = bench.make_volume(r, n)
        r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
        del vghf, nrm
        torch.cuda.empty_cache()
        bench.configure(r, "cfg3", n, 1024, 512)
        frame = torch.zeros((1024 * 1024, 4), dtype=torch.float32, device="cuda")
        for split in SPLITS:
            r.set_option("slab_split", split)
            for _ in range(70):     # (the schedule's weights, and with them the cuts, settle over the first frames)
                r.render_device(frame.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            r.timing_reset()
            t0 = time.perf_counter()
            K = 20
            for _ in range(K):
                r.render_device(frame.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / K * 1e3
            kms, kn = r.timing_read()
            print("nranks %d rank %d split %d: %.3f ms/frame wall, kernel %.3f ms, kernel id %d; tiles cut %d, workgroups %d: longest %.3f ms, sum %.1f ms; slices streamed %.3f" %
                  (nranks, rank, split, t, kms, r.last_frame_info()[0], r.stat("slab_split_tiles"), r.stat("slab_workgroups"), r.stat("slab_tile_ms_max"), r.stat("slab_tile_ms_sum"), r.stat("slab_streamed_fraction")), flush=True)
        r.close()
