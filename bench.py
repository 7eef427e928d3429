#!/usr/bin/env python3
"""bench.py -- headline benchmark of the volume ray-march hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): Msamples/s (+ fps) of one frame of the 512^3 VGH volume at
1024^2 x 512 planes: config 3 -- f32 VGH, 2-D LevWidget transfer function, gradient Phong.
A "step" is one frame; inputs are synthetic and already resident in HBM when the timed region
starts.  N > 1 is sort-last: the volume is sharded by brick region (strong scaling), each rank
ray-marches the full viewport against its shard and the frames are merged by one RCCL
all-to-all + ordered "over" + gather; that merge is inside the timed region.

Extra objects on the JSON line:
  roofline      dominant kernel (the ray-marcher) vs the HBM roofline, HIP-event timed in here
  cpu_baseline  the CPU checker (oracle/, kind "port") timed on this box's host cores on a
                bounded band of rows of the same frame (rank 0, N=1 only)
  north_star    same kernel on the 1024^3 f32 VGH volume (config 4's single-GPU case), the
                case BASELINE.json's >= 60 % HBM-roofline target is quoted on (N=1 only)
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, Chip-level parameters)


def pmc_traffic(workload):
    """HBM bytes per launch of the ray-march kernel from the PMC passes of THIS command
    (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs; FETCH_SIZE doubled for gfx950),
    summarised by tools/summarise_profiles.py into profiles/.  Counters cannot be read from inside
    the process, so the figure is the committed one for the default workload, or null."""
    paths = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_traffic.json")) \
        if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if not paths:
        return None
    with open(os.path.join(ROOT, "profiles", paths[-1])) as f:
        t = json.load(f)
    return t.get(workload, {}).get("hbm_bytes_per_launch")


def pmc_insts(workload):
    """Per-launch instruction counters of the ray-march kernel from the committed SQ passes
    (profiles/rNN_pmc_sq_<cfg3|ns>.txt, written by tools/profile_round.sh + tools/pmc_summary.py)."""
    d = os.path.join(ROOT, "profiles")
    suffix = "_pmc_sq_%s.txt" % {"cfg3": "cfg3", "north_star": "ns", "cfg3_dense": "cfg3_dense", "north_star_dense": "ns_dense", "cfg5": "cfg5"}.get(workload, "ns")
    paths = sorted(p for p in os.listdir(d) if p.endswith(suffix)) if os.path.isdir(d) else []
    if not paths:
        return None
    out = {}
    with open(os.path.join(d, paths[-1])) as f:
        for line in f:
            w = line.split()
            if len(w) >= 5 and w[0].startswith("SQ_") and w[-2] == "mean":
                out[w[0]] = float(w[-1])
    out["file"] = "profiles/" + paths[-1]
    return out


VALU_PEAK_GINST = 256 * 4 * 2.4 / 4.0   # wave64 VALU instructions/ns the chip can issue: 256 CUs x 4 SIMDs, one per 4 cycles at 2.4 GHz


def roofline_valu(kms, workload):
    """Second roofline of the 512^3 frame: it is bound by vector-instruction issue, not by HBM.  achieved =
    wave-level VALU instructions per launch (SQ_INSTS_VALU of the committed PMC pass of this kernel on this
    input) over the kernel time measured live; peak = one VALU instruction per SIMD every 4 cycles."""
    c = pmc_insts(workload)
    if not c or "SQ_INSTS_VALU" not in c or kms <= 0:
        return None
    ach = c["SQ_INSTS_VALU"] / (kms * 1e-3) / 1e9
    out = {"bound": "valu", "achieved": ach, "peak": VALU_PEAK_GINST, "unit": "Gwave-inst/s", "frac": ach / VALU_PEAK_GINST,
           "valu_insts_per_launch": c["SQ_INSTS_VALU"], "salu_insts_per_launch": c.get("SQ_INSTS_SALU"),
           "lds_insts_per_launch": c.get("SQ_INSTS_LDS"), "kernel_ms": kms, "source": c["file"],
           "note": "packed-fp32 instructions (v_pk_fma_f32) count once but occupy the SIMD twice as long (tools/valu_probe.hip)"}
    if c.get("SQ_INSTS_SALU") is not None:
        out["salu_per_valu"] = c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"]
    if c.get("SQ_LDS_BANK_CONFLICT") is not None and c.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    return out


def load_package():
    name = "simian_spacemonkey_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(ROOT, "simian-spacemonkey_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"),
                                                  submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# --------------------------------------------------------------------------- camera (no oracle)

def rotation(axis, deg):
    a = np.asarray(axis, np.float64)
    a /= np.linalg.norm(a)
    t = np.deg2rad(deg)
    c, s = np.cos(t), np.sin(t)
    x, y, z = a
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                  [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                  [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])
    M = np.eye(4)
    M[:3, :3] = R
    return M


def look_at(eye, at, up):
    eye, at, up = (np.asarray(v, np.float64) for v in (eye, at, up))
    f = at - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    M = np.eye(4)
    M[0, :3], M[1, :3], M[2, :3] = s, u, -f
    T = np.eye(4)
    T[:3, 3] = -eye
    return M @ T


def translate(v):
    T = np.eye(4)
    T[:3, 3] = v
    return T


EYE, AT, UP = (0, 0, -7), (0, 0, 0), (0, 1, 0)          # gluvv.cpp:263-271
LIGHT, INTENS = (0, 0, -5), 0.75                        # gluvv.cpp:293-305
FRUSTUM = (-0.5 / 7, 0.5 / 7, -0.5 / 7, 0.5 / 7)        # unit volume fills the view


def modelview(xform, fsize):
    """LookAt * T(trans=0) * R(xform) * T(-fSize/2)  (VolumeRenderable.cpp:40-49)"""
    M = look_at(EYE, AT, UP) @ xform @ translate([-f / 2 for f in fsize])
    return [float(v) for v in M.T.reshape(-1)]  # column-major


# --------------------------------------------------------------------------- workload set-up

def make_volume(r, n, seed=1, kind=1):
    """synthetic scalar -> f32 VGH + u8 normals, all on the GPU (smk_prep.hip); returns tensors.
    kind 1: the reference's own test volume, genvol spheres (`-spheres 4 -p 10 -pscale .7 -pwrap 3 3 3
    -pabs -blur -bw 1 1 1 .7`, srand(seed)), generated on the GPU byte for byte as the CPU
    checker's Perlin-pinned restatement does (tests/test_gpu_prep.py); kind 0: round 1's smooth shells."""
    dims = (n, n, n)
    scalar = torch.empty((n, n, n), dtype=torch.uint8, device="cuda")
    r.synth_volume_device(kind, seed, dims, scalar.data_ptr())
    vgh8 = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    vghf = torch.empty((n, n, n, 3), dtype=torch.float32, device="cuda")
    r.make_vgh_device(scalar.data_ptr(), 0, dims, 1, vgh8.data_ptr(), vghf.data_ptr())
    nrm = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    r.normals_vgh_device(vgh8.data_ptr(), 3, dims, 0, nrm.data_ptr())
    del scalar, vgh8
    return vghf, nrm


def configure(r, workload, n, size, planes):
    g = os.path.join(ROOT, "tests", "golden")
    xform = rotation((1, 1, 0), 30)                     # SURVEY 8d second pose
    xf = [float(v) for v in xform.T.reshape(-1)]
    if workload == "cfg3":
        r.set_tf2d(np.load(os.path.join(g, "tf_cfg3_levwidget.npy")), None)
    else:  # cfg4: separable (v,g) x (h) classification = the reference's live 3-D TF
        r.set_tf2d(np.load(os.path.join(g, "tf_cfg3_levwidget.npy")),
                   np.load(os.path.join(g, "tf_h_slider05.npy")))
    mv = modelview(xform, (1.0, 1.0, 1.0))
    r.set_camera(mv, FRUSTUM, (1.0, 20.0), size, size)
    r.set_sampling(0.0, planes, 1.0, 1)
    r.set_shading("r8k", LIGHT, EYE, AT, xf, INTENS)
    r.set_perturb(None, None, None)
    return xform, mv


def run_frames(r, nframes, frame, world, pipeline):
    """nframes frames back to back; returns nothing (caller brackets with barrier+sync)"""
    for _ in range(nframes):
        if world == 1:
            r.render_device(frame.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        else:
            pipe, order = pipeline
            pipe.frame(frame, order)
    if world > 1:
        pipeline[0].drain()


SETTLE_FRAMES = 24  # untimed set-up before the warm-up: auto mode times both ray-marchers (4 frames), and the tile schedule
                    # adopts the workgroups' measured durations at once, then after 4, 8 and 16 frames (steady state from there)
#                     and the tile schedule takes its weights from a measured frame


def timed(r, K, W, frame, world, cstate, settle=True):
    if settle:
        run_frames(r, SETTLE_FRAMES, frame, world, cstate)
    run_frames(r, W, frame, world, cstate)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    r.timing_reset()
    t0 = time.perf_counter()
    run_frames(r, K, frame, world, cstate)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([t], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    kms, kn = r.timing_read()
    return t, kms, kn


def roofline(r, kms, alg_bytes, size, note=None, traffic=None):
    """HBM roofline object of the ray-march kernel: algorithmic bytes per launch (every stored voxel once + TF
    + the 16-byte RGBA frame, smk_last_frame_info) over the kernel's HIP-event time.  The slice-ring
    kernel's loaders stop streaming a tile once all its rays are saturated: the voxel bytes are scaled by
    the share of the tiles' slices that was really streamed (counted in-kernel), so an opaque transfer
    function cannot inflate the fraction."""
    frame_bytes = 16.0 * size * size
    streamed = float(r.stat("slab_streamed_fraction"))
    net = (alg_bytes - frame_bytes) * streamed + frame_bytes
    ach = net / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
           "frac_of_achievable_6300": ach / 6300.0,   # float4-copy ceiling measured on MI355X (MI355X_MICROARCH.md, HBM)
           "traffic": traffic,
           "traffic_source": "committed profiles/*_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                             "on this code (counters cannot be read in-process)" if traffic is not None else None,
           "kernel_ms": kms,
           "algorithmic_bytes_per_launch": net, "algorithmic_bytes_if_every_slice_streamed": alg_bytes,
           "slices_streamed_fraction": streamed,
           # SURVEY 8(d)'s formula (every stored voxel once) over the same time: NOT a bandwidth once bricks that cannot
           # hold a visible sample are skipped (option "bricks", smk_bricks.hip) -- it can exceed the HBM peak
           "each_voxel_once_GBps": alg_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0}
    if note:
        out["note"] = note
    return out


def sample_counts(r, frame, size, planes):
    """the nominal sample count next to the samples that lie inside the volume and the samples the kernel
    really interpolates (with empty-space skipping on, samples of layers in which nothing can be visible are never taken;
    a ray stops at alpha == 1).  Counted outside the timed region: the in-volume count by the renderers' membership test
    alone (smk_count_samples), the taken count by the slice-ring kernel's diagnostic instance (option lockstep bit 16)."""
    out = {"nominal": int(size) * int(size) * int(planes), "in_volume": r.count_samples(), "taken": None}
    kernel = r.last_frame_info()[0]
    if kernel == 2:
        r.set_option("lockstep", 1 | 16)
        r.render_device(frame.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if r.last_frame_info()[0] == 2:
            out["taken"] = int(r.stat("slab_inside_lanes"))
        r.set_option("lockstep", 1)
        r.render_device(frame.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    return out


def workgroup_load(r):
    """longest workgroup against the mean load of a workgroup slot (256 CUs x 2 small or x 1 big workgroups), ms"""
    if r.last_frame_info()[0] != 2:
        return None
    mx, sm = float(r.stat("slab_tile_ms_max")), float(r.stat("slab_tile_ms_sum"))
    return {"longest_workgroup_ms": mx, "sum_ms": sm, "sum_ms_per_cu": sm / 256.0, "tiles_cut_in_depth": int(r.stat("slab_split_tiles")),
            "workgroups": int(r.stat("slab_workgroups"))}


def dense_leg(r, work, frame, size, planes, steps, workload_key):
    """The worst case for empty-space skipping at full size: the reference's default (value, gradient) ramp
    (NV20VolRen3D.cpp:1479-1486) -- every sample is visible, classified, shaded and blended; the flags skip nothing."""
    K = max(3, min(steps, 8))
    g = os.path.join(ROOT, "tests", "golden")
    r.set_tf2d(np.load(os.path.join(g, "tf_cfg2_deptex.npy")), None)
    with torch.cuda.stream(work):
        run_frames(r, 60, frame, 1, None)
        t, kms, kn = timed(r, K, 2, frame, 1, None)
    kernel, _, alg = r.last_frame_info()
    out = {"workload": "the same volume, camera and shading under the reference's default ramp table (dense: every sample visible)",
           "ms_per_frame": t / K * 1e3, "Msamples_per_s": float(size) * size * planes / (t / K) / 1e6,
           "kernel": {1: "gather", 2: "slab-staged", 4: "column-stream"}.get(kernel, str(kernel)),
           "roofline": roofline(r, kms, alg, size), "roofline_valu": roofline_valu(kms, workload_key),
           "samples": sample_counts(r, frame, size, planes), "workgroups": workgroup_load(r)}
    return out


def no_flags_leg(r, work, frame, size, planes, steps):
    """The same frame with empty-space skipping off (option "bricks" 0): every slice of every tile is streamed and every
    sample interpolated and classified -- the streaming kernel's own roofline.  The frames are bit-identical
    (tests/test_gpu_bricks.py); the flags only remove work whose result is exactly transparent."""
    K = max(3, min(steps, 10))
    keep = frame.clone()
    r.set_option("bricks", 0)
    with torch.cuda.stream(work):
        # (the tile schedule follows measured workgroup durations, re-adopted every 32 frames of an unchanged tiling:
        #  the switch changes every duration, so the schedule gets time to follow before anything is timed)
        run_frames(r, 80, frame, 1, None)
        t, kms, kn = timed(r, K, 2, frame, 1, None)
    kernel, _, alg = r.last_frame_info()
    rl = roofline(r, kms, alg, size)
    same = bool(torch.equal(keep, frame))
    r.set_option("bricks", 1)
    with torch.cuda.stream(work):      # back to the product's default, schedule settled again
        run_frames(r, 80, frame, 1, None)
    return {"ms_per_frame": t / K * 1e3, "Msamples_per_s": float(size) * size * planes / (t / K) / 1e6,
            "kernel": {1: "gather", 2: "slab-staged"}.get(kernel, str(kernel)), "roofline": rl,
            "frame_bit_identical_to_the_one_with_flags": same}


def column_stream_leg(r, work, frame, size, planes, steps):
    """The same frame on the column-stream kernel (option kernel = 3, smk_cols.hip) with empty-space skipping off: the
    volume re-laid out in columns with their halo, each streamed sequentially once.  Its byte count is what its loaders
    move (the layout's bytes: every voxel and its column's halo once), not an estimate."""
    K = max(3, min(steps, 8))
    keep = frame.clone()
    r.set_option("bricks", 0)
    r.set_option("kernel", 3)
    try:
        with torch.cuda.stream(work):
            run_frames(r, 3, frame, 1, None)
            t, kms, kn = timed(r, K, 1, frame, 1, None, settle=False)
        kernel, _, alg = r.last_frame_info()
        cfg = int(r.stat("cols_config"))
        sb = float(r.stat("cols_stream_bytes"))
        out = {"ms_per_frame": t / K * 1e3, "kernel_ms": kms, "kernel": "column-stream" if kernel == 4 else str(kernel),
               "column_cells": [cfg & 255, (cfg >> 8) & 255], "ring_slots": (cfg >> 16) & 255, "jobs": int(r.stat("cols_jobs")),
               "streamed_bytes_per_launch": sb, "streamed_over_algorithmic": sb / alg if alg else None,
               "streamed_GBps": sb / (kms * 1e-3) / 1e9 if kms > 0 else None,
               "algorithmic_GBps": alg / (kms * 1e-3) / 1e9 if kms > 0 else None, "frac_of_8000": alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS if kms > 0 else None,
               "max_abs_diff_vs_the_slice_ring_frame": float((keep - frame).abs().max().item()),
               "note": "same samples, blend re-associated per column job: <= 2e-5 (tests/test_gpu_cols.py); bound by vector-instruction issue, "
                       "not by its stream (DESIGN.md 4e)"}
    except Exception as e:  # noqa: BLE001  (reported, never silent)
        out = {"error": str(e)[:300]}
    r.set_option("kernel", 0)
    r.set_option("bricks", 1)
    with torch.cuda.stream(work):
        run_frames(r, 40, frame, 1, None)
    return out


def libc_noise_tex(n=32):
    """R8kVolRen3D_cpy::createNoiseTex (:2392-2436): n^3 RGBA8 from libc's srand(1)/rand(), as the adapter makes it"""
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    v = np.array([libc.rand() for _ in range(n * n * n * 4)], dtype=np.float64)
    t = (v.astype(np.float32) / np.float32(2147483647)).astype(np.float64) * .5 + .5 + 1.0 / 512
    return (t * 255).astype(np.int32).astype(np.uint8).reshape(n, n, n, 4)


def north_star_parity(vghf_h, nrm_h, tf_eff, size, planes, xform, mv, gpu_frame, nrays=200):
    """max |GPU - CPU checker| over `nrays` random rays of the north-star frame (the checker marches per ray)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    sc = O.Scene(vghf_h, grad=nrm_h)
    sc.tf_mode, sc.tf_vg = 1, tf_eff
    sc.tf_h, sc.third_axis = np.load(os.path.join(ROOT, "tests", "golden", "tf_h_slider05.npy")), 1
    sc.width = sc.height = size
    sc.steps = planes
    sc.xform = [float(v) for v in xform.T.reshape(-1)]
    sc.mv_override = mv
    sc.shade_mode, sc.use_spec = 1, 1
    sc.frustum = FRUSTUM
    pix = np.random.default_rng(11).integers(size // 8, size - size // 8, size=(nrays, 2)).astype(np.int32)
    ref = sc.render_pixels(pix)
    return float(np.abs(gpu_frame[pix[:, 1], pix[:, 0]] - ref).max())


def extra_legs(r, frame, work, steps):
    """Secondary timings on the same JSON line (N=1): transfer functions that are NOT kind to the kernel,
    and BASELINE config 5.  Each: K frames after a settle, HIP-event kernel time, net roofline."""
    g = os.path.join(ROOT, "tests", "golden")
    out = {}
    K = max(3, min(steps, 8))

    def run(name, size, planes, describe, kernel_opt=0, valu_key=None):
        fr = frame[:size * size]
        r.set_option("kernel", kernel_opt)
        with torch.cuda.stream(work):
            t, kms, kn = timed(r, K, 1, fr, 1, None)
        r.set_option("kernel", 0)
        kernel, _, alg = r.last_frame_info()
        out[name] = {"workload": describe, "ms_per_frame": t / K * 1e3, "Msamples_per_s": float(size) * size * planes / (t / K) / 1e6,
                     "kernel": {1: "gather", 2: "slab-staged", 4: "column-stream"}.get(kernel, str(kernel)),
                     "roofline": roofline(r, kms, alg, size)}
        if valu_key:   # (a leg bound by vector-instruction issue: its own SQ pass under profiles/)
            out[name]["roofline_valu"] = roofline_valu(kms, valu_key)

    # cfg2: 256^3 u8 VGH, the reference's default deptex ramp (NV20VolRen3D.cpp:1479-1486), 512^2 x 256, no shading
    n = 256
    scalar = torch.empty((n, n, n), dtype=torch.uint8, device="cuda")
    r.synth_volume_device(1, 1, (n, n, n), scalar.data_ptr())
    vgh8 = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    r.make_vgh_device(scalar.data_ptr(), 0, (n, n, n), 1, vgh8.data_ptr(), None)
    r.upload_volume_device(vgh8.data_ptr(), (n, n, n), 3, 0, None)
    del scalar
    xform = rotation((1, 1, 0), 30)
    xf = [float(v) for v in xform.T.reshape(-1)]
    r.set_tf2d(np.load(os.path.join(g, "tf_cfg2_deptex.npy")), None)
    r.set_camera(modelview(xform, (1.0, 1.0, 1.0)), FRUSTUM, (1.0, 20.0), 512, 512)
    r.set_sampling(0.0, 256, 1.0, 1)
    r.set_shading("none", LIGHT, EYE, AT, xf, INTENS)
    r.set_perturb(None, None, None)
    run("cfg2_default_ramp", 512, 256, "cfg2: 256^3 u8 VGH (genvol spheres), the reference's default (value, gradient) ramp table, "
        "512x512x256, unshaded: a DENSE transfer function (every sample classified and blended)")
    del vgh8
    # an opaque table on the headline volume: rays saturate within a few samples, the loaders stop early
    n = 512
    vghf, nrm = make_volume(r, n)
    r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    del vghf, nrm
    torch.cuda.empty_cache()
    configure(r, "cfg3", n, 1024, 512)
    opaque = np.load(os.path.join(g, "tf_cfg3_levwidget.npy")).copy()
    opaque[..., 3] = 255
    opaque[..., :3] = np.maximum(opaque[..., :3], 64)
    r.set_tf2d(opaque, None)
    # (slice-ring kernel forced: its loaders count what they stream; in auto mode the gather kernel wins this
    #  frame -- a ray is over after a few samples -- and reads a fraction of the volume nobody counts)
    run("opaque_tf", 1024, 512, "cfg3's volume and camera with an OPAQUE table (alpha 255 everywhere), slice-ring kernel: the bytes "
        "counted are those of the slices really streamed", kernel_opt=2)
    run("opaque_tf_auto", 1024, 512, "the same frame in auto mode (whichever kernel measured faster); no byte count for the gather kernel")
    if out["opaque_tf_auto"]["kernel"] == "gather":
        out["opaque_tf_auto"]["roofline"] = None
    # the worst case for empty-space skipping at the headline size: cfg 3 under the reference's default ramp
    configure(r, "cfg3", n, 1024, 512)
    out["cfg3_dense_ramp"] = dense_leg(r, work, frame[:1024 * 1024], 1024, 512, steps, "cfg3_dense")
    # cfg3 with the shadow check box on (half-angle slicing, light buffer 1024 x quality .5 = 512^2, light up-left of the eye)
    configure(r, "cfg3", n, 1024, 512)
    r.set_shading("r8k", (3.0, 4.0, -3.0), EYE, AT, xf, INTENS)
    r.set_shadow(1, 1024, 0.5)
    run("cfg3_shadows", 1024, 512, "cfg3 with gluvv.light.shadow: 512 half-angle slices; two marches -- one per texel of the 512^2 light buffer "
        "(8 texels x 8 consecutive slices per wave), keeping every slice's buffer, then the eye pass as a frame of the ray-marcher over "
        "the half-angle slices that looks each sample's slice up (smk_shadow.hip); kernel_ms covers both")
    out["cfg3_shadows"]["kernel"] = "light march + " + out["cfg3_shadows"]["kernel"]
    out["cfg3_shadows"]["roofline"] = None   # two kernels of different kinds (a gather-bound march, the slice-ring kernel): no single roofline
    r.set_option("shadow_march", 0)
    run("cfg3_shadows_per_slice", 1024, 512, "the same frame as a launch per slice (option shadow_march 0, the form of rounds 1-2)")
    out["cfg3_shadows_per_slice"]["kernel"] = "shadow slices"
    out["cfg3_shadows_per_slice"]["roofline"] = None
    r.set_option("shadow_march", 1)
    r.set_shadow(0)
    # continuity with round 1: the north-star frame on round 1's input (smooth noisy shells, whose rays saturate
    # earlier: whole tiles stop streaming, which round 1's byte count ignored)
    n1 = 1024
    vghf, nrm = make_volume(r, n1, kind=0)
    r.upload_volume_device(vghf.data_ptr(), (n1, n1, n1), 3, 1, nrm.data_ptr())
    del vghf, nrm
    torch.cuda.empty_cache()
    configure(r, "cfg4", n1, 1024, 512)
    run("north_star_on_round1_input", 1024, 512, "the north-star frame on round 1's synthetic volume (smk_synth_volume_device kind 0)")
    # the same 1024^3 frame from a u8 VGH volume (what the reference's loader produces: MetaVolume quantises to bytes):
    # 8-byte packed voxels carry 6 bytes of data + normal
    vghf, nrm = make_volume(r, n1)
    v8 = (vghf * 255.0).to(torch.uint8)
    del vghf
    r.upload_volume_device(v8.data_ptr(), (n1, n1, n1), 3, 0, nrm.data_ptr())
    del v8, nrm
    torch.cuda.empty_cache()
    configure(r, "cfg4", n1, 1024, 512)
    run("north_star_u8", 1024, 512, "the north-star frame from a 1024^3 u8 VGH volume (6 algorithmic bytes per voxel in 8-byte packed voxels)")
    # BASELINE config 5 on one GPU: two 512^3 fields merged on the GPU (mergeMV + addG), dense 3-D table,
    # noise-perturbed fetch (createNoiseTex's texture, gluvvui's default weights (.2, 0) would displace by 51
    # voxels: SURVEY 8d's (.2, .1) scaled to the volume, see DESIGN), 1024^2 x 1024
    fields = torch.empty((n, n, n, 2), dtype=torch.uint8, device="cuda")
    one = torch.empty((n, n, n), dtype=torch.uint8, device="cuda")
    for e, seed in enumerate((1, 2)):
        r.synth_volume_device(1, seed, (n, n, n), one.data_ptr())
        fields[..., e] = one
    del one
    merged = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    mnrm = torch.empty((n, n, n, 3), dtype=torch.uint8, device="cuda")
    r.merge_fields_device(fields.data_ptr(), 2, (n, n, n), merged.data_ptr(), mnrm.data_ptr())
    del fields
    r.upload_volume_device(merged.data_ptr(), (n, n, n), 3, 0, mnrm.data_ptr(), dmode="V2G")
    del merged, mnrm
    torch.cuda.empty_cache()
    # the dense (v, g, h) table at the reference's own size, 256 x 256 x 4 "panes" (TFWidgetRen.cpp:98-100): cfg 3's
    # LevWidget table in every pane, opacity scaled per pane the way the widgets' boundary emphasis does for every
    # sheet but the second (LevWidget.cpp:704-761)
    if os.environ.get("SMK_BENCH_CFG5_TABLE") == "random16":   # (round 1's table: 16^3 random colours, every sample visible)
        rng = np.random.default_rng(5)
        t3 = rng.integers(0, 256, (16, 16, 16, 4), dtype=np.uint8)
        t3[..., 3] = (t3[..., 3].astype(np.float32) * 0.25).astype(np.uint8)
    else:
        pane = np.load(os.path.join(g, "tf_cfg3_levwidget.npy"))
        t3 = np.stack([pane] * 4).copy()
        for h_, be in enumerate((0.4, 1.0, 0.7, 0.4)):
            t3[h_, ..., 3] = (pane[..., 3].astype(np.float32) * be).astype(np.uint8)
    r.set_option("tf_raw", 1)
    r.set_tf3d(t3)
    r.set_camera(modelview(xform, (1.0, 1.0, 1.0)), FRUSTUM, (1.0, 20.0), 1024, 1024)
    r.set_sampling(0.0, 1024, 1.0, 1)
    r.set_shading("r8k", LIGHT, EYE, AT, xf, INTENS)
    run("cfg5_unperturbed", 1024, 1024, "cfg5 without the perturbation: 2 x 512^3 u8 fields merged (V2G), dense 3-D table %dx%dx%d, R8k Phong, 1024x1024x1024" % (t3.shape[2], t3.shape[1], t3.shape[0]))
    # SURVEY 8(d)'s configuration: weights (.2, .1), scales (.2, 2.1) (R8kVolRen3D_cpy.cpp:1590-1595, gluvvui.cpp:213-267) --
    # a displacement of up to +-77 voxels at 512^3: the brick flags' reach covers the volume, every in-volume sample pays
    # its two noise lookups
    r.set_perturb(libc_noise_tex(32), (.2, .1, 0, 0), (.2, 2.1, 4.5, 8.7))
    run("cfg5", 1024, 1024, "cfg5 as SURVEY 8(d) states it: noise-perturbed fetch, 32^3 noise, weights .2/.1, scales .2/2.1: gather kernel", valu_key="cfg5")
    # ... and the same at a tenth of the weights (round 2's leg: displacements of +-7.7 voxels, which the flags can bracket)
    r.set_perturb(libc_noise_tex(32), (.02, .01, 0, 0), (.2, 2.1, 4.5, 8.7))
    run("cfg5_tenth_of_the_weights", 1024, 1024, "cfg5 with weights .02/.01 (a tenth of the stated ones; NOT the configuration SURVEY 8(d) names): gather kernel")
    r.set_perturb(None, None, None)
    r.set_option("tf_raw", 0)
    return out


def cpu_baseline(vghf, nrm, tf_path, size, planes, xform, mv, gpu_frame, budget_s=12.0):
    """the CPU checker on the host cores, same frame, a bounded band of rows around the centre"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    sc = O.Scene(vghf.cpu().numpy(), grad=nrm.cpu().numpy())
    sc.tf_mode, sc.tf_vg = 1, tf_path   # the opacity-corrected table the GPU frame used
    sc.width = sc.height = size
    sc.steps = planes
    sc.xform = [float(v) for v in xform.T.reshape(-1)]
    sc.mv_override = mv              # the very matrix the GPU frame was rendered with
    sc.shade_mode, sc.use_spec = 1, 1
    sc.frustum = FRUSTUM
    mid = size // 2
    t0 = time.perf_counter()
    sc.render(rows=(mid, mid + 4))
    per_row = (time.perf_counter() - t0) / 4
    rows = int(max(8, min(size, budget_s / max(per_row, 1e-6))))
    rows -= rows % 2
    a, b = mid - rows // 2, mid + rows // 2
    t0 = time.perf_counter()
    img = sc.render(rows=(a, b))
    dt = time.perf_counter() - t0
    passes = 1
    while dt < 0.8 * budget_s and passes < 8:      # a many-core host finishes the band early: repeat it
        sc.render(rows=(a, b))
        passes += 1
        dt = time.perf_counter() - t0
    err = float(np.abs(img[a:b] - gpu_frame[a:b]).max())
    import ctypes
    try:
        threads = ctypes.CDLL("libgomp.so.1").omp_get_max_threads()
    except OSError:
        threads = os.cpu_count()
    return {"value": passes * rows * size * planes / dt / 1e6, "unit": "Msamples/s", "cores": int(threads),
            "kind": "port",
            "sample": "rows %d..%d of the same %dx%dx%d frame, %d pass%s (%.1f s of CPU work)"
                      % (a, b, size, size, planes, passes, "" if passes == 1 else "es", dt),
            "host_cpus": os.cpu_count(), "parity_max_abs_err_vs_gpu": err}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command under
    torch.distributed.run (one process per GPU), relay their output and exit with their code.  Runs
    before anything touches the GPU.  On a box with fewer than N GPUs the ranks rehearse on cuda:0
    (SMK_BENCH_REHEARSE=1: plumbing only, labelled as such on the JSON line)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < n:      # (counting devices does not initialise the GPU)
        env["SMK_BENCH_REHEARSE"] = "1"
        if n > 6:
            print("bench.py: %d ranks cannot share one GPU (process guard); refusing" % n, file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--planes", type=int, default=512)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 gather, 2 slab-staged")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-north-star", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary timings (dense / opaque tables, config 5)")
    ap.add_argument("--north-star-volume", type=int, default=1024)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # SMK_BENCH_REHEARSE=1: developer rehearsal of the N>1 plumbing on a one-GPU box -- every rank
    # uses cuda:0 and the layers travel over gloo through host memory (never a benchmark number)
    rehearse = world > 1 and os.environ.get("SMK_BENCH_REHEARSE") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            local = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if world != a.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d: refusing to report a number for the wrong rank count" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)

    pkg = load_package()
    from simian_spacemonkey_amd import sortlast  # noqa: F401
    pkg.sortlast = sortlast
    dev = local if world > 1 else 0
    torch.cuda.set_device(dev)
    r = pkg.Renderer(dev)            # raises when the HIP library / device is missing
    # (profiling only: SMK_BENCH_BRICKS=0 runs every leg with empty-space skipping off -- the streaming kernel's own counters)
    flags_on = os.environ.get("SMK_BENCH_BRICKS", "1") != "0"
    if not flags_on:
        r.set_option("bricks", 0)
    r.set_option("kernel", a.kernel)

    n, size, planes = a.volume, a.size, a.planes
    vghf, nrm = make_volume(r, n)
    if world > 1:
        r.set_shard(rank, world)
    r.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
    xform, mv = configure(r, "cfg3", n, size, planes)

    npix = size * size
    frame = torch.zeros((npix, 4), dtype=torch.float32, device="cuda")
    cstate = None
    if world > 1:
        def march(ptr, stream):
            r.render_device(ptr, None, stream)
            return r.last_frame_id()

        def repair(frame_id, ptr, stream):
            """a frame the slice-ring kernel flagged is rendered again by the gather kernel, on this rank
            alone, before its layer is exchanged"""
            if r.frame_failed(frame_id) != 1:
                return False
            # (the benchmark's pose is static: a host whose camera moves sets frame i's camera again here, before the
            #  re-render -- the merge itself keeps the visibility order taken when the frame was first rendered)
            r.set_option("kernel", 1)
            r.render_device(ptr, None, stream)
            torch.cuda.synchronize()
            r.set_option("kernel", a.kernel)
            return True

        def over(layers, order_, out_tile, stream):
            r.composite_over_device(layers.data_ptr(), world, order_, layers.shape[1], out_tile.data_ptr(), stream)
        # the merge: behind the C ABI (smk_exchange_*: grouped ncclSend/ncclRecv direct send + ordered over +
        # gather, csrc/smk_exchange.hip) wherever every rank has its own GPU; the rehearsal on one GPU
        # (RCCL does not run two ranks on one device) goes through torch.distributed/gloo instead
        exchange_mode, xchg = "torch.distributed all_to_all_single + gather (gloo, rehearsal)", None
        if not rehearse and os.environ.get("SMK_BENCH_EXCHANGE") == "torch":   # (operator's switch: skip the C exchange)
            exchange_mode = "torch.distributed all_to_all_single + gather over RCCL (SMK_BENCH_EXCHANGE=torch)"
        elif not rehearse:
            ok = 1
            try:
                uid = [pkg.exchange_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                xchg = pkg.Exchange(r, rank, world, npix, id=uid[0])
            except Exception as e:  # noqa: BLE001  (reported on the JSON line, never silent)
                ok, why = 0, str(e)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                exchange_mode = "smk_exchange (C ABI): RCCL grouped ncclSend/ncclRecv direct send + ordered over + gather"
            else:
                if xchg is not None:
                    xchg.close()
                xchg = None
                exchange_mode = "torch.distributed all_to_all_single + gather over RCCL (smk_exchange unavailable on some rank%s)" % (
                    ": " + why if not ok else "")
        if xchg is not None:
            cstate = (sortlast.ExchangePipeline(march, xchg, npix, rank, frame_check=repair), r.shard_order(world))
        else:
            cstate = (sortlast.Pipeline(march, over, npix, via_host=rehearse, frame_check=repair), r.shard_order(world))
        if rank != 0:
            del vghf, nrm
            vghf = nrm = None

    # everything (render kernel, RCCL exchange, composite) is enqueued on ONE non-default torch
    # stream so its handle can be passed through the C ABI and the HIP events bracket the kernel
    # on the stream it really runs on
    work = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(work):
        t, kms, kn = timed(r, a.steps, a.warmup, frame, world, cstate)
    kernel, _, alg_bytes = r.last_frame_info()
    ms = t / a.steps * 1e3
    default_workload = (n, size, planes) == (512, 1024, 512)
    samples = float(size) * size * planes
    out = {
        "metric": "Msamples/s + fps, 512^3 VGH vol @1024^2 x 512 steps",
        "value": samples / (t / a.steps) / 1e6, "unit": "Msamples/s", "fps": a.steps / t,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg3: %d^3 f32 VGH + u8 normals, 2-D LevWidget TF, R8k Phong (diff+spec), "
                               "%dx%d viewport x %d planes, pose 30deg about (1,1,0)" % (n, size, size, planes),
                   "volume": n, "viewport": size, "planes": planes,
                   "parallelism": "sort-last x%d (brick shards; RCCL all-to-all + ordered over + gather, two "
                                  "frames in flight: frame i's exchange overlaps frame i+1's ray-marching)" % world
                   if world > 1 else "single GPU",
                   "kernel": {1: "gather", 2: "slab-staged"}.get(kernel, str(kernel)),
                   # what the frame does NOT do, said on the line itself: bricks of 8x8x8 cells in which no sample can be
                   # visible under the table are neither streamed nor sampled.  Every sample that is dropped
                   # is exactly transparent -- the frame is compared bit for bit with the flags-off one in this very run
                   # (`without_brick_flags`, which is also the leg to read for the streaming kernel's own roofline).  The
                   # flags are made per volume upload (value ranges) and per table refresh (two launches, ~0.1 ms of stream
                   # time); volume and table are static here, so neither falls into the timed steps.
                   "empty_space_skipping": "on (option bricks=1, the default)" if flags_on else "off (SMK_BENCH_BRICKS=0)"},
    }
    if rank == 0:
        out["roofline"] = roofline(r, kms, alg_bytes, size,
                                   note="this rank's shard; %d^3 f32 working set is VALU/LDS-bound by construction "
                                        "(BASELINE.md sec. 2), see north_star" % n,
                                   traffic=pmc_traffic("cfg3") if default_workload and world == 1 else None)
        out["roofline"]["kernel_frames_timed"] = kn
        if default_workload and world == 1:
            out["roofline_valu"] = roofline_valu(kms, "cfg3")
        if world == 1:
            out["samples"] = sample_counts(r, frame, size, planes)
        out["workgroups"] = workgroup_load(r)
        if world == 1 and not a.no_extra and flags_on:
            out["without_brick_flags"] = no_flags_leg(r, work, frame, size, planes, a.steps)
    failures = int(r.stat("slab_failures"))
    if world > 1:
        # every rank's own figures on the one line: kernel time of its shard, longest workgroup against the mean slot load,
        # what it waited for the merge; the ranks that really spoke RCCL
        mine = {"rank": rank, "kernel_ms": kms, "kernel": {1: "gather", 2: "slab-staged", 4: "column-stream"}.get(kernel, str(kernel)),
                "workgroups": workgroup_load(r), "exchange_object": bool(cstate is not None and hasattr(cstate[0], "x")),
                "ms_per_step_this_rank": ms}
        per = [None] * world
        dist.all_gather_object(per, mine)
        out["per_rank"] = per
        # ranks whose layers cross the node through RCCL: all of them unless this is the one-GPU rehearsal (gloo through host
        # memory); of those, the ranks whose merge runs behind the C ABI (smk_exchange_*) rather than through torch.distributed
        out["rccl_ranks"] = 0 if rehearse else sum(1 for p_ in per if p_ is not None)
        out["smk_exchange_ranks"] = 0 if rehearse else sum(1 for p_ in per if p_ and p_["exchange_object"])
        out["exchange"] = exchange_mode
        out["exchange_ms_estimate"] = max(0.0, ms - max(p_["kernel_ms"] for p_ in per if p_))   # what a frame takes beyond the slowest rank's ray-march
        out["frames_repaired"] = int(cstate[0].repaired)
        ft = torch.tensor([failures, out["frames_repaired"]], dtype=torch.int64, device="cuda" if not rehearse else "cpu")
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        failures, out["frames_repaired"] = int(ft[0].item()), int(ft[1].item())
    out["slab_failures"] = failures       # slice-ring frames flagged invalid during the run, all ranks (must be 0)
    if rehearse:
        out["data"] = "synthetic; REHEARSAL (all ranks on cuda:0, gloo through host memory): not a benchmark number"
    if world > 1:
        # outside the timed region: the merged frame against the same frame ray-marched unsharded
        dist.barrier()
        if rank == 0:
            torch.cuda.synchronize()
            merged = frame.clone()
            whole = pkg.Renderer(dev)      # a second context holding the unsharded volume
            whole.upload_volume_device(vghf.data_ptr(), (n, n, n), 3, 1, nrm.data_ptr())
            configure(whole, "cfg3", n, size, planes)
            whole.render_device(frame.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            whole.close()
            out["sortlast_check"] = {"max_abs_err_vs_unsharded_frame": float((merged - frame).abs().max().item()),
                                     "alpha_mean": float(merged[:, 3].mean().item()), "tolerance": 2e-5}
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_cpu:
        torch.cuda.synchronize()
        gpu_frame = frame.view(size, size, 4).cpu().numpy()
        tf_eff, _ = r.tf2d_effective(256, 256)
        out["cpu_baseline"] = cpu_baseline(vghf, nrm, tf_eff, size, planes, xform, mv, gpu_frame)
    if rank == 0 and world == 1 and not a.no_north_star:
        del vghf, nrm
        torch.cuda.empty_cache()
        nn_ = a.north_star_volume
        v2, g2 = make_volume(r, nn_)
        r.upload_volume_device(v2.data_ptr(), (nn_, nn_, nn_), 3, 1, g2.data_ptr())
        v2h = g2h = None
        if not a.no_cpu:       # host copies for the checker (13 + 3 GB), before the device tensors go
            v2h, g2h = v2.cpu().numpy(), g2.cpu().numpy()
        del v2, g2
        torch.cuda.empty_cache()
        xform, mv2 = configure(r, "cfg4", nn_, size, planes)
        k2 = max(3, min(a.steps, 10))
        with torch.cuda.stream(work):
            t2, kms2, kn2 = timed(r, k2, 2, frame, 1, None)
        kernel2, _, alg2 = r.last_frame_info()
        rl2 = roofline(r, kms2, alg2, size, traffic=pmc_traffic("north_star") if default_workload and nn_ == 1024 else None)
        # SURVEY 8(d) counts the f32 VGH triple alone (12 B/voxel); the algorithmic bytes above add the 3
        # normal bytes per voxel the Phong term reads
        rl2["frac_on_vgh_bytes_only"] = rl2["frac"] * (12.0 / 15.0)
        out["north_star"] = {
            "workload": "cfg4 single GPU: %d^3 f32 VGH + u8 normals, (v,g)x(h) TF, R8k Phong, %dx%dx%d" % (nn_, size, size, planes),
            "ms_per_frame": t2 / k2 * 1e3, "fps": k2 / t2, "Msamples_per_s": samples / (t2 / k2) / 1e6,
            "roofline": rl2, "kernel": {1: "gather", 2: "slab-staged"}.get(kernel2, str(kernel2))}
        out["north_star"]["samples"] = sample_counts(r, frame, size, planes)
        out["north_star"]["workgroups"] = workgroup_load(r)
        if not a.no_extra and flags_on:
            out["north_star"]["without_brick_flags"] = no_flags_leg(r, work, frame, size, planes, a.steps)
            out["north_star"]["column_stream_kernel"] = column_stream_leg(r, work, frame, size, planes, a.steps)
        if not a.no_cpu:
            # parity of the north-star frame itself: the CPU checker on 200 random rays of it
            torch.cuda.synchronize()
            ns_frame = frame.view(size, size, 4).cpu().numpy()
            tf_eff2, _ = r.tf2d_effective(256, 256)
            out["north_star"]["parity_max_abs_err"] = north_star_parity(v2h, g2h, tf_eff2, size, planes, xform, mv2, ns_frame)
        del v2h, g2h
        if not a.no_extra:
            out["north_star"]["dense_ramp"] = dense_leg(r, work, frame, size, planes, a.steps, "north_star_dense")
            out["extra"] = extra_legs(r, frame, work, a.steps)
    bad = out["slab_failures"] != 0 or r.stat("slab_retries") != 0
    # a run that was meant to exchange over RCCL and did not (on every rank) must not pass for a scaling point
    rccl_short = world > 1 and not rehearse and rank == 0 and out.get("rccl_ranks") != world
    if not flags_on:
        out["data"] += "; SMK_BENCH_BRICKS=0: empty-space skipping off (a profiling run, not the product's default)"
    if rank == 0:
        print(json.dumps(out))
    if world > 1 and cstate is not None and hasattr(cstate[0], "x"):
        cstate[0].x.close()
    r.close()
    if world > 1:
        dist.destroy_process_group()
    if rccl_short:
        print("bench.py: %s of %d ranks exchange through RCCL (smk_exchange_*): %s" % (out.get("rccl_ranks"), world, out.get("exchange")), file=sys.stderr)
        sys.exit(4)
    if bad:   # a frame the slice-ring kernel flagged would make the timing meaningless: fail loudly
        print("bench.py: slice-ring kernel flagged %d frame(s)" % out["slab_failures"], file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
