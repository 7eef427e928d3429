// smk_slab.hip -- kernel S: the slice-ring ray-marcher (the fast path for 2-D / separable
// classification without perturbation or depth output).  DESIGN.md section 4 has the measurements
// behind every choice below.
//
// The gather kernel (smk_gather.hip) pulls 8 corners per sample through TA/L1: every 128-B line
// is re-requested from L2 several times and waves spend ~87 % of their time in s_waitcnt.  Here
// the volume is streamed instead:
//
//   * the principal axis S of the view (largest |ray direction| component in voxel space) is
//     chosen on the host; U is the memory-contiguous axis, V the third one (a lazily built
//     x-major copy of the volume serves S = x);
//   * a workgroup owns a pixel tile: NW consumer waves (one lane = one ray, a wave = a compact
//     8x8 sub-tile) plus NL loader waves;
//   * the loaders stream, front to back, the (u,v) window of every S-slice the tile's ray bundle
//     crosses from HBM straight into an LDS ring with LDS-DMA (global_load_lds_dwordx4, no VGPR
//     round trip): a lean issue loop (scalar address bumps, per-slice lane masks), a counted
//     s_waitcnt vmcnt(N) that retires the oldest slice in flight, a `landed` word per loader;
//   * every consumer wave advances on its own (no workgroup barrier in the main loop): one
//     sample per lane and iteration, taken when the two slices it touches have landed; the 8
//     corners come from LDS in one batch of reads behind one wait; the wave's progress (minimum
//     over its lanes) lets the loaders recycle ring slots;
//   * RGBA stays in registers front to back; 16 B per pixel leave the kernel.
//
// Each voxel row piece a tile needs is read once per tile; neighbouring tiles share the fringe
// (served by the XCD's L2: tiles are dealt to XCDs in contiguous, equal-work runs).
//
// Sample placement, membership and interpolation order are EXACTLY those of the gather kernel
// (same fma chains), so the two kernels agree bit for bit and the CPU checker on positions.
// Reference semantics: see smk_device.h.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <type_traits>

#include "smk_device.h"

// wave-uniform description of one launch
struct SlabParams {
  int perm;                    // 0: S=z (U=x,V=y)  1: S=y (U=x,V=z)  2: S=x (U=y,V=z; x-major copy)
  int au, av, as;              // model-axis index of U, V, S
  long long strideV, strideS;  // voxel strides of the layout in use (U stride is 1)
  int Ou, Ov, Os;              // stored-box origin along U,V,S (global voxel index)
  int Du, Dv, Ds;              // stored-box dims along U,V,S
  int wu;                      // 16-byte units per window row that are loaded at most (<= wp)
  int wv;                      // window rows that are loaded
  int wp;                      // LDS row pitch in 16-byte units, a multiple of 8: the slot image is flat with this
                               // pitch, so the (row, column) a DMA lane serves repeats every `per` chunks = `rpg` rows
  int per, rpg;                // chunks and rows per group: per = wp / gcd(64, wp), rpg = 64 / gcd(64, wp)
  int groups;                  // row groups per slice = ceil(wv / rpg); chunks = groups * per
  int mask_need;               // loaders fetch only what each slice needs of the window (big windows)
  int chunks;                  // DMA wave-instructions per slice = ceil(wv / rows per chunk), uniform
  int slot_bytes;              // chunks * 1024
  int nslots;                  // ring size
  int maxfly;                  // slices a loader keeps in flight ((maxfly-1) * its chunks <= 63)
  int wstep;                   // a wave steps when slices up to its slowest lane's position + 1 + wstep have landed
  int pmask;                   // consumers publish progress when (iteration & pmask) == 0
  int dir;                     // +1: rays advance towards +S, -1: towards -S
  int tw, th;                  // pixel tile
  const void *vox;             // layout base (native or x-major)
  int use_ah;                  // third-axis alpha served from a 1-D LDS table (<= 3 channels)
  int use_occ;                 // (V,G) occupancy bitmap copied to LDS
  int fast_tf;                 // alpha-first classification with 8-byte texel loads (no third axis, or use_ah)
  const unsigned char *bricks;  // brick flags of the stored box (smk_bricks.hip) or null: see "EMPTY LAYERS" in the kernel
  int bsu, bsv, bss;           // their strides along U, V, S (in bricks)
  const int2 *order;           // workgroup of each block: {tile | piece << 20 | pieces << 26, cut fractions lo | hi << 8} (work-balanced
                               // schedule, .x = -1: none), see smk_launch_slab and DEPTH SEGMENTS
  unsigned *tile_ticks;        // [5][ntiles]: duration of each tile's workgroup in 100 MHz ticks (next frame's weights) |
                               // slices its loaders streamed | slices of its range (the loaders stop once every ray of
                               // the tile is saturated: what was NOT streamed is not counted as read, smk_last_frame_info)
  int ntiles;
  int *status;                 // host-visible word: status_tag | (1 = protocol time-out, 2 = window bound violated)
  int status_tag;              // the frame's id << 8: a word written late, into a slot that has been handed on, is told apart by it
  float *diag;                 // [16] diagnostic counters (lockstep bit 16) or null
  unsigned *trace;             // [nblocks][8] per-workgroup timeline record (lockstep bit 32, see smk.h) or null
  float4 *seg_out;             // [maxseg - 1][W * H]: partial frames of the depth segments 1.. of split tiles (DEPTH SEGMENTS), or null
  unsigned *piece_ticks;       // [ntiles][8]: duration of every piece of a split tile (where the next cuts come from)
};

#define SLAB_EPS 0.02f
// cache policy of the LDS-DMA stream ("" = default, " nt" = non-temporal); an experiment knob
#ifndef SLAB_DMA_POLICY
#define SLAB_DMA_POLICY ""
#endif
#define SLAB_DONE 0x3fffffff

template <int DT>
struct VoxT;
template <>
struct VoxT<0> {
  typedef uint2 type;
};
template <>
struct VoxT<1> {
  typedef float4 type;
};

typedef float v2f __attribute__((ext_vector_type(2)));

// The 8 corners of one sample as ONE batch of LDS reads behind ONE wait.  Written as asm: left
// to hipcc the reads are either split into partial ds_read2_b32 pieces per use or (behind an
// optimisation barrier) waited for one by one -- 8 LDS round trips per sample.  a/b = byte
// addresses of corner (u,v) in the two slices, ap/bp = the same one row up; +VB = one voxel on.
// Only the channels classification needs are read here (2 or 3 floats / the 4 data bytes); the
// packed normals follow in a second batch for the samples that turn out to be visible: 16-24
// instead of 32 VGPRs live across the batch, and less LDS traffic for the transparent majority.
typedef float v3f __attribute__((ext_vector_type(3)));
#define SLAB_READ8(INS, OFF)                                                                                              \
  asm volatile(INS " %0, %8\n\t" INS " %1, %8 offset:" OFF "\n\t" INS " %2, %9\n\t" INS " %3, %9 offset:" OFF "\n\t"       \
               INS " %4, %10\n\t" INS " %5, %10 offset:" OFF "\n\t" INS " %6, %11\n\t" INS " %7, %11 offset:" OFF "\n\t"  \
               "s_waitcnt lgkmcnt(0)"                                                                                     \
               : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])   \
               : "v"(a), "v"(ap), "v"(b), "v"(bp)                                                                         \
               : "memory")
// f32 voxels {c0, c1, c2, normal bits}: first two / three channels
__device__ __forceinline__ void slab_read8(unsigned a, unsigned ap, unsigned b, unsigned bp, v2f (&q)[8]) { SLAB_READ8("ds_read_b64", "16"); }
__device__ __forceinline__ void slab_read8(unsigned a, unsigned ap, unsigned b, unsigned bp, v3f (&q)[8]) { SLAB_READ8("ds_read_b96", "16"); }
// whole voxels (normals included) in one batch: workgroups that own a CU alone have the registers
// for it, and can then release their ring slots before classification and shading (EARLY below)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void slab_read8_full(unsigned a, unsigned ap, unsigned b, unsigned bp, v4f (&q)[8]) { SLAB_READ8("ds_read_b128", "16"); }
__device__ __forceinline__ void slab_read8_full(unsigned a, unsigned ap, unsigned b, unsigned bp, v2u (&q)[8]) { SLAB_READ8("ds_read_b64", "8"); }
// u8 voxels {4 data bytes, normal bits}: the data dword
__device__ __forceinline__ void slab_read8_u8(unsigned a, unsigned ap, unsigned b, unsigned bp, uint32_t (&q)[8]) { SLAB_READ8("ds_read_b32", "8"); }
// the packed normals of the same 8 corners (dword NOFF of the voxel)
__device__ __forceinline__ void slab_read8_nb16(unsigned a, unsigned ap, unsigned b, unsigned bp, uint32_t (&q)[8]) {
  a += 12; ap += 12; b += 12; bp += 12;
  SLAB_READ8("ds_read_b32", "16");
}
// the third float channel of the 8 corners (16-byte voxels), read on its own behind the (v, g) occupancy bit
__device__ __forceinline__ void slab_read8_h16(unsigned a, unsigned ap, unsigned b, unsigned bp, uint32_t (&q)[8]) {
  a += 8; ap += 8; b += 8; bp += 8;
  SLAB_READ8("ds_read_b32", "16");
}
__device__ __forceinline__ void slab_read8_nb8(unsigned a, unsigned ap, unsigned b, unsigned bp, uint32_t (&q)[8]) {
  a += 4; ap += 4; b += 4; bp += 4;
  SLAB_READ8("ds_read_b32", "8");
}
#undef SLAB_READ8

// per-slice table entry: where the slice's window sits in the ring and in the volume
struct SlabEnt {
  int base;      // LDS byte address of GLOBAL voxel (u=0, v=0) of this slice's slot image:
                 // corner address = base + v * pitch_bytes + u * voxel_bytes
  unsigned pack;  // loader: window origin u0 | v0 << 11 (stored-box voxels), (units the slice needs - 1) << 22,
                  // rows of the window the slice can spare, in sixteenths of wv, << 28
};

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef __attribute__((address_space(3))) char *lds_cptr_w;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15)
    W(16) W(17) W(18) W(19) W(20) W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31)
    W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40) W(41) W(42) W(43) W(44) W(45) W(46) W(47)
    W(48) W(49) W(50) W(51) W(52) W(53) W(54) W(55) W(56) W(57) W(58) W(59) W(60) W(61) W(62) W(63)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef W
}

__device__ __forceinline__ int lds_ld(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(int *p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDS reads of the LOADER wave go through inline asm: with an LDS-DMA in flight hipcc puts
// s_waitcnt vmcnt(0) in front of every LDS read it can see (it cannot prove the read does not
// alias the DMA destination), which would drain the whole stream once per loop iteration.
// (cdna_hip_programming.md 5.7: the wait for an asm load is ours to place -- it is in the string.)
typedef __attribute__((address_space(3))) const void *lds_cptr_t;
__device__ __forceinline__ int raw_lds_b32(const void *p) {
  int v;
  unsigned a = (unsigned)(size_t)(lds_cptr_t)p;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ uint2 raw_lds_b64(const void *p) {
  uint2 v;
  unsigned a = (unsigned)(size_t)(lds_cptr_t)p;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ void raw_lds_st_b32(void *p, int v) {
  unsigned a = (unsigned)(size_t)(lds_cptr_t)p;
  asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
}

// minimum over the 64 lanes of a fully active wave, on the DPP network (no LDS traffic)
// (the compiler spends 27 instructions on it -- v_mov + s_nop + v_mov_dpp + v_min per step, four v_readlane for the row
//  minima; written out with v_min_i32_dpp and row_bcast it is 13, bit-identical frames, and no faster on either frame)
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));  // row_half_mirror
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));  // row_mirror
  // every row of 16 lanes now holds its own minimum
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// bilinear RGBA8 lookup like smk_tex2d, for tables of >= 2x2 texels: the clamped texel pair is
// always (i, i+1) then, so the four texels are two 8-byte loads at a 32-bit offset
struct SlabTexel4 {
  uint32_t a, b, c, d;
  float fs, ft;
};
__device__ __forceinline__ SlabTexel4 slab_tex2d_fetch(const uint32_t *tex, int ss, int s0, int t0, float fs, float ft) {
  SlabTexel4 o;
  o.fs = fs;
  o.ft = ft;
  const unsigned off = (unsigned)(t0 * ss + s0) * 4u;
  const char *tb = reinterpret_cast<const char *>(tex);
  uint2 lo, hi;
  __builtin_memcpy(&lo, tb + off, 8);
  __builtin_memcpy(&hi, tb + (off + (unsigned)ss * 4u), 8);
  o.a = lo.x;
  o.b = lo.y;
  o.c = hi.x;
  o.d = hi.y;
  return o;
}
__device__ __forceinline__ float slab_tex_chan(const SlabTexel4 &x, int k) {
  return smk_lerp(smk_lerp(smk_ub(x.a, k), smk_ub(x.b, k), x.fs), smk_lerp(smk_ub(x.c, k), smk_ub(x.d, k), x.fs), x.ft) * SMK_INV255;
}

// NW = consumer waves (64 rays each), NL = loader waves; the block has (NW+NL)*64 threads, the
// last NL waves are loaders: loader l streams DMA chunks l, l+NL, ... of every slice.
// second launch-bound = waves per SIMD wanted: workgroups of 5/9/10 waves only double up on a CU
// (2 x 9 waves = 5 on one SIMD) if the kernel stays within 96 VGPRs
// TF: 1 = 2-D (V,G) table x optional third-axis alpha (NV20VolRen3D.cpp:544-596), 2 = dense 3-D (v,g,h)
// table (TFWidgetRen.cpp:779-845; BASELINE configs 4/5)
// (BR: the instance knows about brick flags -- EMPTY LAYERS; frames without flags run instances that carry none of it:
//  the run-time test alone, three per loop turn, cost them 12 % in scalar registers spilled)
// (SHD: the frame's planes are the half-angle slices of a frame with shadows -- SmkShadowRays; the eye pass of smk_shadow.hip.
//  Compile-time: the instances live in smk_slab_shadow.hip, SLAB_PART 2)
template <int DT, int SH, int PERM, int NW, int NL, bool DIAG, int TF = 1, bool BR = true, bool SHD = false>
#ifndef SLAB_BIG_WAVES
#define SLAB_BIG_WAVES 12  // workgroups of more waves than this are "big": one per CU
#endif
// (second argument: waves per SIMD the register allocation must allow -- two small workgroups per CU)
__global__ __launch_bounds__((NW + NL) * 64, ((NW + NL) == 9) ? 6 : ((NW + NL) == 5 || (NW + NL) == 10) ? 5 : ((NW + NL) == 11 ? 3 : ((NW + NL) == 12 && 12 <= SLAB_BIG_WAVES) ? 6 : ((NW + NL) == 14 && 14 <= SLAB_BIG_WAVES) ? 7 : 4)) void smk_k_slab(const RenderParams P, const SlabParams Q) {
  constexpr int UPV = DT == 0 ? 2 : 1;   // voxels per 16-byte DMA unit
  constexpr int VB = DT == 0 ? 8 : 16;   // bytes per voxel
  // (global_load_lds_dwordx3 does NOT compact: it writes 12 bytes per lane at a 16-byte lane stride
  //  -- tools/dma_layout_probe.hip -- so staging only {c0,c1,c2} needs a 12-byte HBM plane)
  constexpr int VBL = DT == 0 ? 3 : 4;   // log2
  constexpr int NTH = (NW + NL) * 64;
  // big workgroups (one per CU, 128 VGPRs each): read whole voxels, release ring slots early
#ifndef SLAB_EARLY
#define SLAB_EARLY 1  // (re-measured with the loaders in pairs: without the early release 4.56 vs 4.11 ms on the 1024^3 frame)
#endif
  constexpr bool BIG = (NW + NL) > SLAB_BIG_WAVES;
  constexpr bool EARLY = BIG && SLAB_EARLY;
  // ... and their loaders skip the row groups a slice does not need, counting DMA instructions per
  // slice; small workgroups keep every slice the same number of instructions (cheaper bookkeeping:
  // measured 3 % on the 512^3 frame, where the loaders' issue slots are the consumers')
#ifndef SLAB_SMALL_FIFO
#define SLAB_SMALL_FIFO 0
#endif
  constexpr bool FIFO = BIG || SLAB_SMALL_FIFO;
  // Small workgroups: the loaders take WHOLE slices in turn (loader l streams slices l, l + NL, ...) instead of a share of
  // the row groups of every slice.  A slice costs a loader ~120 scalar instructions before its first DMA (ring check,
  // table entry, 64-bit source address, column masks); with 3 DMA instructions per loader and slice that overhead was the
  // larger part of the loaders' time, and they were busy 90 % of the frame.  Big workgroups keep the split: their ring is
  // too short for NL slices being filled at once.
#ifndef SLAB_ALT
#define SLAB_ALT 1
#endif
  // Generalised: the loaders form NLG groups; group g streams slices g, g + NLG, ..., and the LPG loaders of a group
  // share the row groups of such a slice.  Small workgroups: NLG = NL (a loader per slice).  Big ones: SLAB_BIG_NLG
  // (1 = every loader works on every slice).
#ifndef SLAB_BIG_NLG
#define SLAB_BIG_NLG 2
#endif
  constexpr int NLG = !SLAB_ALT ? 1 : (BIG ? ((NL % SLAB_BIG_NLG) == 0 ? SLAB_BIG_NLG : 1) : NL);
  constexpr int LPG = NL / NLG;
  constexpr int QSTEP = NLG;
  extern __shared__ __align__(16) unsigned char smem[];
  // LDS carve: ring [nslots][slot_bytes] | slice table [Ds] | control words | alpha_H
  SlabEnt *wtab = reinterpret_cast<SlabEnt *>(smem + (size_t)Q.nslots * Q.slot_bytes);
  // control words: [0] smin [1] smax [3] error flag [4..4+NL) landed per loader [8..8+NW) progress
  int *ctl = reinterpret_cast<int *>(smem + (((size_t)Q.nslots * Q.slot_bytes + (size_t)Q.Ds * sizeof(SlabEnt) + 15) & ~(size_t)15));
  // third-axis alpha as a 1-D table: with <= 3 channels the (H,4th) lookup has t = 0, i.e. row
  // 0 of deptex2 with a zero t-weight, so lerp(row0[s0], row0[s1], fs) is the SAME float
  float *ah = reinterpret_cast<float *>(ctl + 8 + 32);
  // occupancy bitmap of the (V,G) table (smk_api.hip refresh_tf2d), a copy per workgroup
  const uint32_t *occ = reinterpret_cast<const uint32_t *>(ah + (Q.use_ah ? P.sv : 0));

  // order entry: tile | segment << 20 | segments of the tile << 26 (DEPTH SEGMENTS below); -1 = none
  const int2 oent = Q.order[blockIdx.x];
  const int ocode = oent.x;
  if (ocode < 0) return;  // whole workgroup leaves together
  const int tile = ocode & 0xfffff, seg = (ocode >> 20) & 63, nseg = max((ocode >> 26) & 31, 1);
  const int cut_lo = oent.y & 255, cut_hi = (oent.y >> 8) & 255;  // this piece's share of the tile's slice positions, in 255ths
  const bool flags = BR && Q.bricks != nullptr;
  const bool tracing = DIAG && Q.trace != nullptr && (P.lockstep & 32);  // (diagnostic: workgroup timeline)
  // the workgroup's duration feeds the next frame's schedule (see smk_launch_slab): one scalar
  // timestamp at each end and one 4-byte store per tile
  const unsigned trace_t0 = (unsigned)__builtin_amdgcn_s_memrealtime();
  const int ty = tile / P.ntx, tx = tile - ty * P.ntx;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: roles and loops stay wave-uniform
  const bool is_loader = wave >= NW;
  const int lid = wave - NW;  // loader index
  // consumer wave = 8x8 pixel sub-tile; waves laid out row-major over the tile
  const int wpr = Q.tw >> 3;
  const int i = tx * Q.tw + (wave % wpr) * 8 + (lane & 7);
  const int j = ty * Q.th + (wave / wpr) * 8 + (lane >> 3);
  const bool live = !is_loader && i < P.W && j < P.H;

  const smk_raycoef &rc = P.rc;
  const float px = __fmaf_rn((float)i + 0.5f, rc.pxs, rc.pxl);
  const float py = __fmaf_rn((float)j + 0.5f, rc.pys, rc.pyl);
  float A[3], B[3], tauA, dtau;  // (tauA, dtau: frames with shadows only -- the ray parameter of plane q is fma(q, dtau, tauA), smk_ray_AB)
  const bool ray_ok = smk_ray_AB_t<SHD>(P, px, py, A, B, tauA, dtau);
  // conservative plane range (identical to the gather kernel)
  float tenter = 0.0f, texit = (float)(rc.nplanes - 1);
  bool empty = rc.nplanes <= 0 || !live || !ray_ok;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (fabsf(B[a]) > 1e-20f) {
      float inv = 1.0f / B[a];
      // (the box is widened by SMK_RANGE_EPS voxels: a ray that runs ALONG a face -- the centre row of an odd
      //  viewport along a shard boundary -- has fma(m, B, A) round onto the face for every m although
      //  (lo - A) / B says it leaves at m = 0; the exact per-sample test decides, this only brackets it)
      float t1 = (P.lo[a] - SMK_RANGE_EPS - A[a]) * inv, t2 = (P.hi[a] + SMK_RANGE_EPS - A[a]) * inv;
      tenter = fmaxf(tenter, fminf(t1, t2) - 2.0f);
      texit = fminf(texit, fmaxf(t1, t2) + 2.0f);
    } else if (!(A[a] >= P.lo[a] && A[a] <= P.hi[a])) {
      empty = true;
    }
  }
  // Free clip plane (glClipPlane, NV20VolRen3D.cpp:346-357): fragments on its negative side do not exist.  The plane's value
  // along a ray, cplane . (p(q), 1), is monotone in the plane index q like the coordinates: the kept samples are an interval
  // that ends where it crosses zero -- folded into the ray's range HERE, the marching loop never hears of it (as a test per
  // sample in the loop it cost every frame without a clip plane 4-5 %: round 2).
  if (P.cplane_on) {
    const float c0 = __fmaf_rn(A[0], P.cplane[0], __fmaf_rn(A[1], P.cplane[1], __fmaf_rn(A[2], P.cplane[2], P.cplane[3])));
    const float c1 = __fmaf_rn(B[0], P.cplane[0], __fmaf_rn(B[1], P.cplane[1], B[2] * P.cplane[2]));
    if (fabsf(c1) > 1e-20f) {
      const float tz = -c0 / c1;  // the crossing, in planes
      if (c1 > 0.0f) tenter = fmaxf(tenter, tz - 2.0f);
      else texit = fminf(texit, tz + 2.0f);
    } else if (!(c0 >= 0.0f)) {
      empty = true;
    }
  }
  // Frames with shadows: a sample exists where its ray parameter fma(q, dtau, tauA) is positive -- monotone in q like the clip
  // plane's value, folded into the range the same way.
  if (SHD && ray_ok) {
    if (fabsf(dtau) > 1e-30f) {
      const float tz = -tauA / dtau;
      if (dtau > 0.0f) tenter = fmaxf(tenter, tz - 2.0f);
      else texit = fminf(texit, tz + 2.0f);
    } else if (!(tauA > 0.0f)) {
      empty = true;
    }
  }
  int m = (int)floorf(fmaxf(tenter, 0.0f));
  int m1 = (int)ceilf(fminf(texit, (float)(rc.nplanes - 1)));
  if (empty || !(tenter <= texit)) m1 = m - 1;
  // Exact first/last inside sample.  A coordinate fma(q, B, A) is monotone in q (one correctly
  // rounded operation), so the samples that pass the membership predicate of the gather kernel
  // -- lo <= p <= hin on every axis, hin = hi itself on a top face, else the float just below
  // it -- form ONE interval of q; the conservative range above brackets it with a few planes of
  // slack, so testing its ends here removes the per-sample test from the marching loop.
#ifndef SLAB_NO_INTERVAL
  {
    auto inside = [&](int q) -> bool {
      const float qf = (float)q;
      const float p0 = __fmaf_rn(qf, B[0], A[0]), p1 = __fmaf_rn(qf, B[1], A[1]), p2 = __fmaf_rn(qf, B[2], A[2]);
      bool in = ((int)(smk_clampf(p0, P.lo[0], P.hin[0]) == p0) & (int)(smk_clampf(p1, P.lo[1], P.hin[1]) == p1) &
                 (int)(smk_clampf(p2, P.lo[2], P.hin[2]) == p2)) != 0;
      // (the gather kernel's own fma chain for the plane: the same samples pass, bit for bit)
      if (P.cplane_on) in = in && __fmaf_rn(p0, P.cplane[0], __fmaf_rn(p1, P.cplane[1], __fmaf_rn(p2, P.cplane[2], P.cplane[3]))) >= 0.0f;
      if (SHD) in = in && smk_tau_ok(tauA, dtau, q);
      return in;
    };
    int mf = m1 + 1, ml = m - 1;
    int qa = m, qb = m1;
    // (the ends move inwards until they are inside; rays that only graze the region end empty)
    while (true) {
      const bool go_a = qa <= m1 && mf > m1, go_b = qb >= m && ml < m;
      if (!__any(go_a || go_b)) break;  // (a few steps: the slack is +-2 planes)
      if (go_a) {
        if (inside(qa)) mf = qa;
        ++qa;
      }
      if (go_b) {
        if (inside(qb)) ml = qb;
        --qb;
      }
    }
    m = mf;  // (no inside sample at all: mf = m1 + 1 > ml, the ray is empty)
    m1 = ml;
  }
#endif

  constexpr int AS = PERM == 0 ? 2 : (PERM == 1 ? 1 : 0);
  constexpr int AU = PERM == 2 ? 1 : 0;
  constexpr int AV = PERM == 0 ? 1 : 2;
  const int NS = P.N[AS], NU = P.N[AU], NV = P.N[AV];

  // base slice index of plane q on this ray (a sample reads slices i0 and i0+1)
  auto base_slice = [&](int q) -> int {
    float s = __fmaf_rn((float)q, B[AS], A[AS]);
    float sc = smk_clampf(s, 0.0f, (float)(NS - 1));
    return min((int)sc, NS - 2);
  };
  // the same, also handing out the clamped coordinate: exactly smk_lin_clamp's (xc, i0) of the principal axis, which the
  // sample of plane q needs again one loop turn later (carried instead of recomputed: 4 VALU per turn; within noise on both frames)
  auto base_slice_c = [&](int q, float &sc_out) -> int {
    float s = __fmaf_rn((float)q, B[AS], A[AS]);
    sc_out = smk_clampf(s, 0.0f, (float)(NS - 1));
    return min((int)sc_out, NS - 2);
  };
  float car_sc = 0.f;  // clamped principal-axis coordinate and base slice of THIS ray's next sample (plane m)
  int car_i = 0;

  // ---- workgroup slice range
  if (tid == 0) {
    ctl[0] = 0x7fffffff;
    ctl[1] = -0x7fffffff;
    ctl[2] = 0;  // slices the loaders did not have to stream (EMPTY LAYERS)
    ctl[3] = 0;  // error flag (bounded spins, window bound)
    for (int l = 0; l < 4; ++l) ctl[4 + l] = l < NL ? l % NLG : 0x7fffffff;  // (a loader's first slice) absent loaders never hold anyone back
  }
  if (tid < 32) ctl[8 + tid] = (NL > 4 && tid >= 16 && tid < 12 + NL) ? (tid - 12) % NLG : SLAB_DONE;  // (ctl[24..27]: loaders 4..7)
  __syncthreads();
  {
    int lo = 0x7fffffff, hi = -0x7fffffff;
    if (m <= m1) {
      int a0 = base_slice(m), a1 = base_slice(m1);
      lo = min(a0, a1);
      hi = max(a0, a1);
    }
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, __shfl_xor(lo, o));
      hi = max(hi, __shfl_xor(hi, o));
    }
    if (lane == 0 && lo <= hi) {
      atomicMin(&ctl[0], lo);
      atomicMax(&ctl[1], hi);
    }
  }
  __syncthreads();
  int smin = ctl[0], smax = ctl[1];
  // ---- DEPTH SEGMENTS.  A tile whose workgroup would run long (measured, see the launcher) is rendered by `nseg`
  // workgroups: the positions of its slice range are cut into nseg runs (where, the launcher decides from the pieces'
  // measured durations: equal WORK, not equal depth), workgroup `seg` takes the samples whose base
  // slice lies in its run -- base slices are monotone in the plane index, so a ray's share is a sub-interval of its
  // planes, found by an estimate and the exact evaluation -- and writes a partial frame; the partial frames are merged
  // in marching order afterwards (smk_k_slab_merge: front-to-back over, or max).  What changes is the association of
  // the blend, nothing else: the frame agrees with the unsplit one to a few ulp per segment.
  if (nseg > 1 && smin <= smax) {
    const int np_full = smax - smin + 1;
    const int q0 = (int)((long long)cut_lo * np_full / 255), q1 = (int)((long long)cut_hi * np_full / 255);  // positions [q0, q1)
    // base slices of this segment
    const int b_lo = Q.dir > 0 ? smin + q0 : smax - q1 + 1, b_hi = Q.dir > 0 ? smin + q1 - 1 : smax - q0;
    if (m <= m1) {
      if (q1 <= q0) {
        m1 = m - 1;
      } else {
        const float inv = 1.0f / B[AS];
        // first plane whose base slice is inside [b_lo, b_hi] in marching order: s reaches the run's near face
        const float s_near = Q.dir > 0 ? (float)b_lo : (float)(b_hi + 1), s_far = Q.dir > 0 ? (float)(b_hi + 1) : (float)b_lo;
        auto in_run = [&](int q) -> bool { const int b = base_slice(q); return b >= b_lo && b <= b_hi; };
        auto before = [&](int q) -> bool { const int b = base_slice(q); return Q.dir > 0 ? b < b_lo : b > b_hi; };
        // (at an end of the volume the clamped base slice takes in everything beyond: no estimate, the ray's own end)
        const bool open_near = Q.dir > 0 ? b_lo <= 0 : b_hi >= NS - 2, open_far = Q.dir > 0 ? b_hi >= NS - 2 : b_lo <= 0;
        int qa = open_near ? m : max(m, min(m1 + 1, (int)floorf((s_near - A[AS]) * inv) - 1));
        int qb = open_far ? m1 : min(m1, max(m - 1, (int)ceilf((s_far - A[AS]) * inv) + 1));
        bool bad = false;
#pragma unroll 1
        for (int k = 0; qa <= m1 && before(qa); ++k) { ++qa; if (k > 8) { bad = true; break; } }   // up to the run
#pragma unroll 1
        for (int k = 0; qa > m && !before(qa - 1); ++k) { --qa; if (k > 8) { bad = true; break; } }  // (never started inside it)
#pragma unroll 1
        for (int k = 0; qb >= qa && !in_run(qb) && !before(qb); ++k) { --qb; if (k > 8) { bad = true; break; } }  // back into the run
#pragma unroll 1
        for (int k = 0; qb < m1 && (in_run(qb + 1) || before(qb + 1)); ++k) { ++qb; if (k > 8) { bad = true; break; } }
        if (bad) ctl[3] = 1;  // the bracket did not close: reported, the frame is rendered again another way
        m = qa;
        m1 = (qb >= qa && in_run(qb) && in_run(qa)) ? qb : qa - 1;
      }
    }
    smin = b_lo;
    smax = b_hi;
    if (q1 <= q0) { smin = 0x7fffffff; smax = -0x7fffffff; }
  }
  const bool phases = tracing && (P.lockstep & 128);  // (diagnostic: where the set-up's time goes, instead of the loader's cycles)
  unsigned ph1 = 0, ph2 = 0, ph3 = 0;
  if (phases) ph1 = (unsigned)__builtin_amdgcn_s_memrealtime() - trace_t0;
  const int dir = Q.dir, nslots = Q.nslots;
  const unsigned pitch_b = 16u * (unsigned)Q.wp;  // LDS row pitch in bytes
  const unsigned ring_addr = (unsigned)(size_t)(lds_cptr_t)smem;
  // positions p = 0..npos-1 in marching order: base slice b(p) = dir>0 ? smin+p : smax-p;
  // load order q = 0..npos: slice L(q) = dir>0 ? smin+q : smax+1-q; position p reads L(p), L(p+1)
  const int npos = smax - smin + 1;
  // ---- per-slice windows of this tile (every thread fills some table entries).  Every window
  // has the SAME shape (Q.wu units x Q.wv rows, host-sized to cover the widest bundle section)
  // so a DMA lane's source offset inside the window never changes; only its origin moves: the
  // bbox, over the tile's 4 corner rays, of every position a sample touching slice sl can have
  // (s in [sl-1, sl+1], stretched to the volume faces at the ends)
  if (npos > 0) {
    float cA[4][3], cB[4][3];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int ci = min(tx * Q.tw + ((c & 1) ? Q.tw - 1 : 0), P.W - 1);
      int cj = min(ty * Q.th + ((c & 2) ? Q.th - 1 : 0), P.H - 1);
      float cx = __fmaf_rn((float)ci + 0.5f, rc.pxs, rc.pxl), cy = __fmaf_rn((float)cj + 0.5f, rc.pys, rc.pyl);
      float cta, cdt;
      (void)smk_ray_AB_t<SHD>(P, cx, cy, cA[c], cB[c], cta, cdt);  // (the launcher declines frames whose rays can run parallel to the slices)
    }
    if (Q.use_ah)
      for (int e = tid; e < P.sv; e += NTH) ah[e] = smk_ub(P.tf_h[e], 3);
    if (Q.use_occ) {
      uint32_t *occ_w = const_cast<uint32_t *>(occ);
      for (int e = tid; e < P.occ_roww * (TF == 2 ? P.s3g : P.sg); e += NTH) occ_w[e] = P.tf_occ[e];
    }
    const int wuv = Q.wu * UPV;  // window width in voxels
    for (int q = tid; q <= npos; q += NTH) {
      int sl = dir > 0 ? smin + q : smax + 1 - q;  // global slice index
      int e = sl - Q.Os;
      if (e < 0 || e >= Q.Ds) continue;
      float s_lo = sl <= 1 ? -0.5f : (float)(sl - 1), s_hi = sl >= NS - 2 ? (float)NS - 0.5f : (float)(sl + 1);
      float umin = 1e30f, umax = -1e30f, vmin = 1e30f, vmax = -1e30f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float ib = 1.0f / cB[c][AS];
        float ma = (s_lo - cA[c][AS]) * ib, mb = (s_hi - cA[c][AS]) * ib;
        float ua = __fmaf_rn(ma, cB[c][AU], cA[c][AU]), ub = __fmaf_rn(mb, cB[c][AU], cA[c][AU]);
        float va = __fmaf_rn(ma, cB[c][AV], cA[c][AV]), vb = __fmaf_rn(mb, cB[c][AV], cA[c][AV]);
        umin = fminf(umin, fminf(ua, ub));
        umax = fmaxf(umax, fmaxf(ua, ub));
        vmin = fminf(vmin, fminf(va, vb));
        vmax = fmaxf(vmax, fmaxf(va, vb));
      }
      // texel pair of coordinate x is floor(clamp(x)), +1; SLAB_EPS absorbs fp differences
      // between this bbox and the per-sample chains
      int u0 = (int)floorf(fminf(fmaxf(umin - SLAB_EPS, 0.0f), (float)(NU - 2)));
      int u1 = (int)floorf(fminf(fmaxf(umax + SLAB_EPS, 0.0f), (float)(NU - 2))) + 1;
      int v0 = (int)floorf(fminf(fmaxf(vmin - SLAB_EPS, 0.0f), (float)(NV - 2)));
      int v1 = (int)floorf(fminf(fmaxf(vmax + SLAB_EPS, 0.0f), (float)(NV - 2))) + 1;
      // to stored-box coordinates, clipped to it
      u0 = max(u0 - Q.Ou, 0);
      v0 = max(v0 - Q.Ov, 0);
      u1 = min(u1 - Q.Ou, Q.Du - 1);
      v1 = min(v1 - Q.Ov, Q.Dv - 1);
      if (UPV == 2) u0 &= ~1;  // rows start on whole 16-byte units (the stored U extent is even)
      // fixed-shape window: slide it back inside the stored box where it would stick out
      const int wu0 = min(u0, Q.Du - wuv), wv0 = min(v0, Q.Dv - Q.wv);
      if (u1 - wu0 + 1 > wuv || v1 - wv0 + 1 > Q.wv) ctl[3] = 2;  // host bound violated: reported, never silent
      // what this slice really needs of the fixed-shape window (the loader masks the rest)
      const int need_u = max((u1 - wu0 + UPV) / UPV, 1), need_v = max(v1 - wv0 + 1, 1);
      SlabEnt ent;
      ent.pack = (unsigned)wu0 | ((unsigned)wv0 << 11) | ((unsigned)(min(need_u, Q.wu) - 1) << 22) |
                 ((unsigned)min(15, max(Q.wv - need_v, 0) / ((Q.wv + 15) / 16)) << 28);  // rows spared, in 1/16ths of wv (rounded down)
      ent.base = (int)ring_addr + (q % nslots) * Q.slot_bytes - (Q.Ov + wv0) * (int)pitch_b - (Q.Ou + wu0) * VB;
      wtab[e] = ent;
    }
  }
  if (phases) {
    __syncthreads();
    ph2 = (unsigned)__builtin_amdgcn_s_memrealtime() - trace_t0;
  }
  // ---- EMPTY LAYERS.  The cells between slices b and b + 1 ("layer b") that this tile's rays can cross lie in the
  // overlap of the two slices' windows.  When every brick that overlap touches is flagged empty (smk_bricks.hip: no
  // sample in it can be visible under the current table), nobody in the tile samples layer b -- bit 0 of the entry --
  // and a slice whose two neighbouring layers are both empty is not streamed at all -- bit 1.  (The entries' base
  // addresses are multiples of 8.)  A sample that IS taken therefore finds both its slices loaded: its layer's bit 0 is
  // clear, which keeps bit 1 of both slices clear.  An entry whose slice is not streamed needs no address: it holds,
  // above the two bits, how many empty layers follow one another from this one on in marching order (to the end of
  // the tile's range = "the rest"); rays and loaders step over such a run at once (see there).
  // Three steps, a barrier between them; the flags are read from memory once per layer of BRICKS, a lane per brick:
  // (the first version read them per slice and thread, one after the other: 40 us per workgroup, a fifth of the frame)
  if (flags && npos > 0) {
    constexpr int BL = SMK_BRICK_LOG2;
    __syncthreads();
    auto extent = [&](int e, int &ulo, int &uhi, int &vlo, int &vhi) {
      const unsigned pk = wtab[e].pack;
      ulo = (int)(pk & 0x7ffu);
      vlo = (int)((pk >> 11) & 0x7ffu);
      uhi = ulo + (int)(((pk >> 22) & 0x3fu) + 1u) * UPV - 1;
      vhi = vlo + (Q.wv - (int)(pk >> 28) * ((Q.wv + 15) / 16)) - 1;
    };
    // (1) per layer of bricks along S: origin (first brick the tile's windows touch there) and the flags of the 8 x 8
    // bricks from it on, as a 64-bit mask -- kept in the ring's memory, which nobody uses before the last barrier
    uint4 *lmask = reinterpret_cast<uint4 *>(smem);
    const int nlay = ((Q.Ds - 1) >> BL) + 1;
    // (eight layers per wave and round, so that a workgroup's loads are all in flight at once: lane 8 k + j reads the
    //  window origin of slice j of layer k, the eight lanes of a group reduce to the layer's origin; two rounds of four
    //  layers with every lane looping over the entries took 9-10 us of the set-up's 19)
    static_assert(SMK_BRICK_LOG2 == 3, "the lane mapping below assumes eight slices per layer of bricks");
    for (int bl0 = wave * 8; bl0 < nlay; bl0 += (NW + NL) * 8) {
      int ulo = 0x7fffffff, vlo = 0x7fffffff;
      {
        const int e = ((bl0 + (lane >> 3)) << BL) + (lane & 7);
        const int sl = e + Q.Os;
        if (e < Q.Ds && sl >= smin && sl <= smax + 1) {
          const unsigned pk = wtab[e].pack;
          ulo = (int)(pk & 0x7ffu);
          vlo = (int)((pk >> 11) & 0x7ffu);
        }
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        ulo = min(ulo, __shfl_xor(ulo, o));
        vlo = min(vlo, __shfl_xor(vlo, o));
      }
      int bu0[8], bv0[8];
      unsigned f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int lu = __builtin_amdgcn_readlane(ulo, k * 8), lv = __builtin_amdgcn_readlane(vlo, k * 8);
        bu0[k] = lu >> BL;
        bv0[k] = lv >> BL;
        f[k] = 0;
        const int bu = bu0[k] + (lane & 7), bv = bv0[k] + (lane >> 3);
        if (bl0 + k < nlay && lu != 0x7fffffff && bu <= ((Q.Du - 1) >> BL) && bv <= ((Q.Dv - 1) >> BL))
          f[k] = Q.bricks[(size_t)(bl0 + k) * Q.bss + (size_t)bv * Q.bsv + (size_t)bu * Q.bsu];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned long long m = __ballot(f[k] != 0);
        if (lane == 0 && bl0 + k < nlay) lmask[bl0 + k] = make_uint4((unsigned)bu0[k], (unsigned)bv0[k], (unsigned)m, (unsigned)(m >> 32));
      }
    }
    __syncthreads();
    // (2) per slice: is its layer empty for this tile (bit 0), does anybody need the slice (bit 1)
    auto layer_empty = [&](int sl) -> bool {  // layer sl, smin <= sl <= smax
      const int e = sl - Q.Os;
      int ulo, uhi, vlo, vhi;
      extent(e, ulo, uhi, vlo, vhi);
      if (e + 1 < Q.Ds) {  // (slice sl + 1 <= smax + 1 is in the tile's range: its entry is filled)
        int u2, u3, v2, v3;
        extent(e + 1, u2, u3, v2, v3);
        ulo = max(ulo, u2);
        uhi = min(uhi, u3);
        vlo = max(vlo, v2);
        vhi = min(vhi, v3);
      }
      // lower corners of the cells: one less than the voxels at the top end
      uhi = min(uhi, Q.Du - 1) - 1;
      vhi = min(vhi, Q.Dv - 1) - 1;
      if (uhi < ulo || vhi < vlo) return true;
      const uint4 L = lmask[e >> BL];
      const int c0 = (ulo >> BL) - (int)L.x, c1 = (uhi >> BL) - (int)L.x, r0 = (vlo >> BL) - (int)L.y, r1 = (vhi >> BL) - (int)L.y;
      if (c0 < 0 || r0 < 0 || c1 > 7 || r1 > 7) return false;  // (outside the square that was looked at: not known to be empty)
      const unsigned long long cols = (unsigned long long)((0xffu >> (7 - c1)) & (0xffu << c0)) * 0x0101010101010101ull;
      const unsigned long long rows = (~0ull >> (8 * (7 - r1))) & (~0ull << (8 * r0));
      const unsigned long long m = (unsigned long long)L.z | ((unsigned long long)L.w << 32);
      return (m & cols & rows) == 0;
    };
    for (int q = tid; q <= npos; q += NTH) {
      const int sl = dir > 0 ? smin + q : smax + 1 - q;
      const int e = sl - Q.Os;
      if (e < 0 || e >= Q.Ds) continue;
      const bool mine = sl <= smax && !layer_empty(sl);
      const bool below = sl - 1 >= smin && e - 1 >= 0 && !layer_empty(sl - 1);
      int fl = mine ? 0 : 1;
      if (!mine && !below) fl |= 2;
      if (fl) wtab[e].base |= fl;
    }
    __syncthreads();
    // (3) run lengths, by one wave: 64 positions at a time from the far end, a ballot each (the run that starts at a
    // position = the trailing ones of the mask from its bit on, + the next block's first run when it reaches the end)
    if (wave == 0) {
      int carry = 0x40000;  // behind the last position: "the rest"
      for (int blk = (npos - 1) >> 6; blk >= 0; --blk) {
        const int pq = (blk << 6) + lane;                  // position in marching order
        const int sl = dir > 0 ? smin + pq : smax - pq;    // its layer
        const int e = sl - Q.Os;
        int bits = 3;                                      // (behind the range: counts as empty)
        if (pq < npos) bits = (e >= 0 && e < Q.Ds) ? (wtab[e].base & 3) : 0;
        const unsigned long long m = __ballot((bits & 1) != 0);
        const unsigned long long rest = ~(m >> lane);      // (the shift brings in zeros: the count stops at the block's end)
        int run = rest ? (int)__builtin_ctzll(rest) : 64;
        if (run == 64 - lane) run += carry;
        if (pq < npos && bits == 3) wtab[e].base = (min(run, 0x7ffff) << 2) | 3;
        carry = __builtin_amdgcn_readlane(run, 0);
      }
      if (lane == 0) {  // the slice behind the last layer (no layer of its own)
        const int e = smax + 1 - Q.Os;
        if (e >= 0 && e < Q.Ds && (wtab[e].base & 3) == 3) wtab[e].base = (1 << 2) | 3;
      }
    }
  }
  // every consumer wave announces the first position it needs before anyone moves on
  const int psgn = dir > 0 ? 1 : -1, poff = dir > 0 ? -smin : smax;  // position of base slice b = psgn*b + poff
  int pb = SLAB_DONE;  // position of this ray's next sample
  if (m <= m1) {
    car_i = base_slice_c(m, car_sc);
    pb = psgn * car_i + poff;
  }
  int pos = SLAB_DONE;
  if (!is_loader && npos > 0) {
    pos = wave_min_i32(pb);
    if (lane == 0) ctl[8 + wave] = pos;
  }
  __syncthreads();  // table, alpha_H, control words visible; LAST workgroup barrier
  if (phases) ph3 = (unsigned)__builtin_amdgcn_s_memrealtime() - trace_t0;
  float C0 = 0.f, C1 = 0.f, C2 = 0.f, C3 = 0.f;

  if (npos > 0) {
    if (is_loader) {
      // ================================ loader wave ============================================
      // Streams load indices q = 0..npos in order.  Slot of q is q % nslots; it may be rewritten
      // once every consumer is past position q - nslots (positions < min progress are done).
      // Every slice is exactly `chunks` DMA wave-instructions, so the in-order vmcnt tells which
      // slices have landed.  The window image is flat on a fixed pitch, so the (row, column) a
      // lane serves -- its source offset inside the window -- is the same in every group of rows
      // and every slice, and a chunk costs the wave a few scalar instructions, two compares (the
      // slice's own extent masks the lanes it does not need) + one global_load_lds: measured with
      // tools/dma_probe.hip,
      // ONE such wave per CU streams 5.2-5.9 TB/s chip-wide (94-104 cycles per KiB), while a
      // compiler-scheduled loop with per-lane address arithmetic stays at ~415 cycles per KiB
      // whatever the memory behind it -- instruction issue, not HBM, is what a loader must save.
      __builtin_amdgcn_s_setprio(3);  // the stream must never wait for issue slots behind pollers
      const char *gv = reinterpret_cast<const char *>(Q.vox);
      // row groups g = lid, lid+NL, ... of every slice are mine; a group is `per` chunks = `rpg` rows
      const int per = Q.per, rpg = Q.rpg, groups = Q.groups;
      const int sg = lid % NLG;          // my slices: sg, sg + NLG, ...
      const int gl = lid / NLG, gn = LPG;  // my first row group of such a slice, and the stride to my next one
      const int mygroups = (groups - gl + gn - 1) / gn;
      const int mych = mygroups * per;  // DMA wave-instructions of a whole window (this loader's share)
      const unsigned strideVb = (unsigned)(Q.strideV * (long long)VB);  // bytes, < 2^32
      // unit 64*k + lane of a group sits at (row, column) = divmod(64*k + lane, wp): fixed per lane and phase k
      unsigned voff[7], rowk[7], colk[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const unsigned g = 64u * k + lane;
        rowk[k] = g / (unsigned)Q.wp;
        colk[k] = g - rowk[k] * (unsigned)Q.wp;
        voff[k] = rowk[k] * strideVb + colk[k] * 16u;
      }
      const size_t gstep = (size_t)(gn * rpg) * strideVb;  // source advance from one of my groups to the next
      const size_t strideSb = (size_t)Q.strideS * VB;
      const bool l2hot = DIAG && (P.lockstep & 8) != 0;           // (diagnostic: every slice re-reads one slice)
      // (ALT: q runs over MY slices lid, lid + NL, ...; `landed` stays the published word: every slice below it that is
      //  mine has landed, so the minimum over the loaders' words is the complete prefix as before)
      int q = sg, inflight = 0, landed = sg, idle = 0, minp = 0, slot_q = sg % nslots, fly_total = 0;
      int fly_counts = 0;  // lane (q & 63): DMA wave-instructions of load index q (a scalar array in one VGPR)
      if (DIAG && (P.lockstep & 64)) minp = 0x3ffffff0;  // (diagnostic: free-running stream, nobody consumes)
      // progress words, read by lane 0 alone (NW <= 16 words as four b128 reads)
      auto poll_progress = [&]() -> int {
        int v = SLAB_DONE;
        if (lane < 16) v = raw_lds_b32(&ctl[8 + lane]);  // words >= NW stay at SLAB_DONE
        v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
        v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
        v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
        v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
        return __builtin_amdgcn_readlane(v, 0);
      };
      // one LDS-DMA wave-instruction: 16 B per active lane from src + voff to LDS dst + lane*16
      // (saddr form: no per-chunk VALU; M0 written in the statement that reads it)
      unsigned keep_m0;
#define SLAB_DMA(src_, dst_, voff_)                                                                                 \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3" SLAB_DMA_POLICY "\n\ts_mov_b32 m0, %0" \
               : "=&s"(keep_m0)                                                                                     \
               : "v"(voff_), "s"(dst_), "s"(src_)                                                                   \
               : "memory")
      // table entries of 64 consecutive load indices, one per lane, refreshed every 64 slices:
      // a slice's window origin is then one v_readlane away instead of an LDS round trip
      // (Measured and dropped in round 2: the 64-bit source offset as two more lane arrays, and the
      //  column masks cached across slices in a VGPR bit set -- 1024^3 3.54 -> 3.74 ms: the scalar
      //  per-slice arithmetic below is cheaper than it looks, the extra VALU is not.)
      int ent_uv = 0, tab_base = -64;
      int ent_run = 0;  // ... and, for a slice nobody needs, the length of the run of empty layers it starts (EMPTY LAYERS)
      int n_skipped = 0;
      // EAGER PUBLICATION.  A slice used to be published when its loader, done issuing the NEXT one, waited
      // for it -- up to a slice's issue time (~0.5 us) after it had landed, on a ring that is a few slices
      // deep.  The wave's outstanding vector-memory count can be READ (s_getreg HW_REG_IB_STS: VM_CNT in
      // bits [3:0], its two high bits in [23:22]; tools/vmcnt_probe.hip), and the DMAs complete in order:
      // after every group of chunks the loader looks, and publishes each in-flight slice whose last
      // instruction is no longer outstanding.  `cur` = instructions of the slice being issued so far
      // (younger than everything in flight).  The blocking wait stays for when nothing can be issued.
#ifndef SLAB_EAGER
#define SLAB_EAGER 1
#endif
      // (big workgroups only: on the small ones' deep ring a late word costs little, and the look costs
      //  the kernel scalar registers -- 97 spilled SGPRs against 39)
      constexpr bool EAGER = SLAB_EAGER && BIG;
      auto publish_landed = [&](int cur) {
        if (!EAGER || inflight == 0) return;
        const unsigned st = __builtin_amdgcn_s_getreg((31 << 11) | 7);
        const int out = (int)((st & 0xfu) | (((st >> 22) & 3u) << 4));
        const int before = landed;
        while (inflight > 0) {
          const int c_old = FIFO ? __builtin_amdgcn_readlane(fly_counts, (q - inflight * QSTEP) & 63) : mych;
          const int younger = (FIFO ? fly_total : mych * inflight) - c_old + cur;
          if (out > younger) break;
          if (FIFO) fly_total -= c_old;
          --inflight;
          landed += QSTEP;
        }
        if (landed != before) raw_lds_st_b32(&ctl[(lid < 4 ? 4 : 20) + lid], landed);
      };
      const bool prof = DIAG && (P.lockstep & 48) != 0 && Q.diag != nullptr;  // (diagnostic: where a loader's cycles go)
      long long t_issue = 0, t_wait = 0, t_idle = 0, t_mark = 0;
      const long long t_start = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
      while (landed <= npos) {
        // ---- issue while the ring has room (progress is re-polled only when it blocks us)
        bool stop = false;
        if (prof) t_mark = __builtin_amdgcn_s_memtime();
        while (q <= npos && inflight < Q.maxfly) {
          if ((q & ~63) != tab_base) {  // (q steps by QSTEP: entering a new block of 64 load indices, not hitting its first one)
            tab_base = q & ~63;
            const int ql = tab_base + lane;
            const int e = (dir > 0 ? smin + ql : smax + 1 - ql) - Q.Os;
            ent_uv = -1;
            ent_run = 0;
            if (ql <= npos && e >= 0 && e < Q.Ds) {
              const auto ent = raw_lds_b64(&wtab[e]);
              ent_uv = (int)ent.y;  // (.pack; never -1: u0 < 2^11)
              if (((int)ent.x & 3) == 3) ent_run = (int)ent.x >> 2;
            }
          }
          // EMPTY LAYERS.  A slice nobody samples is not streamed, and a whole run of them is stepped over at once: the
          // entry of an unneeded slice holds how many empty layers follow one another from its own on, and a slice between
          // two empty layers is unneeded.  Nothing is written, so no ring room is asked for.  Slices in flight are waited
          // for first when the run is long (their words are published in order); a short run behind slices in flight
          // goes slice by slice below.
          // (slice q + k lies between the layers of positions q + k - 1 and q + k when the march goes up the slice index,
          //  q + k and q + k + 1 when it goes down: there the last slice of the run borders the layer behind it)
          int run = flags ? __builtin_amdgcn_readlane(ent_run, q & 63) : 0;
          if (dir < 0 && run > 1) --run;
          run = min(run, npos + 1 - q);
          if (run > 0 && (inflight == 0 || run >= 8)) {
            if (inflight > 0) {
              wait_vmcnt(0);
              landed += inflight * QSTEP;
              inflight = 0;
              if (FIFO) fly_total = 0;
            }
            const int mine = (run + QSTEP - 1) / QSTEP;  // my slices among q .. q + run - 1
            if (gl == 0) n_skipped += mine;
            q += mine * QSTEP;
            landed = q;
            raw_lds_st_b32(&ctl[(lid < 4 ? 4 : 20) + lid], landed);
            slot_q = q % nslots;
            continue;
          }
          if (q - nslots >= minp) {
            minp = poll_progress();
            if (DIAG && (P.lockstep & 64)) minp = 0x3ffffff0;
            if (minp >= SLAB_DONE) {  // every consumer finished: the rest is not needed
              stop = true;
              break;
            }
            if (q - nslots >= minp) break;
          }
          const int uv = __builtin_amdgcn_readlane(ent_uv, q & 63);
          // Small windows count on every slice being the same number of DMA instructions (the in-order vmcnt): there an
          // unneeded slice may issue nothing only while none is in flight before it -- else it goes the uniform way with
          // `mych` one-unit reads.
          bool skip = run > 0;
          if (!FIFO && inflight > 0) skip = false;
          if (skip && gl == 0) ++n_skipped;
          int issued = 0;  // DMA wave-instructions of this slice (this loader's share)
          const unsigned dst0 = ring_addr + (unsigned)(slot_q * Q.slot_bytes + gl * per * 1024);
          if (skip) {
            // nothing
          } else if (uv != -1) {
            // (small windows: the whole shape -- the saving would not pay for the partial-group path)
#ifndef SLAB_LIGHT_FULLROWS
#define SLAB_LIGHT_FULLROWS 0
#endif
            const unsigned need_u = (SLAB_LIGHT_FULLROWS && !FIFO) ? (unsigned)Q.wp : Q.mask_need ? (((unsigned)uv >> 22) & 0x3fu) + 1u : (unsigned)Q.wu;
            const unsigned need_v = Q.mask_need ? (unsigned)Q.wv - ((unsigned)uv >> 28) * (unsigned)((Q.wv + 15) / 16) : (unsigned)Q.wv;
            const int sl = (dir > 0 ? smin + q : smax + 1 - q) - Q.Os;
            const unsigned u0 = (unsigned)uv & 0x7ffu, v0 = ((unsigned)uv >> 11) & 0x7ffu;
            const char *src = gv + (l2hot ? (size_t)0 : (size_t)sl * strideSb) + ((size_t)v0 * strideVb + (size_t)u0 * VB) +  // (64-bit: v0 * strideVb passes 4 GiB when V is the slowest axis of a 1024^3 volume)
                              (size_t)(gl * rpg) * strideVb;
            unsigned dst = dst0, row0 = (unsigned)(gl * rpg);
            // per = wp / gcd(64, wp) is 1, 3, 5 or 7 (the host picks such a pitch)
#define SLAB_GROUP(CHUNK)              \
  CHUNK(0)                             \
  if (per >= 3) {                      \
    CHUNK(1) CHUNK(2)                  \
    if (per >= 5) {                    \
      CHUNK(3) CHUNK(4)                \
      if (per >= 7) { CHUNK(5) CHUNK(6) } \
    }                                  \
  }
            // Lanes outside what THIS slice needs of the window stay idle (the fixed shape is sized
            // for the widest section of the bundle; at a voxel per pixel the mean need is ~70 % of
            // it).  Column masks are per slice; groups past the needed rows are not issued at all;
            // only the group the needed rows end in pays a per-lane row test (lane 0 always loads
            // there, so that a group is `per` wave-instructions: the in-order vmcnt counts slices
            // through the per-slice instruction counts kept in fly_counts).
            bool cm[7];
            unsigned long long mk[7];  // the same column masks as wave-uniform lane masks (EXEC values)
#pragma unroll
            for (int k = 0; k < 7; ++k) {
              cm[k] = k < per && colk[k] < need_u;
              mk[k] = __builtin_amdgcn_ballot_w64(cm[k]);
            }
#define CM(k) cm[k]
            for (int g = 0; g < mygroups; ++g) {
              if (FIFO && row0 >= need_v) break;  // nothing of this group (or the following ones) is needed: not issued, not counted
              issued += per;
              // (small workgroups issue every row group of the window whatever the slice needs of it -- the instruction
              //  count is the same either way -- so a group that lies inside the window goes the branch-free way even
              //  when its last rows are not needed; the host makes the window a whole number of groups where it can)
              if (row0 + (unsigned)rpg <= (BIG ? need_v : (unsigned)Q.wv)) {
                if (per == 3) {
                  // a whole group in ONE statement: EXEC takes each chunk's column mask in turn, M0 steps
                  // through the chunks' LDS images -- three scalar instructions per chunk and no branch
                  // (every chunk has a column-0 lane, so no mask is empty)
                  unsigned long long keep_exec;
                  asm volatile("s_mov_b32 %[km], m0\n\ts_mov_b64 %[ke], exec\n\ts_mov_b32 m0, %[dst]\n\t"
                               "s_mov_b64 exec, %[e0]\n\tglobal_load_lds_dwordx4 %[v0], %[src]" SLAB_DMA_POLICY "\n\ts_add_u32 m0, m0, 0x400\n\t"
                               "s_mov_b64 exec, %[e1]\n\tglobal_load_lds_dwordx4 %[v1], %[src]" SLAB_DMA_POLICY "\n\ts_add_u32 m0, m0, 0x400\n\t"
                               "s_mov_b64 exec, %[e2]\n\tglobal_load_lds_dwordx4 %[v2], %[src]" SLAB_DMA_POLICY "\n\t"
                               "s_mov_b64 exec, %[ke]\n\ts_mov_b32 m0, %[km]"
                               : [km] "=&s"(keep_m0), [ke] "=&s"(keep_exec)
                               : [dst] "s"(dst), [src] "s"(src), [e0] "s"(mk[0]), [e1] "s"(mk[1]), [e2] "s"(mk[2]), [v0] "v"(voff[0]),
                                 [v1] "v"(voff[1]), [v2] "v"(voff[2])
                               : "memory", "scc");
                } else if (per == 1) {  // one chunk per group (pitches 8, 16, 32, 64): the same, once
                  unsigned long long keep_exec;
                  asm volatile("s_mov_b32 %[km], m0\n\ts_mov_b64 %[ke], exec\n\ts_mov_b32 m0, %[dst]\n\t"
                               "s_mov_b64 exec, %[e0]\n\tglobal_load_lds_dwordx4 %[v0], %[src]" SLAB_DMA_POLICY "\n\t"
                               "s_mov_b64 exec, %[ke]\n\ts_mov_b32 m0, %[km]"
                               : [km] "=&s"(keep_m0), [ke] "=&s"(keep_exec)
                               : [dst] "s"(dst), [src] "s"(src), [e0] "s"(mk[0]), [v0] "v"(voff[0])
                               : "memory", "scc");
                } else {
#define SLAB_CHUNK_ROWS_OK(k) \
  if (CM(k)) SLAB_DMA(src, dst + k * 1024u, voff[k]);
                  SLAB_GROUP(SLAB_CHUNK_ROWS_OK)
#undef SLAB_CHUNK_ROWS_OK
                }
              } else {
#define SLAB_CHUNK_MASKED(k)                                                                  \
  {                                                                                           \
    const bool in_need = CM(k) && row0 + rowk[k] < need_v;                                    \
    const unsigned vo = in_need ? voff[k] : 0u; /* lane 0 re-reads the group's first unit */  \
    if (in_need || lane == 0) SLAB_DMA(src, dst + k * 1024u, vo);                             \
  }
                SLAB_GROUP(SLAB_CHUNK_MASKED)
#undef SLAB_CHUNK_MASKED
              }
              src += gstep;
              dst += (unsigned)(gn * per * 1024);
              row0 += (unsigned)(gn * rpg);
#ifndef SLAB_PUBLISH_PER_GROUP
#define SLAB_PUBLISH_PER_GROUP 1
#endif
              if (SLAB_PUBLISH_PER_GROUP) publish_landed(issued);
            }
            if (!SLAB_PUBLISH_PER_GROUP) publish_landed(issued);
#undef SLAB_GROUP
#undef CM
          } else if (!FIFO) {
            // slice outside the stored box (never read): uniform counting wants its instructions all the same
            unsigned dst = dst0;
            for (int c = 0; c < mych; ++c) {
              SLAB_DMA(gv, dst, voff[0] * 0u);
              dst += 1024u;
            }
          }
          if (FIFO) {  // big windows: slices differ in what they issue, the counts are kept per slice
            fly_counts = lane == (q & 63) ? issued : fly_counts;
            fly_total += issued;
          }
          q += QSTEP;
          ++inflight;
          slot_q += QSTEP;
          if (slot_q >= nslots) slot_q -= nslots;  // (QSTEP <= 8 < 3 <= nslots is not guaranteed: see the host's ring check)
        }
        if (prof) {
          const long long t = __builtin_amdgcn_s_memtime();
          t_issue += t - t_mark;
          t_mark = t;
        }
        if (stop) break;
        publish_landed(0);
        if (EAGER && inflight < Q.maxfly && q <= npos && q - nslots < minp) continue;  // room again: issue on
        if (inflight > 0) {
          // retire the oldest slice in flight: everything but the younger slices' DMAs is done
          if (FIFO) {
            fly_total -= __builtin_amdgcn_readlane(fly_counts, (q - inflight * QSTEP) & 63);
            wait_vmcnt(fly_total);
          } else {
            wait_vmcnt(mych * (inflight - 1));  // small windows: every slice is mych wave-instructions
          }
          --inflight;
          landed += QSTEP;
          raw_lds_st_b32(&ctl[(lid < 4 ? 4 : 20) + lid], landed);
          if (prof) t_wait += (long long)__builtin_amdgcn_s_memtime() - t_mark;
        } else {
          const int flagged = raw_lds_b32(&ctl[3]);
          if (++idle > (1 << 22) || flagged) {  // bounded spin (see consumers)
            if (!flagged) raw_lds_st_b32(&ctl[3], 1);  // (a violated window bound stays the reported cause)
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          if (prof) t_idle += (long long)__builtin_amdgcn_s_memtime() - t_mark;
        }
      }
#undef SLAB_DMA
      wait_vmcnt(0);
      if (n_skipped && lane == 0) atomicAdd(&ctl[2], n_skipped);
      if (tracing && !phases && lane == 0 && lid == 0) {
        unsigned *t = Q.trace + 8 * (size_t)blockIdx.x;
        t[4] = (unsigned)(t_issue >> 6);
        t[5] = (unsigned)(t_wait >> 6);
        t[6] = (unsigned)(t_idle >> 6);
      }
      if (prof && lane == 0 && lid == 0) {
        atomicAdd(&Q.diag[4], (float)t_issue * 1e-3f);
        atomicAdd(&Q.diag[5], (float)t_wait * 1e-3f);
        atomicAdd(&Q.diag[6], (float)t_idle * 1e-3f);
        atomicAdd(&Q.diag[7], (float)((long long)__builtin_amdgcn_s_memtime() - t_start) * 1e-3f);
      }
    } else {
      // ================================ consumer waves ==========================================
      // One sample per lane and iteration.  A lane takes its next sample as soon as the two
      // slices it touches have landed; the wave's progress is the position of its slowest lane.
      // While data is ahead of the wave every unfinished lane is active in every iteration (a
      // fixed slice step per iteration would leave the lanes without a sample in it idle).
      // slices landed = the slowest loader's count
      // (one 16-byte LDS read per poll: the four words are adjacent and aligned)
      auto landed_all = [&]() -> int {
        int4 v;
        const unsigned a = (unsigned)(size_t)(lds_cptr_t)(ctl + 4);
        if constexpr (NL > 4) {  // loaders 4..7 publish at ctl[24..27] (words that stay at SLAB_DONE when unused)
          int4 w;
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:80\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v), "=&v"(w) : "v"(a) : "memory");
          return min(min(min(v.x, v.y), min(v.z, v.w)), min(min(w.x, w.y), min(w.z, w.w)));
        }
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        return min(min(v.x, v.y), min(v.z, v.w));
      };
      const bool stream_only = DIAG && (P.lockstep & 2) != 0;  // (diagnostic: consume nothing)
      if (DIAG && (P.lockstep & 64)) pb = SLAB_DONE;   // (diagnostic: free-running stream)
      const bool count = DIAG && (P.lockstep & 48) != 0 && Q.diag != nullptr;
      float n_it = 0.f, n_act = 0.f, n_in = 0.f, n_hit = 0.f, n_anyhit = 0.f, n_lead = 0.f, n_waits = 0.f, n_wstep = 0.f;
      float n_work = 0.f, n_own = 0.f;
      int have = 0;  // cached copy of `landed` (monotonic): re-polled only when a lane is blocked on it
      const float inv_bs = 1.0f / B[AS];  // (planes per slice along this ray; the principal axis' B is never 0)
      // table row of the sample's base slice: entry index = bs - Os = psgn*pb + (-psgn*poff - Os)
      const int eoff = -psgn * poff - Q.Os;
      const unsigned wtab_addr = (unsigned)(size_t)(lds_cptr_t)wtab;
      for (int it = 0;; ++it) {
        const bool want = pb < SLAB_DONE;
        if (!__any(want)) break;  // every ray of this wave is finished
        // progress = position of the slowest lane (DPP reduction, ~12 VALU), published every
        // iteration on a short ring and every other one where a stale value only delays slot
        // recycling by a step
        if (!EARLY && !(it & Q.pmask)) {
          const int plo = wave_min_i32(pb);
          if (plo != pos) {  // positions below plo are done: their lower slices may be recycled
            pos = plo;       // (every slot read of earlier iterations has returned: slab_read8 waits)
            if (lane == 0) lds_st(&ctl[8 + wave], pos);
          }
        }
        // Take a step only when the whole band of this wave can: lanes sit up to one sample
        // spacing (wstep slices) ahead of the slowest one, and an iteration costs the wave the
        // same issue slots whether 5 or 64 lanes take part.  Stepping the moment ONE lane's slices
        // have landed was measured at up to 2.3x the iterations per ray on a starved stream.
        int need = min(pos + 2 + Q.wstep, npos + 1);
        if (have < need) {
          have = landed_all();
          if (have < need) {
            // make sure `pos` is this wave's true minimum before sleeping on it
            const int plo = wave_min_i32(pb);
            if (plo != pos) {
              pos = plo;
              if (lane == 0) lds_st(&ctl[8 + wave], pos);
              need = min(pos + 2 + Q.wstep, npos + 1);
            }
            if (count) n_waits += 1.f;
            for (int spins = 0; have < need; ++spins) {
              const int flagged = lds_ld(&ctl[3]);
              if (spins > (1 << 22) || flagged) {  // bounded: never hang the GPU on a protocol bug
                if (!flagged) lds_st(&ctl[3], 1);
                have = 0x3ffffff0;
                m1 = m - 1;
                pb = SLAB_DONE;
                break;
              }
              // (backing off further for waves whose first slice is many slices away was measured
              //  slower: reaction time matters more than the polls' issue slots)
#ifndef SLAB_POLL_SLEEP
#define SLAB_POLL_SLEEP 2
#endif
              __builtin_amdgcn_s_sleep(SLAB_POLL_SLEEP);
              have = landed_all();
            }
          }
          have = __builtin_amdgcn_readfirstlane(have);
        }
        // (Measured and dropped: SLAB_REPS samples per lane and turn of the loop, to pay the loop's own
        //  cost -- progress word, poll, flow control -- once per two or three samples.  512^3 f32 frame,
        //  REPS 1 / 2 / 3: 1.89 / 2.02 / 1.97 ms; the scalar instruction count did not move (4.74e8 ->
        //  4.81e8) and the vector one rose 12 %: lanes whose next slices have not landed sit the extra
        //  sample out, so the second body mostly runs with few lanes.  Kept as a build knob.)
#ifndef SLAB_REPS
#define SLAB_REPS 1
#endif
#pragma unroll
        for (int rep = 0; rep < (EARLY ? 1 : SLAB_REPS); ++rep) {
        const bool act = pb < SLAB_DONE && pb + 2 <= have;
        if (count && rep == 0) {
          n_lead += (float)(have - pos);
          n_wstep += (float)Q.wstep;
        }
        asm volatile("" ::: "memory");  // slot reads stay behind the poll
        bool d_hit = false, d_own = false;  // (diagnostic counters only)
        // ---- part A: where the sample is and its corner addresses; EARLY: the whole corner batch
        float fx = 0.f, fy = 0.f, fz = 0.f;
        unsigned a0 = 0, b0 = 0;
        typename std::conditional<DT == 0, v2u, v4f>::type rq[EARLY ? 8 : 1];
        float early_nsc = 0.f;
        int early_ni = 0;
        bool work = act && !stream_only;
        int base_a = 0, base_b = 0;
        if (work) {
          // the two slices' slot images (one 8-byte table entry each, adjacent): issued first,
          // the position arithmetic below covers the LDS round trip
          const int *te = reinterpret_cast<const int *>(smem + (wtab_addr - ring_addr)) + 2 * (__mul24(psgn, pb) + eoff);  // (24-bit multiply: full rate; |pb| < 4096 on a lane that works)
          base_a = te[0];
          base_b = BR ? (te[2] & ~1) : te[2];  // (bit 0: the NEXT layer's flag)
        }
        // EMPTY LAYERS: nothing in this sample's layer can be visible for any ray of the tile -- the sample is exactly
        // transparent, and its slices may not even have been streamed
        if (flags) work = work && !(base_a & 1);
        // the plane of this ray's next sample: the next one, or -- from a layer that starts a run of empty ones (the
        // entry holds its length, see the set-up) -- the first plane whose base slice lies behind the run.  The planes
        // in between fall into empty layers, all of them: positions are monotone in the plane index.
        int m_next = m + 1;
#ifndef SLAB_NO_JUMP
        if (flags && act && (base_a & 3) == 3) {
          const int ptar = pb + (base_a >> 2);  // first position behind the run
          if (ptar >= npos) m_next = m1 + 1;    // nothing but empty layers to the end of the tile's range
          else {
            // s(m) = fma(m, B, A) reaches the target slice at m = t; floor(t) is never behind the exact answer (the
            // division is off by far less than a plane), the exact base slices of the planes from there on decide
            const float starget = (float)(dir > 0 ? smin + ptar : smax - ptar + 1);
            int mj = max((int)floorf((starget - A[AS]) * inv_bs), m + 1);
#pragma unroll 1
            for (int k = 0; k < 4 && mj <= m1; ++k) {
              if (__mul24(psgn, base_slice(mj)) + poff >= ptar) break;
              ++mj;
            }
            m_next = mj;
          }
        }
#endif
        if (work) {
          const float mf = (float)m;
          // (no membership test: [m, m1] is exactly the inside interval, see the set-up)
          // (the principal axis' clamped coordinate and base index were computed when this sample's position in
          //  the stream was: fma(m, B, A) -> clamp -> min((int), N - 2), the very operations of smk_lin_clamp)
          int x0 = 0, x1, y0 = 0, y1, z0 = 0, z1;
          if constexpr (AS != 0) smk_lin_clamp(__fmaf_rn(mf, B[0], A[0]), P.N[0], x0, x1, fx);
          else { x0 = car_i; fx = car_sc - (float)car_i; }
          if constexpr (AS != 1) smk_lin_clamp(__fmaf_rn(mf, B[1], A[1]), P.N[1], y0, y1, fy);
          else { y0 = car_i; fy = car_sc - (float)car_i; }
          if constexpr (AS != 2) smk_lin_clamp(__fmaf_rn(mf, B[2], A[2]), P.N[2], z0, z1, fz);
          else { z0 = car_i; fz = car_sc - (float)car_i; }
          (void)x1; (void)y1; (void)z1;
          const int iu = AU == 0 ? x0 : y0, iv = AV == 1 ? y0 : z0;  // global voxel indices
          if (count && flags)  // (diagnostic: would this lane's OWN brick have let it skip the sample?)
            d_own = Q.bricks[(size_t)((car_i - Q.Os) >> SMK_BRICK_LOG2) * Q.bss + (size_t)((iv - Q.Ov) >> SMK_BRICK_LOG2) * Q.bsv +
                             (size_t)((iu - Q.Ou) >> SMK_BRICK_LOG2) * Q.bsu] != 0;
          const unsigned lo_off = __umul24((unsigned)iv, pitch_b) + ((unsigned)iu << VBL);  // (24-bit multiply: full rate)
          a0 = (unsigned)base_a + lo_off;
          b0 = (unsigned)base_b + lo_off;
          if constexpr (EARLY) slab_read8_full(a0, a0 + pitch_b, b0, b0 + pitch_b, rq);
        }
        if constexpr (EARLY) {
          // everything this iteration needs of the ring is in registers: release the slots NOW, not
          // a classification + shading later -- on a 5-slot ring the loaders otherwise sit blocked
          // for most of the consumers' iteration
          int pbn = pb;
          float nsc = car_sc;
          int ni = car_i;
          if (act) {
            if (m_next <= m1) {
              ni = base_slice_c(m_next, nsc);
              pbn = __mul24(psgn, ni) + poff;
            } else pbn = SLAB_DONE;
          }
          early_nsc = nsc;
          early_ni = ni;
          // (every turn: on the 5-slot ring a progress word that is one turn late costs 16 % -- 4.07 vs 4.71 ms)
          const int plo = wave_min_i32(pbn);
          if (plo != pos && plo < SLAB_DONE) {
            pos = plo;
            if (lane == 0) lds_st(&ctl[8 + wave], pos);
          }
        }
        if (work) {
          // corners <ds><dv><du> -> model order <dx><dy><dz>; lerp order x, y, z like the gather kernel
#define QI(dx, dy, dz) (PERM == 0 ? ((dz) * 4 + (dy) * 2 + (dx)) : PERM == 1 ? ((dy) * 4 + (dz) * 2 + (dx)) : ((dx) * 4 + (dz) * 2 + (dy)))
#define TRI(E)                                                                                                             \
  smk_lerp(smk_lerp(smk_lerp(E(0, 0, 0), E(1, 0, 0), fx), smk_lerp(E(0, 1, 0), E(1, 1, 0), fx), fy),                       \
           smk_lerp(smk_lerp(E(0, 0, 1), E(1, 0, 1), fx), smk_lerp(E(0, 1, 1), E(1, 1, 1), fx), fy), fz)
          float ch0, ch1, ch2 = 0.f, ch3 = 0.f;
          // big workgroups, float voxels, separable table: the third channel is only looked at behind the (v, g) quad's
          // occupancy bit -- one sample in fourteen on the 1024^3 frame -- and its corners are in registers anyway, so
          // its interpolation (14 packed instructions) waits until then
          // (small workgroups: the ring slots are still held, so the third channel is READ only then, too: 8 x 8 bytes
          //  per sample instead of 8 x 12)
          // (byte voxels: all four channels come in one word per corner, the interpolation alone is deferred)
          const bool lazy_h = (TF == 1 && Q.fast_tf) || (TF == 2 && Q.use_occ);
          uint32_t q8[DT == 0 && !EARLY ? 8 : 1];  // small workgroups, byte voxels: the corners' data words
          auto tri_h_early = [&]() -> float {
            if constexpr (EARLY && DT == 0) {
#define E2(dx, dy, dz) smk_ub(rq[QI(dx, dy, dz)].x, 2)
              return TRI(E2) * SMK_INV255;
#undef E2
            } else if constexpr (DT == 0) {
#define E2(dx, dy, dz) smk_ub(q8[QI(dx, dy, dz)], 2)
              return TRI(E2) * SMK_INV255;
#undef E2
            } else if constexpr (EARLY && DT == 1) {
#define E2(dx, dy, dz) rq[QI(dx, dy, dz)].z
              return TRI(E2);
#undef E2
            } else if constexpr (DT == 1) {
              uint32_t hq[8];
              slab_read8_h16(a0, a0 + pitch_b, b0, b0 + pitch_b, hq);
#define E2(dx, dy, dz) __uint_as_float(hq[QI(dx, dy, dz)])
              return TRI(E2);
#undef E2
            } else {
              return 0.f;
            }
          };
          if constexpr (EARLY && DT == 1) {
#define E0(dx, dy, dz) rq[QI(dx, dy, dz)].x
#define E1(dx, dy, dz) rq[QI(dx, dy, dz)].y
            ch0 = TRI(E0);
            ch1 = TRI(E1);
            if ((TF == 2 || P.third_axis) && !lazy_h) ch2 = tri_h_early();
#undef E1
#undef E0
          } else if constexpr (EARLY && DT == 0) {
#define E0(dx, dy, dz) smk_ub(rq[QI(dx, dy, dz)].x, 0)
#define E1(dx, dy, dz) smk_ub(rq[QI(dx, dy, dz)].x, 1)
#define E2(dx, dy, dz) smk_ub(rq[QI(dx, dy, dz)].x, 2)
#define E3(dx, dy, dz) smk_ub(rq[QI(dx, dy, dz)].x, 3)
            ch0 = TRI(E0) * SMK_INV255;
            ch1 = TRI(E1) * SMK_INV255;
            if ((TF == 2 || P.third_axis) && !lazy_h) {
              ch2 = TRI(E2) * SMK_INV255;
              if (P.nelts == 4) ch3 = TRI(E3) * SMK_INV255;
            }
#undef E3
#undef E2
#undef E1
#undef E0
          } else if constexpr (DT == 1) {
            if ((TF == 2 || P.third_axis) && !lazy_h) {
              v3f q[8];
              slab_read8(a0, a0 + pitch_b, b0, b0 + pitch_b, q);
#define E0(dx, dy, dz) q[QI(dx, dy, dz)].x
#define E1(dx, dy, dz) q[QI(dx, dy, dz)].y
#define E2(dx, dy, dz) q[QI(dx, dy, dz)].z
              ch0 = TRI(E0);
              ch1 = TRI(E1);
              ch2 = TRI(E2);
#undef E2
            } else {
              v2f q[8];
              slab_read8(a0, a0 + pitch_b, b0, b0 + pitch_b, q);
              ch0 = TRI(E0);
              ch1 = TRI(E1);
#undef E1
#undef E0
            }
          } else {
            uint32_t (&q)[DT == 0 && !EARLY ? 8 : 1] = q8;
            if constexpr (DT == 0 && !EARLY) slab_read8_u8(a0, a0 + pitch_b, b0, b0 + pitch_b, q8);
#define E0(dx, dy, dz) smk_ub(q[QI(dx, dy, dz)], 0)
#define E1(dx, dy, dz) smk_ub(q[QI(dx, dy, dz)], 1)
#define E2(dx, dy, dz) smk_ub(q[QI(dx, dy, dz)], 2)
#define E3(dx, dy, dz) smk_ub(q[QI(dx, dy, dz)], 3)
            ch0 = TRI(E0) * SMK_INV255;
            ch1 = TRI(E1) * SMK_INV255;
            if ((TF == 2 || P.third_axis) && !lazy_h) {
              ch2 = TRI(E2) * SMK_INV255;
              if (P.nelts == 4) ch3 = TRI(E3) * SMK_INV255;
            }
#undef E3
#undef E2
#undef E1
#undef E0
          }
#undef TRI
          float4 col;
          bool hit;
          SlabTexel4 tx4 = {0, 0, 0, 0, 0.f, 0.f};
          if (TF == 1 && Q.fast_tf) {
            int s0, s1, t0, t1;
            float fs, ft;
            smk_lin_clamp(__fmaf_rn(ch0, (float)P.sv, -0.5f), P.sv, s0, s1, fs);
            smk_lin_clamp(__fmaf_rn(ch1, (float)P.sg, -0.5f), P.sg, t0, t1, ft);
            // occupancy bit of the texel quad (LDS): clear => alpha is exactly 0, no fetch.  Most
            // samples of a typical transfer function end here, without the L2 round trip.
            bool maybe = true;
            if (Q.use_occ) maybe = (occ[__mul24(t0, P.occ_roww) + (s0 >> 5)] >> (s0 & 31)) & 1u;
            col.w = 0.0f;
            if (maybe) {
              // alpha first (all four (V,G) texels are needed for it anyway); colour only on a hit
              tx4 = slab_tex2d_fetch(P.tf_vg, P.sv, s0, t0, fs, ft);
              col.w = slab_tex_chan(tx4, 3);
              if (Q.use_ah) {  // third-axis alpha (the same products as smk_classify)
                if (lazy_h) ch2 = tri_h_early();
                int h0, h1;
                float fh;
                smk_lin_clamp(__fmaf_rn(ch2, (float)P.sv, -0.5f), P.sv, h0, h1, fh);
                col.w *= smk_lerp(ah[h0], ah[h0 + 1], fh) * SMK_INV255;
              }
              col.w = smk_sat(col.w);
            }
            hit = col.w != 0.0f;
          } else if (TF == 2 && Q.use_occ) {
            // dense 3-D table: the (v, g) base texel's occupancy bit, folded over the sheets (smk_set_tf3d), decides
            // whether the eight-texel gather can return anything but alpha == 0
            int s0, s1, t0, t1;
            float fs, ft;
            smk_lin_clamp(__fmaf_rn(ch0, (float)P.s3v, -0.5f), P.s3v, s0, s1, fs);
            smk_lin_clamp(__fmaf_rn(ch1, (float)P.s3g, -0.5f), P.s3g, t0, t1, ft);
            col.w = 0.0f;
            hit = false;
            if ((occ[__mul24(t0, P.occ_roww) + (s0 >> 5)] >> (s0 & 31)) & 1u) {
              ch2 = tri_h_early();  // (the third coordinate of the lookup: only now)
              hit = smk_classify<DT, TF>(P, ch0, ch1, ch2, ch3, col);
            }
          } else {
            hit = smk_classify<DT, TF>(P, ch0, ch1, ch2, ch3, col);
          }
          d_hit = hit;
          if (hit) {
            if (TF == 1 && Q.fast_tf) {
              col.x = slab_tex_chan(tx4, 0);
              col.y = slab_tex_chan(tx4, 1);
              col.z = slab_tex_chan(tx4, 2);
            }
            float4 src;
            // frames with shadows: the light-buffer colour over the sample, as the slices nearer the light left it
            // (smk_shadow.hip; the sample's own position: the gather kernel's fma chain)
            float shadow[3];
            const float *shp = nullptr;
            if (TF != 0 && SHD) {
              const float mf = (float)m;
              smk_shadow_term(P, m, __fmaf_rn(mf, B[0], A[0]), __fmaf_rn(mf, B[1], A[1]), __fmaf_rn(mf, B[2], A[2]), shadow);
              shp = shadow;
            }
            if (TF == 0) {
              src = col;  // the 1-D colour table's entries are premultiplied (TLUT.cpp:65-71), as in the gather kernel
            } else if (SH == 0) {
              src = smk_shade_sample<0>(P, col, 0.f, 0.f, 0.f, 0.f, shp);
            } else {
              uint32_t nb[8];  // the packed normals of the same corners: second batch, or already here (EARLY)
              if constexpr (EARLY) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                  if constexpr (DT == 1) nb[k] = __float_as_uint(rq[k].w);
                  else nb[k] = rq[k].y;
                }
              } else if (DT == 1) slab_read8_nb16(a0, a0 + pitch_b, b0, b0 + pitch_b, nb);
              else slab_read8_nb8(a0, a0 + pitch_b, b0, b0 + pitch_b, nb);
#define NB(dx, dy, dz) nb[QI(dx, dy, dz)]
              float n0 = smk_nrm(NB(0, 0, 0), NB(1, 0, 0), NB(0, 1, 0), NB(1, 1, 0), NB(0, 0, 1), NB(1, 0, 1), NB(0, 1, 1), NB(1, 1, 1), 0, fx, fy, fz);
              float n1 = smk_nrm(NB(0, 0, 0), NB(1, 0, 0), NB(0, 1, 0), NB(1, 1, 0), NB(0, 0, 1), NB(1, 0, 1), NB(0, 1, 1), NB(1, 1, 1), 1, fx, fy, fz);
              float n2 = smk_nrm(NB(0, 0, 0), NB(1, 0, 0), NB(0, 1, 0), NB(1, 1, 0), NB(0, 0, 1), NB(1, 0, 1), NB(0, 1, 1), NB(1, 1, 1), 2, fx, fy, fz);
#undef NB
              src = smk_shade_sample<SH>(P, col, n0, n1, n2, ch1, shp);
            }
            // first-hit depth (the gather kernel's `first`): the first sample that passes classification finds the accumulated
            // alpha still exactly 0, no later one does -- nothing is carried through the loop for it
            if (P.depth != nullptr && C3 == 0.0f) P.depth[(size_t)j * P.W + i] = __fmaf_rn((float)m, rc.dtau, rc.tau0) * P.znear;
            if (P.blend == SMK_BLEND_MAX) {  // GL_MAX (gluvvShadeMIP): no order, no termination
              C0 = fmaxf(C0, src.x);
              C1 = fmaxf(C1, src.y);
              C2 = fmaxf(C2, src.z);
              C3 = fmaxf(C3, src.w);
            } else {
              float w = 1.0f - C3;
              C0 = __fmaf_rn(w, src.x, C0);
              C1 = __fmaf_rn(w, src.y, C1);
              C2 = __fmaf_rn(w, src.z, C2);
              C3 = __fmaf_rn(w, src.w, C3);
              // exact early termination: once A == 1.0f every later weight (1-A) is exactly 0,
              // so no later sample can change C or A
              if (C3 == 1.0f) m1 = m;
            }
          }
#undef QI
        }
        if (act) {
          m = m_next;
          if constexpr (EARLY) {
            // (the next sample's slice was worked out when the ring slots were released; m1 may since have shrunk to m - 1
            //  -- the ray saturated -- which ends it)
            if (m <= m1) {
              car_sc = early_nsc;
              car_i = early_ni;
              pb = __mul24(psgn, car_i) + poff;
            } else pb = SLAB_DONE;
          } else if (m <= m1) {
            car_i = base_slice_c(m, car_sc);
            pb = __mul24(psgn, car_i) + poff;
          } else pb = SLAB_DONE;
        }
        if (count) {
          n_it += 1.f;
          n_act += (float)__popcll(__ballot(act));
          n_in += (float)__popcll(__ballot(work));  // lanes that interpolate a sample (an empty layer's are skipped before that)
          n_hit += (float)__popcll(__ballot(d_hit));
          n_anyhit += __any(d_hit) ? 1.f : 0.f;
          n_work += __any(work) ? 1.f : 0.f;
          n_own += __any(work && d_own) ? 1.f : 0.f;
        }
        }  // rep
      }
      if (lane == 0) lds_st(&ctl[8 + wave], SLAB_DONE);
      if (tracing && lane == 0) atomicAdd(Q.trace + 8 * (size_t)blockIdx.x + 7, (unsigned)n_it);
      if (count && lane == 0) {
        atomicAdd(&Q.diag[0], n_it);
        atomicAdd(&Q.diag[1], n_act);
        atomicAdd(&Q.diag[2], n_in);
        atomicAdd(&Q.diag[3], n_hit);
        atomicAdd(&Q.diag[8], n_anyhit);
        atomicAdd(&Q.diag[14], n_work);
        atomicAdd(&Q.diag[15], n_own);
        atomicAdd(&Q.diag[9], n_lead);
        atomicAdd(&Q.diag[10], n_waits);
        atomicAdd(&Q.diag[11], n_wstep);
        // (how much of the tile's slice range this wave did not need: it finished at position `pos`)
        atomicAdd(&Q.diag[12], (float)max(npos - min(pos, npos), 0) / (float)max(npos, 1));
        atomicAdd(&Q.diag[13], 1.0f);
      }
    }
  }
  if (live) {
    size_t o = (size_t)j * P.W + i;
    float4 *outp = seg == 0 ? P.out : Q.seg_out + (size_t)(seg - 1) * ((size_t)P.W * P.H);
    outp[o] = make_float4(C0, C1, C2, C3);
    if (P.depth != nullptr && C3 == 0.0f) P.depth[o] = __int_as_float(0x7f800000);  // (no sample passed classification)
  }
  // errors are reported, never swallowed: the host turns a non-zero status into a failed frame
  if (npos > 0) {
    __syncthreads();
    if (tid == 0 && ctl[3]) *(volatile int *)Q.status = Q.status_tag | ctl[3];
  }
  if (tid == 0 && Q.tile_ticks && nseg > 1) {  // (a split tile's words are sums over its workgroups; zeroed by the launcher)
    const unsigned dur = max((unsigned)__builtin_amdgcn_s_memrealtime() - trace_t0, 1u);
    if (Q.piece_ticks) Q.piece_ticks[(size_t)tile * 8 + seg] = dur;
    atomicAdd(&Q.tile_ticks[tile], dur);
    atomicAdd(&Q.tile_ticks[Q.ntiles + tile], npos > 0 ? (unsigned)max(min(max(min(min(ctl[4], ctl[5]), min(ctl[6], ctl[7])), 0), npos + 1) - ctl[2], 0) : 0u);
    atomicAdd(&Q.tile_ticks[2 * Q.ntiles + tile], npos > 0 ? (unsigned)(npos + 1) : 0u);
  } else if (tid == 0 && Q.tile_ticks) {
    Q.tile_ticks[tile] = max((unsigned)__builtin_amdgcn_s_memrealtime() - trace_t0, 1u);
    // (ctl[4..7] = the loaders' landed words, final after the barrier above; their minimum = slices completely streamed)
    // (less the slices that were not streamed because nobody samples them: EMPTY LAYERS)
    Q.tile_ticks[Q.ntiles + tile] = npos > 0 ? (unsigned)max(min(max(min(min(ctl[4], ctl[5]), min(ctl[6], ctl[7])), 0), npos + 1) - ctl[2], 0) : 0u;
    Q.tile_ticks[2 * Q.ntiles + tile] = npos > 0 ? (unsigned)(npos + 1) : 0u;
    // (the product's own timeline, read by smk_get_trace when no diagnostic instance ran: when the workgroup started, and where)
    Q.tile_ticks[3 * Q.ntiles + tile] = trace_t0;
    Q.tile_ticks[4 * Q.ntiles + tile] = (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xff00u) | (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu);
  }
  if (tracing && tid == 0) {
    unsigned *t = Q.trace + 8 * (size_t)blockIdx.x;
    if (phases) {  // 100 MHz ticks since the workgroup started: slice range known | windows tabled | flags, runs, last barrier
      t[4] = ph1;
      t[5] = ph2;
      t[6] = ph3;
    }
    t[0] = trace_t0;
    t[1] = (unsigned)__builtin_amdgcn_s_memrealtime();
    t[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    t[3] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) | ((unsigned)tile << 8) | ((unsigned)max(npos, 0) << 20);  // XCC_ID, tile, slices
  }
}

// ------------------------------------------------------------------------------- host side

// This file is compiled three times (build time: the instances are most of it): as itself -- host side + the byte-voxel
// instances --, through smk_slab_f32.hip (SLAB_PART 1) -- the float-voxel instances alone -- and through smk_slab_shadow.hip
// (SLAB_PART 2) -- the instances of the eye pass of frames with shadows.
#ifndef SLAB_PART
#define SLAB_PART 0
#endif
#if SLAB_PART == 0
// DEPTH SEGMENTS, second half: the partial frames of a split tile, merged in marching order.  `list` entries: tile | segments << 20.
__global__ __launch_bounds__(256) void smk_k_slab_merge(const int2 *list, int tw, int th, int ntx, int W, int H, const float4 *seg_out, float4 *out, int use_max) {
  const int code = list[blockIdx.x].x;
  const int tile = code & 0xfffff, nseg = code >> 20;
  const int ty = tile / ntx, tx = tile - ty * ntx;
  const size_t npix = (size_t)W * H;
  for (int p = threadIdx.x; p < tw * th; p += blockDim.x) {
    const int i = tx * tw + p % tw, j = ty * th + p / tw;
    if (i >= W || j >= H) continue;
    const size_t o = (size_t)j * W + i;
    float4 C = out[o];
    for (int k = 1; k < nseg; ++k) {
      const float4 sgm = seg_out[(size_t)(k - 1) * npix + o];
      if (use_max) {
        C = make_float4(fmaxf(C.x, sgm.x), fmaxf(C.y, sgm.y), fmaxf(C.z, sgm.z), fmaxf(C.w, sgm.w));
      } else {
        const float w = 1.0f - C.w;
        C.x = __fmaf_rn(w, sgm.x, C.x);
        C.y = __fmaf_rn(w, sgm.y, C.y);
        C.z = __fmaf_rn(w, sgm.z, C.z);
        C.w = __fmaf_rn(w, sgm.w, C.w);
      }
    }
    out[o] = C;
  }
}

// a ray's coefficients on the host, for planning: at a real-valued position (px, py) of the image plane, in double -- the
// kernels' float chains (smk_ray_AB) round differently by less than the planning's own slack
static void host_ray_at(const RenderParams &P, double px, double py, double A[3], double B[3]) {
  const smk_raycoef &rc = P.rc;
  if (!P.sh.on) {
    for (int a = 0; a < 3; ++a) {
      A[a] = px * rc.Ax[a] + py * rc.Ay[a] + rc.Ac[a];
      B[a] = px * rc.Bx[a] + py * rc.By[a] + rc.Bc[a];
    }
    return;
  }
  const SmkShadowRays &sh = P.sh;  // frames with shadows: half-angle slices (smk_internal.h)
  const double nD = px * sh.nDx + py * sh.nDy + sh.nDc, tauA = sh.numA / nD, dtau = sh.dB / nD;
  for (int a = 0; a < 3; ++a) {
    const double D = px * sh.Dx[a] + py * sh.Dy[a] + sh.Dc[a];
    A[a] = sh.Ec[a] + tauA * D;
    B[a] = dtau * D;
  }
}
static void host_ray(const RenderParams &P, int i, int j, double A[3], double B[3]) {
  const smk_raycoef &rc = P.rc;
  const float px = fmaf((float)i + 0.5f, rc.pxs, rc.pxl), py = fmaf((float)j + 0.5f, rc.pys, rc.pyl);
  if (P.sh.on) {
    host_ray_at(P, px, py, A, B);
    return;
  }
  for (int a = 0; a < 3; ++a) {
    A[a] = fmaf(px, rc.Ax[a], fmaf(py, rc.Ay[a], rc.Ac[a]));
    B[a] = fmaf(px, rc.Bx[a], fmaf(py, rc.By[a], rc.Bc[a]));
  }
}

// ---- S-extent of (a tile's ray bundle between the first and the last sample plane) /\ (the region
// box), in voxel index coordinates.  The bundle is the pyramid section spanned by the rays through
// the tile's outer pixel EDGES (half a pixel beyond the corner pixels' centres, so a one-pixel-wide
// tile is not degenerate); rays are affine in the pixel coordinate, so every ray of the tile lies
// inside it.  Both bodies are convex: the extrema over the intersection sit on its vertices = the
// vertices of the box faces clipped by the pyramid's six planes + the pyramid's corners inside the
// box.  Corner rays alone do not bound this: all four may miss a volume that projects inside the
// tile, and a ray through a side face enters anywhere between the front and the back face.
namespace {
struct SlabVec { double v[3]; };

int slab_clip_polygon(const SlabVec *in, int n, const double pl[4], SlabVec *out) {  // keeps pl.(p,1) >= 0
  int m = 0;
  for (int k = 0; k < n; ++k) {
    const SlabVec &a = in[k], &b = in[(k + 1) % n];
    const double da = pl[0] * a.v[0] + pl[1] * a.v[1] + pl[2] * a.v[2] + pl[3];
    const double db = pl[0] * b.v[0] + pl[1] * b.v[1] + pl[2] * b.v[2] + pl[3];
    if (da >= 0) out[m++] = a;
    if ((da >= 0) != (db >= 0)) {
      const double t = da / (da - db);
      SlabVec c;
      for (int i = 0; i < 3; ++i) c.v[i] = a.v[i] + t * (b.v[i] - a.v[i]);
      out[m++] = c;
    }
  }
  return m;
}

bool slab_bundle_slice_range_exact(const RenderParams &P, double fx0, double fy0, double fx1, double fy1, int as, double *smin,
                                   double *smax);
// The common case without clipping: when the four corner rays enter the box through ONE face and leave it through ONE
// face (inside the sampled plane range), so does every ray between them -- the rays through a face form a convex set --
// and bundle /\ box is the hexahedron of the four entry and four exit points: its S-extent is theirs.  (A bundle that
// contains a box edge or vertex, or is cut by the first / last sample plane, goes the exact way below.)
static bool slab_bundle_slice_range_fast(const RenderParams &P, double fx0, double fy0, double fx1, double fy1, int as, double *smin,
                                         double *smax) {
  const smk_raycoef &rc = P.rc;
  const double fx[4] = {fx0, fx1, fx1, fx0}, fy[4] = {fy0, fy0, fy1, fy1};
  const double q0 = -0.5, q1 = (double)(rc.nplanes - 1) + 0.5, eps = 1e-3;
  int fin = -1, fout = -1;
  double mn = 1e300, mx = -1e300;
  for (int c = 0; c < 4; ++c) {
    const double px = fx[c] * (double)rc.pxs + (double)rc.pxl, py = fy[c] * (double)rc.pys + (double)rc.pyl;
    double A[3], B[3], te = -1e300, tx = 1e300;
    int ie = -1, ix = -1;
    host_ray_at(P, px, py, A, B);
    for (int a = 0; a < 3; ++a) {
      const double lo = (double)P.lo[a] - eps, hi = (double)P.hi[a] + eps;
      if (fabs(B[a]) < 1e-12) {
        if (A[a] < lo || A[a] > hi) return false;
        continue;
      }
      const double t1 = (lo - A[a]) / B[a], t2 = (hi - A[a]) / B[a];
      const double tn = t1 < t2 ? t1 : t2, tf = t1 < t2 ? t2 : t1;
      if (tn > te) { te = tn; ie = 2 * a + (t1 < t2 ? 0 : 1); }
      if (tf < tx) { tx = tf; ix = 2 * a + (t1 < t2 ? 1 : 0); }
    }
    if (!(te < tx) || te < q0 || tx > q1 || ie < 0 || ix < 0) return false;
    if (c == 0) { fin = ie; fout = ix; }
    else if (ie != fin || ix != fout) return false;
    const double se = A[as] + te * B[as], sx = A[as] + tx * B[as];
    mn = std::min(mn, std::min(se, sx));
    mx = std::max(mx, std::max(se, sx));
  }
  *smin = std::max(mn, (double)P.lo[as]);
  *smax = std::min(mx, (double)P.hi[as]);
  return true;
}

bool slab_bundle_slice_range(const RenderParams &P, double fx0, double fy0, double fx1, double fy1, int as, double *smin,
                             double *smax) {
  static const int check = getenv("SMK_DEBUG_SCAN") ? atoi(getenv("SMK_DEBUG_SCAN")) : 0;  // (developer: 1 = compare the short way with the exact one, 2 = exact only)
  if (check != 2 && slab_bundle_slice_range_fast(P, fx0, fy0, fx1, fy1, as, smin, smax)) {
    if (!check) return true;
    double a = 0, b = 0;
    const bool ok = slab_bundle_slice_range_exact(P, fx0, fy0, fx1, fy1, as, &a, &b);
    if (!ok || fabs(a - *smin) > 5e-3 || fabs(b - *smax) > 5e-3)  // (the exact way pads its clipping planes by 1e-6 of the scene's scale)
      fprintf(stderr, "[smk] SCAN MISMATCH tile (%g,%g)-(%g,%g): fast [%.9g, %.9g] exact %d [%.9g, %.9g]\n", fx0, fy0, fx1, fy1, *smin, *smax, (int)ok, a, b);
    return true;
  }
  return slab_bundle_slice_range_exact(P, fx0, fy0, fx1, fy1, as, smin, smax);
}

bool slab_bundle_slice_range_exact(const RenderParams &P, double fx0, double fy0, double fx1, double fy1, int as, double *smin,
                                   double *smax) {
  const smk_raycoef &rc = P.rc;
  const double fx[4] = {fx0, fx1, fx1, fx0}, fy[4] = {fy0, fy0, fy1, fy1};  // cyclic
  const double q0 = -0.5, q1 = (double)(rc.nplanes - 1) + 0.5;
  SlabVec F[4][2];
  double cen[3] = {0, 0, 0}, scale = 1.0;
  for (int c = 0; c < 4; ++c) {
    const double px = fx[c] * (double)rc.pxs + (double)rc.pxl, py = fy[c] * (double)rc.pys + (double)rc.pyl;
    double Ar[3], Br[3];
    host_ray_at(P, px, py, Ar, Br);
    for (int a = 0; a < 3; ++a) {
      const double A = Ar[a], B = Br[a];
      F[c][0].v[a] = A + q0 * B;
      F[c][1].v[a] = A + q1 * B;
      cen[a] += (F[c][0].v[a] + F[c][1].v[a]) / 8.0;
      scale = std::max(scale, std::max(fabs(F[c][0].v[a]), fabs(F[c][1].v[a])));
    }
  }
  double planes[6][4];
  int npl = 0;
  auto add_plane = [&](const SlabVec &a, const SlabVec &b, const SlabVec &c) {
    double e1[3], e2[3], n[4];
    for (int i = 0; i < 3; ++i) { e1[i] = b.v[i] - a.v[i]; e2[i] = c.v[i] - a.v[i]; }
    n[0] = e1[1] * e2[2] - e1[2] * e2[1];
    n[1] = e1[2] * e2[0] - e1[0] * e2[2];
    n[2] = e1[0] * e2[1] - e1[1] * e2[0];
    const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(len > 1e-12 * scale * scale)) return;  // degenerate: not clipping keeps a superset
    for (int i = 0; i < 3; ++i) n[i] /= len;
    n[3] = -(n[0] * a.v[0] + n[1] * a.v[1] + n[2] * a.v[2]);
    if (n[0] * cen[0] + n[1] * cen[1] + n[2] * cen[2] + n[3] < 0)
      for (int i = 0; i < 4; ++i) n[i] = -n[i];
    // every corner of the pyramid stays inside (fp slack, sides that are not exactly planar)
    double worst = 0;
    for (int c = 0; c < 4; ++c)
      for (int e = 0; e < 2; ++e)
        worst = std::min(worst, n[0] * F[c][e].v[0] + n[1] * F[c][e].v[1] + n[2] * F[c][e].v[2] + n[3]);
    n[3] += -worst + 1e-6 * scale;
    for (int i = 0; i < 4; ++i) planes[npl][i] = n[i];
    ++npl;
  };
  for (int c = 0; c < 4; ++c) add_plane(F[c][0], F[c][1], F[(c + 1) & 3][0]);
  add_plane(F[0][0], F[1][0], F[2][0]);
  add_plane(F[0][1], F[1][1], F[2][1]);

  const double eps = 1e-3;
  double lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = (double)P.lo[a] - eps; hi[a] = (double)P.hi[a] + eps; }
  double mn = 1e300, mx = -1e300;
  for (int c = 0; c < 4; ++c)
    for (int e = 0; e < 2; ++e) {
      const double *p = F[c][e].v;
      if (p[0] >= lo[0] && p[0] <= hi[0] && p[1] >= lo[1] && p[1] <= hi[1] && p[2] >= lo[2] && p[2] <= hi[2]) {
        mn = std::min(mn, p[as]);
        mx = std::max(mx, p[as]);
      }
    }
  for (int a = 0; a < 3; ++a)
    for (int side = 0; side < 2; ++side) {
      const int b = (a + 1) % 3, c = (a + 2) % 3;
      SlabVec poly[2][24];
      const double bb[4] = {lo[b], hi[b], hi[b], lo[b]}, cc[4] = {lo[c], lo[c], hi[c], hi[c]};
      for (int k = 0; k < 4; ++k) {
        poly[0][k].v[a] = side ? hi[a] : lo[a];
        poly[0][k].v[b] = bb[k];
        poly[0][k].v[c] = cc[k];
      }
      int n = 4, cur = 0;
      for (int k = 0; k < npl && n > 0; ++k) {
        n = slab_clip_polygon(poly[cur], n, planes[k], poly[cur ^ 1]);
        cur ^= 1;
      }
      for (int k = 0; k < n; ++k) {
        mn = std::min(mn, poly[cur][k].v[as]);
        mx = std::max(mx, poly[cur][k].v[as]);
      }
    }
  if (!(mn <= mx)) return false;
  *smin = std::max(mn, (double)P.lo[as]);
  *smax = std::min(mx, (double)P.hi[as]);
  return true;
}
}  // namespace
#endif  // SLAB_PART == 0

template <int DT, int SH, int PERM, int NW, int NL, bool DIAG, int TF = 1, bool BR = true, bool SHD = false>
static hipError_t launch_slab(const RenderParams &P, const SlabParams &Q, size_t lds, int nblocks, hipStream_t s) {
  auto k = smk_k_slab<DT, SH, PERM, NW, NL, DIAG, TF, BR, SHD>;
  static bool attr_set[64] = {};  // per device: the attribute belongs to the function ON the current device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(k, dim3(nblocks), dim3((NW + NL) * 64), lds, s, P, Q);
  return hipGetLastError();
}

// plan + launch; returns hipErrorNotSupported when the configuration must use the gather kernel

// ---- instance dispatch, one function per voxel type (one translation unit each)
hipError_t smk_slab_dispatch_u8(const RenderParams &P, const SlabParams &Q, int tf_mode, int shade_kind, int nw, int nl, bool diag, size_t lds,
                                int nblocks, const char **why, hipStream_t s);
hipError_t smk_slab_dispatch_f32(const RenderParams &P, const SlabParams &Q, int tf_mode, int shade_kind, int nw, int nl, bool diag, size_t lds,
                                 int nblocks, const char **why, hipStream_t s);
// ... and one for the eye pass of frames with shadows (SHD instances: 2-D / 3-D table, R8k shading or none, both voxel types)
hipError_t smk_slab_dispatch_shadow(const RenderParams &P, const SlabParams &Q, int dtype, int tf_mode, int shade_kind, int nw, int nl, size_t lds,
                                    int nblocks, const char **why, hipStream_t s);
#if SLAB_PART == 2
hipError_t smk_slab_dispatch_shadow(const RenderParams &P, const SlabParams &Q, int dtype, int tf_mode, int shade_kind, int nw, int nl, size_t lds,
                                    int nblocks, const char **why, hipStream_t s) {
#define GO(D, S, R, N, L)                                                                                  \
  if (dtype == D && shade_kind == S && Q.perm == R && nw == N && nl == L) {                              \
    if (tf_mode == 2) {                                                                                  \
      if (Q.bricks) return (launch_slab<D, S, R, N, L, false, 2, true, true>(P, Q, lds, nblocks, s));    \
      return (launch_slab<D, S, R, N, L, false, 2, false, true>(P, Q, lds, nblocks, s));                 \
    }                                                                                                    \
    if (tf_mode == 1) {                                                                                  \
      if (Q.bricks) return (launch_slab<D, S, R, N, L, false, 1, true, true>(P, Q, lds, nblocks, s));    \
      return (launch_slab<D, S, R, N, L, false, 1, false, true>(P, Q, lds, nblocks, s));                 \
    }                                                                                                    \
    *why = "shadows need a 2-D or 3-D table";                                                            \
    return hipErrorNotSupported;                                                                         \
  }
#define GO_NW(D, S, R) GO(D, S, R, 8, 2) GO(D, S, R, 10, 2) GO(D, S, R, 12, 4)
#define GO_R(D, S) GO_NW(D, S, 0) GO_NW(D, S, 1) GO_NW(D, S, 2)
  GO_R(0, 0) GO_R(0, 1) GO_R(1, 0) GO_R(1, 1)
#undef GO_R
#undef GO_NW
#undef GO
  *why = "no shadow instance for this configuration";
  return hipErrorNotSupported;
}
#else
#if SLAB_PART == 0
hipError_t smk_slab_dispatch_u8(const RenderParams &P, const SlabParams &Q, int tf_mode, int shade_kind, int nw, int nl, bool diag, size_t lds,
                                int nblocks, const char **why, hipStream_t s) {
  const int dtype = 0;
#else
hipError_t smk_slab_dispatch_f32(const RenderParams &P, const SlabParams &Q, int tf_mode, int shade_kind, int nw, int nl, bool diag, size_t lds,
                                 int nblocks, const char **why, hipStream_t s) {
  const int dtype = 1;
#endif
  (void)diag;
#define GO(D, S, R, N, L)                                                                              \
  if (dtype == D && shade_kind == S && Q.perm == R && nw == N && nl == L) {                          \
    if (tf_mode == 2) {                                                                              \
      if constexpr ((N == 8 && L == 2) || (N == 10 && L == 2) || (N == 12 && L == 4)) {              \
        if (Q.bricks) return (launch_slab<D, S, R, N, L, false, 2, true>(P, Q, lds, nblocks, s));  \
        return (launch_slab<D, S, R, N, L, false, 2, false>(P, Q, lds, nblocks, s));     \
      }                                                                                              \
      *why = "no dense-3-D-table instance for this tile size";                                       \
      return hipErrorNotSupported;                                                                   \
    }                                                                                                \
    if (tf_mode == 0) {                                                                              \
      if constexpr (S == 0 && ((N == 8 && L == 2) || (N == 10 && L == 2) || (N == 12 && L == 4)))    \
        return (launch_slab<D, S, R, N, L, false, 0, false>(P, Q, lds, nblocks, s));     \
      *why = "no colour-table instance for this tile size";                                          \
      return hipErrorNotSupported;                                                                   \
    }                                                                                                \
    if constexpr (D == 1 && S == 1) {                                                                \
      if (diag) return (launch_slab<D, S, R, N, L, true>(P, Q, lds, nblocks, s));        \
    }                                                                                                \
    if (Q.bricks) return (launch_slab<D, S, R, N, L, false, 1, true>(P, Q, lds, nblocks, s));  \
    return (launch_slab<D, S, R, N, L, false, 1, false>(P, Q, lds, nblocks, s));         \
  }
  // product tile shapes: 32x16 px with 8+2 waves, 32x24 px with 12+4; the others are experiment knobs (option "tile")
#ifdef SLAB_ALL_TILES
#define GO_NW(D, S, R) GO(D, S, R, 4, 1) GO(D, S, R, 6, 2) GO(D, S, R, 8, 1) GO(D, S, R, 8, 2) GO(D, S, R, 8, 4) GO(D, S, R, 9, 2) GO(D, S, R, 12, 2) GO(D, S, R, 12, 4) GO(D, S, R, 8, 8) GO(D, S, R, 10, 6) GO(D, S, R, 10, 2)
#else
#define GO_NW(D, S, R) GO(D, S, R, 8, 2) GO(D, S, R, 10, 2) GO(D, S, R, 12, 4)
#endif
#define GO_R(D, S) GO_NW(D, S, 0) GO_NW(D, S, 1) GO_NW(D, S, 2)
#if SLAB_PART == 0
#ifndef SLAB_FEW_INSTANCES
  GO_R(0, 0) GO_R(0, 1) GO_R(0, 2)
#endif
#else
#ifdef SLAB_FEW_INSTANCES
  GO_R(1, 1)
#else
  GO_R(1, 0) GO_R(1, 1) GO_R(1, 2)
#endif
#endif
#undef GO_R
#undef GO_NW
#undef GO
  *why = "no kernel instance for this tile size";
  return hipErrorNotSupported;
}
#endif  // SLAB_PART != 2

#if SLAB_PART == 0
hipError_t smk_launch_slab(RenderParams P, int dtype, int tf_mode, int shade_kind, int opt_T, int opt_tile, int forced,
                           const void *vox_native, const void *vox_xmajor, SlabAux *aux, const char **why,
                           hipStream_t s) {
  const int opt_fly = (opt_T >> 8) & 0xff;  // (developer knobs travel packed: slab_T | slab_fly << 8 | slab_ns << 16 | slab_sched << 24)
  const int opt_ns = (opt_T >> 16) & 0xff;
  const int opt_sched = (opt_T >> 24) & 0xf;  // (experiment knob: order of an XCD's tiles, see the schedule)
  const int opt_split = P.depth ? 1 : aux->opt_split;  // DEPTH SEGMENTS: 0 auto (measured long tiles), 1 off, 2.. every tile in that many (a depth output: off -- the merge pass knows colours only)
  opt_T &= 0xff;
  *why = nullptr;
  if (tf_mode < 0 || tf_mode > 2) { *why = "no classification mode"; return hipErrorNotSupported; }
  if (tf_mode == 0 && (!P.tlut || P.tlut_size < 1)) { *why = "no colour table"; return hipErrorNotSupported; }
  if (tf_mode == 0) shade_kind = 0;  // (the scalar renderer does not shade, VolumeRenderer.cpp:576-587)
  if (tf_mode == 2 && (!P.tf3d || P.s3v < 1 || P.s3g < 1 || P.s3h < 1)) { *why = "no 3-D table"; return hipErrorNotSupported; }
  if (P.pert_on) { *why = "perturbation"; return hipErrorNotSupported; }
  // (back-to-front frames -- VolumeRenderer.cpp:590 -- are composited FRONT TO BACK here: "over" is associative, the slices
  //  stream one way; what changes is the association of the blend -- a few ulp per sample, as with depth segments -- and a
  //  saturated ray may stop.  Option kernel = 1 renders them in the reference's own order.)
  if (P.N[0] < 2 || P.N[1] < 2 || P.N[2] < 2) { *why = "volume thinner than 2 voxels"; return hipErrorNotSupported; }
  if (dtype == 1 && !P.n_in_w) { *why = "4-channel f32 voxels"; return hipErrorNotSupported; }
  if (P.rc.nplanes <= 0) { *why = "no planes"; return hipErrorNotSupported; }
  if (P.sh.on) {
    // frames with shadows: the component of a ray along the slice normal is affine in the pixel coordinate; where it keeps
    // its sign over the viewport's corners no ray runs parallel to the slices (the planning divides by it)
    double lo_n = 1e300, hi_n = -1e300;
    for (int c = 0; c < 4; ++c) {
      const double px = ((c & 1) ? (double)P.W : 0.0) * P.rc.pxs + P.rc.pxl, py = ((c & 2) ? (double)P.H : 0.0) * P.rc.pys + P.rc.pyl;
      const double nD = px * P.sh.nDx + py * P.sh.nDy + P.sh.nDc;
      lo_n = std::min(lo_n, nD);
      hi_n = std::max(hi_n, nD);
    }
    if (!(lo_n > 0 || hi_n < 0) || std::min(fabs(lo_n), fabs(hi_n)) < 1e-6 * std::max(fabs(lo_n), fabs(hi_n))) {
      *why = "half-angle slices parallel to some eye ray";
      return hipErrorNotSupported;
    }
  }

  // principal axis from the central ray
  double Ac[3], Bc[3];
  host_ray(P, P.W / 2, P.H / 2, Ac, Bc);
  int as = 0;
  for (int a = 1; a < 3; ++a)
    if (fabs(Bc[a]) > fabs(Bc[as])) as = a;
  SlabParams Q;
  memset(&Q, 0, sizeof Q);
  Q.status = aux->h_status + aux->status_slot;
  Q.status_tag = aux->status_tag;
  Q.diag = aux->d_diag;
  Q.as = as;
  if (as == 2) { Q.perm = 0; Q.au = 0; Q.av = 1; }
  else if (as == 1) { Q.perm = 1; Q.au = 0; Q.av = 2; }
  else { Q.perm = 2; Q.au = 1; Q.av = 2; }
  if (Q.perm == 2 && !vox_xmajor) { *why = "x-major copy unavailable"; return hipErrorNotSupported; }
  Q.dir = Bc[as] > 0 ? 1 : -1;
  Q.Ou = P.O[Q.au]; Q.Ov = P.O[Q.av]; Q.Os = P.O[as];
  Q.Du = P.D[Q.au]; Q.Dv = P.D[Q.av]; Q.Ds = P.D[as];
  if (Q.perm == 0) { Q.strideV = P.D[0]; Q.strideS = (long long)P.D[0] * P.D[1]; Q.vox = vox_native; }
  else if (Q.perm == 1) { Q.strideV = (long long)P.D[0] * P.D[1]; Q.strideS = P.D[0]; Q.vox = vox_native; }
  else { Q.strideV = P.D[1]; Q.strideS = (long long)P.D[1] * P.D[2]; Q.vox = vox_xmajor; }  // [x][z][y]
  // (the free clip plane lives in the kernel's set-up: the kept samples of a ray are an interval of planes, see there; as a
  //  run-time test per sample in the consumers' loop it cost every frame WITHOUT a clip plane 4-5 % at 512^3 -- round 2)
  // an empty region (a clip plane outside a shard's box): the gather kernel's explicit comparisons
  // render it as nothing; the median-of-three membership test here needs lo <= hi
  for (int a = 0; a < 3; ++a)
    if (!(P.lo[a] <= P.hin[a])) { *why = "region is empty"; return hipErrorNotSupported; }
  if (Q.Ds > 4096) { *why = "more than 4096 slices"; return hipErrorNotSupported; }
  // u8 voxels are 8 B: the DMA moves 16-B units, so rows must start and end on even voxels
  if (dtype == 0 && ((Q.Du & 1) || (Q.strideV & 1) || (Q.strideS & 1))) { *why = "odd U extent for 8-byte voxels"; return hipErrorNotSupported; }

  // workgroup shape: consumer waves are 8x8 pixel sub-tiles; NL loader waves.
  //   light windows: 32x16 tile, 8+1 waves, two workgroups per CU
  //   heavy windows (1024^3 f32 at a voxel per pixel): 24x32 tile, 12+4 waves, one per CU; the
  //   tile is narrow along U so that a window row (tile + drift + pair) fits a 32-unit LDS pitch
  // (one loader wave moves ~10 B/cycle at best, MI355X_MICROARCH.md 'ldsdma-fill'; a heavy
  //  stream needs several per CU)
  struct Cfg { int tw, th, nl; };
  Cfg cfgs[3] = {{32, 16, 2}, {32, 24, 4}, {32, 24, 4}};
  int ncfg = 2;
  if (opt_tile == 1) { cfgs[0] = {16, 16, 1}; ncfg = 1; }
  else if (opt_tile == 2) { cfgs[0] = {24, 32, 4}; ncfg = 1; }
  else if (opt_tile == 3) { cfgs[0] = {24, 16, 2}; ncfg = 1; }
  else if (opt_tile == 4) { cfgs[0] = {32, 24, 4}; ncfg = 1; }
  else if (opt_tile == 5) { cfgs[0] = {32, 16, 2}; ncfg = 1; }
  else if (opt_tile == 6) { cfgs[0] = {32, 16, 1}; ncfg = 1; }
  else if (opt_tile == 7) { cfgs[0] = {16, 32, 2}; ncfg = 1; }
  else if (opt_tile == 8) { cfgs[0] = {32, 16, 4}; ncfg = 1; }
  else if (opt_tile == 9) { cfgs[0] = {16, 32, 1}; ncfg = 1; }
  else if (opt_tile == 10) { cfgs[0] = {24, 32, 2}; ncfg = 1; }
  else if (opt_tile == 11) { cfgs[0] = {24, 24, 2}; ncfg = 1; }
  else if (opt_tile == 12) { cfgs[0] = {48, 16, 4}; ncfg = 1; }
  else if (opt_tile == 13) { cfgs[0] = {16, 48, 4}; ncfg = 1; }
  else if (opt_tile == 14) { cfgs[0] = {32, 24, 2}; ncfg = 1; }
  // (round 2: MORE loader waves -- 10+6 on 40x16 / 16x40 px, 8+8 on 32x16 / 16x32 -- on the 1024^3 frame: 5.10 / 7.76 /
  //  6.30 / 7.21 ms against 4.33 for 12+4 on 32x24: the smaller tiles' extra fringe outweighs the issue slots)
  else if (opt_tile == 15) { cfgs[0] = {40, 16, 6}; ncfg = 1; }
  else if (opt_tile == 16) { cfgs[0] = {16, 40, 6}; ncfg = 1; }
  else if (opt_tile == 17) { cfgs[0] = {32, 16, 8}; ncfg = 1; }
  else if (opt_tile == 18) { cfgs[0] = {16, 32, 8}; ncfg = 1; }
  else if (opt_tile == 19) { cfgs[0] = {48, 16, 4}; ncfg = 1; }
  else if (opt_tile == 22) { cfgs[0] = {16, 48, 2}; ncfg = 1; }
  else if (opt_tile == 23) { cfgs[0] = {48, 16, 2}; ncfg = 1; }
  else if (opt_tile == 20) { cfgs[0] = {40, 16, 2}; ncfg = 1; }
  else if (opt_tile == 21) { cfgs[0] = {16, 40, 2}; ncfg = 1; }
  const int upv = dtype == 0 ? 2 : 1;
  // Small-workgroup shape: 10 + 2 waves on 16x40 or 40x16 pixels, or 8 + 2 on 32x16 -- whichever needs the fewest DMA
  // instructions per ray for THIS view (the window's width is rounded up to whole 128-byte units of LDS pitch, so the
  // answer depends on the pose: cfg 3's gives 7 / 640, 8 / 640 and 6 / 512 rays, and 1.45 / 1.57 / 1.55 ms).  Two
  // twelve-wave workgroups fill a CU's 24 wave slots at this kernel's 75-80 VGPRs; two ten-wave ones leave four idle.
  // A probing pass sizes the candidates' windows (each scan is kept, see below), the real pass plans the winner.
  static const bool dbg_time = getenv("SMK_DEBUG_TIME") != nullptr;  // (developer: where the host's planning time goes)
  static double dbg_t[4] = {0, 0, 0, 0};
  static int dbg_n = 0;
  auto dbg_now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
  const double dbg_t0 = dbg_time ? dbg_now() : 0;
  double dbg_scan = 0;
  struct DbgExit {
    bool on; double t0; double *acc; decltype(dbg_now) *now;
    ~DbgExit() { if (on) *acc += (*now)() - t0; }
  } dbg_exit{dbg_time, dbg_t0, &dbg_t[2], &dbg_now};
  // (a tie is won by the shape with the longer window rows, see the probe: same DMA count, rows twice as long; A/B on the cfg 3 frame turned
  //  25 / 30 / 33 / 36 / 40 degrees, five alternations each: 0.611 / 0.577 / 0.614 / 0.636 / 0.625 ms against the narrow
  //  shape's 0.606 / 0.614 / 0.627 / 0.654 / 0.641, tools/shape_ab.py)
  Cfg cand[3] = {{40, 16, 2}, {16, 40, 2}, {32, 16, 2}};
  const bool choose = opt_tile == 0 && SLAB_BIG_WAVES >= 12;
  int best = -1, best_wu = 0;
  double best_score = 1e300;
  // The choice is kept while the view keeps its principal axis, direction and sizes (re-examined every 64 frames): a
  // camera that moves every frame must not pay the probe -- nor flip between two shapes of nearly equal score, which
  // would throw away the measured schedule weights of the tiling each time.  The probe itself scans every fourth tile
  // row and column plus the borders (its answer only ranks the shapes; the real pass sizes the winner's window fully).
  struct ShapeKey { int as, dir, W, H, dtype, N[3]; float lo[3], hi[3]; } skey;
  memset(&skey, 0, sizeof skey);
  skey.as = as; skey.dir = Q.dir; skey.W = P.W; skey.H = P.H; skey.dtype = dtype;
  for (int a = 0; a < 3; ++a) { skey.N[a] = P.N[a]; skey.lo[a] = P.lo[a]; skey.hi[a] = P.hi[a]; }
  const bool shape_known = choose && aux->shape_key.size() == sizeof skey && !memcmp(aux->shape_key.data(), &skey, sizeof skey) &&
                           aux->shape_choice >= 0 && ++aux->shape_age < 64;
  if (shape_known) best = aux->shape_choice;
  for (int pass = (choose && !shape_known) ? 0 : 1; pass < 2; ++pass) {
  int alt = -1;  // the other twelve-wave shape: tried before the big workgroup where the chosen one turns out not to fit
  if (pass == 1 && best >= 0) {
    cfgs[0] = cand[best];
    // (the probe sizes windows from a sparse scan and can rank a shape by a window it will not get: near a pitch step the
    //  real window of the narrow tile is half as big again, its ring no longer fits half a CU -- a turning camera then
    //  rendered 60 frames in a row on the big workgroup, 0.72 ms, where the other small shape takes 0.63)
    alt = best == 0 ? 1 : 0;
    cfgs[1] = cand[alt];
    cfgs[2] = {32, 24, 4};
    ncfg = 3;
    if (!shape_known) {
      aux->shape_key.assign(reinterpret_cast<const unsigned char *>(&skey), reinterpret_cast<const unsigned char *>(&skey) + sizeof skey);
      aux->shape_choice = best;
      aux->shape_age = 0;
    }
  }
  const Cfg *list = pass == 0 ? cand : cfgs;
  const int nlist = pass == 0 ? 3 : ncfg;
  for (int ci = 0; ci < nlist; ++ci) {
    const int tw = list[ci].tw, th = list[ci].th, nl = list[ci].nl;
    const int nw = (tw / 8) * (th / 8);
    Q.tw = tw; Q.th = th;
    P.ntx = (P.W + tw - 1) / tw;
    P.nty = (P.H + th - 1) / th;
    P.tiles_per_xcd = (P.ntx * P.nty + 7) / 8;

    // every ray must advance along S in the same direction and not too obliquely; window bound:
    // bundle cross-section extent (corner rays of every tile) at the two S faces + drift over the
    // two-slice interval a window covers + texel pair + eps
    double max_eu = 0, max_ev = 0, max_drift_u = 0, max_drift_v = 0;
    std::vector<int> work((size_t)P.ntx * P.nty, 1);  // slices each tile streams (schedule weight)
    // The scan of all tiles below is a function of camera, region and tile shape alone; a frame of an unchanged
    // view reuses the last one (per tile ~0.3 us of double arithmetic: 0.65 ms for the 2048 tiles of a 1024^2
    // viewport -- hidden behind a 1.9 ms kernel, but not behind the 0.3 ms one of an eighth of the volume)
    struct ScanKey {
      smk_raycoef rc;
      int W, H, tw, th, as, au, av, dir, N[3], top[3];
      float lo[3], hi[3], hin[3];
      float sh[18];  // frames with shadows: the eye rays' coefficients (SmkShadowRays), else zeros
    } key;
    memset(&key, 0, sizeof key);
    key.rc = P.rc;
    if (P.sh.on) {
      const SmkShadowRays &h = P.sh;
      const float v[18] = {1.0f, h.Ec[0], h.Ec[1], h.Ec[2], h.Dc[0], h.Dc[1], h.Dc[2], h.Dx[0], h.Dx[1], h.Dx[2], h.Dy[0], h.Dy[1], h.Dy[2],
                           h.nDc, h.nDx, h.nDy, h.numA, h.dB};
      memcpy(key.sh, v, sizeof v);
    }
    key.W = P.W; key.H = P.H; key.tw = tw; key.th = th; key.as = as; key.au = Q.au; key.av = Q.av; key.dir = Q.dir;
    for (int a = 0; a < 3; ++a) { key.N[a] = P.N[a]; key.top[a] = P.top[a]; key.lo[a] = P.lo[a]; key.hi[a] = P.hi[a]; key.hin[a] = P.hin[a]; }
    int slot = -1;
    for (int k = 0; k < 4; ++k)
      if (aux->scan[k].key.size() == sizeof key && !memcmp(aux->scan[k].key.data(), &key, sizeof key) && aux->scan[k].work.size() == work.size()) slot = k;
    const bool scan_hit = slot >= 0;
    if (!scan_hit) slot = aux->scan_next++ & 3;
    SlabAux::Scan &scan = aux->scan[slot];
    const double dbg_s0 = dbg_time ? dbg_now() : 0;
    if (scan_hit) {
      max_eu = scan.v[0]; max_ev = scan.v[1]; max_drift_u = scan.v[2]; max_drift_v = scan.v[3];
      work = scan.work;
    } else
    {
      // The rows of tiles are scanned by a few host threads (a pool the context keeps): ~0.13 us of double arithmetic per
      // tile is 0.27 ms for the 2048 tiles of a 1024^2 viewport on one thread -- as much as a shard's whole kernel when the
      // camera moves every frame.  Each thread keeps its own maxima and its own refusal; rows write disjoint tiles.
      struct Part { double eu = 0, ev = 0, du = 0, dv = 0; const char *why = nullptr; };
      auto scan_rows = [&](int ty0, int ty1, Part &pt) {
        for (int tyi = ty0; tyi < ty1; ++tyi)
          for (int txi = 0; txi < P.ntx; ++txi) {
            if (pass == 0 && !(((txi & 3) == 0 || txi == P.ntx - 1) && ((tyi & 3) == 0 || tyi == P.nty - 1))) continue;  // (sparse probe)
            double cA[4][3], cB[4][3];
            for (int c = 0; c < 4; ++c) {
              int cx = std::min(txi * tw + ((c & 1) ? tw - 1 : 0), P.W - 1);
              int cy = std::min(tyi * th + ((c & 2) ? th - 1 : 0), P.H - 1);
              double *A = cA[c], *B = cB[c];
              host_ray(P, cx, cy, A, B);
              if (!(B[as] * Q.dir > 0) || fabs(B[as]) < 1e-12) { pt.why = "rays do not share a marching direction"; return; }
              double du = fabs(B[Q.au] / B[as]), dv = fabs(B[Q.av] / B[as]);
              // (3 voxels of drift per slice: close-ups with a wide frustum reach ~2.5 at the frame's edge and still
              //  run 2-3x faster here than on the gather kernel; the window bound below grows with the drift)
              if (du > 3.0 || dv > 3.0) { pt.why = "view too oblique for the principal axis"; return; }
              pt.du = std::max(pt.du, du);
              pt.dv = std::max(pt.dv, dv);
            }
            // the slices this tile can stream: S-extent of its ray bundle inside the region (exact for
            // the continuous bundle, see slab_bundle_slice_range)
            double smin_t, smax_t;
            if (!slab_bundle_slice_range(P, (double)(txi * tw), (double)(tyi * th), (double)std::min(txi * tw + tw, P.W),
                                         (double)std::min(tyi * th + th, P.H), as, &smin_t, &smax_t))
              continue;  // the bundle misses the region: nothing to stream
            work[(size_t)tyi * P.ntx + txi] = 16 + (int)(smax_t - smin_t);
            // cross-section of the bundle where THIS tile streams: it is linear in s (perspective), so
            // the two ends of the tile's own slice range bound it.  A sample at s reads slices floor(s)
            // and floor(s)+1, and the window of slice j covers s in [j-1, j+1] (stretched by half a
            // slice at the volume faces): 2.5 slices beyond the range.  (Bounding by the volume's S
            // faces instead costs 25-30 % window area at a voxel per pixel: rays are not inside the
            // volume where they are widest apart.)
            const double pad = 2.5 + 1e-2;
            const double se[2] = {std::max(-0.5, smin_t - pad), std::min((double)P.N[as] - 0.5, smax_t + pad)};
            for (int f = 0; f < 2; ++f) {
              double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
              for (int c = 0; c < 4; ++c) {
                double mm = (se[f] - cA[c][as]) / cB[c][as];
                double u = cA[c][Q.au] + cB[c][Q.au] * mm, v = cA[c][Q.av] + cB[c][Q.av] * mm;
                umin = std::min(umin, u); umax = std::max(umax, u);
                vmin = std::min(vmin, v); vmax = std::max(vmax, v);
              }
              pt.eu = std::max(pt.eu, umax - umin);
              pt.ev = std::max(pt.ev, vmax - vmin);
            }
          }
      };
      const int nthreads = (pass == 1 && P.nty >= 16 && P.ntx * P.nty >= 512) ? smk_host_pool_size() : 1;
      std::vector<Part> parts((size_t)std::max(nthreads, 1));
      if (nthreads <= 1) {
        scan_rows(0, P.nty, parts[0]);
      } else {
        smk_host_pool_run(nthreads, [&](int k) { scan_rows((int)((long long)P.nty * k / nthreads), (int)((long long)P.nty * (k + 1) / nthreads), parts[(size_t)k]); });
      }
      for (const Part &pt : parts) {
        if (pt.why) { *why = pt.why; return hipErrorNotSupported; }
        max_eu = std::max(max_eu, pt.eu); max_ev = std::max(max_ev, pt.ev);
        max_drift_u = std::max(max_drift_u, pt.du); max_drift_v = std::max(max_drift_v, pt.dv);
      }
    }
    if (dbg_time) dbg_scan += dbg_now() - dbg_s0;
    if (!scan_hit && pass == 1) {  // (a scan that bailed out above returned, a probe is sparse: only complete ones are kept)
      scan.key.assign(reinterpret_cast<const unsigned char *>(&key), reinterpret_cast<const unsigned char *>(&key) + sizeof key);
      scan.v[0] = max_eu; scan.v[1] = max_ev; scan.v[2] = max_drift_u; scan.v[3] = max_drift_v;
      scan.work = work;
    }
    // a window spans s in [j-1, j+1] (2 slices of drift; 2.5 at a face; 3 where slice 1 of a
    // three-slice volume touches both); a coordinate range of extent e touches at most ceil(e) + 2
    // texels (pair included); eps for the fp32 chains
    const double span = P.N[as] <= 3 ? 3.0 : 2.5;
    int Wu = (int)ceil(max_eu + span * max_drift_u + 2 * SLAB_EPS) + 2;
    int Wv = (int)ceil(max_ev + span * max_drift_v + 2 * SLAB_EPS) + 2;
    if (dtype == 0) Wu = ((Wu + 1) & ~1) + 2;  // even width, even alignment of the origin
    Wu = std::min(Wu, Q.Du);
    Wv = std::min(Wv, Q.Dv);
    if (Wu < 2 || Wv < 2) { *why = "degenerate window"; return hipErrorNotSupported; }
    // fixed window shape: wu 16-byte units per row on an LDS pitch of the next multiple of 8 units
    // (128 B); the slot image is flat, so the (row, column) a DMA lane serves repeats every
    // per = wp / gcd(64, wp) chunks = rpg = 64 / gcd(64, wp) rows ("group")
    Q.wu = Wu / upv;
    Q.wv = Wv;
    if (Q.wu > 64) { if (ci + 1 < nlist || pass == 0) continue; *why = "window wider than one DMA chunk"; return hipErrorNotSupported; }
    if (Q.Du > 2047 || Q.Dv > 2047) { *why = "stored box wider than 2047 voxels across the view"; return hipErrorNotSupported; }
    // The pitch: the next multiple of 8 units -- or, where that gives fewer DMA instructions per slice, the next multiple
    // of 4 whose period is one the loaders know (per = 1, 3, 5, 7: pitches 12, 20, 28): a 17-unit row on a pitch of 20
    // (two groups of 16 rows x 5 chunks) instead of 24 (four groups of 8 rows x 3 chunks) is 10 instructions for 12, and a
    // ring that fits half a CU again.
    {
      auto shape = [&](int wp, int &per, int &rpg, int &groups) {
        int g = 64, r = wp;
        while (r) { int t = g % r; g = r; r = t; }  // gcd(64, wp)
        per = wp / g;
        rpg = 64 / g;
        groups = (Q.wv + rpg - 1) / rpg;
      };
      int wp8 = (Q.wu + 7) & ~7, per8, rpg8, gr8;
      shape(wp8, per8, rpg8, gr8);
      Q.wp = wp8; Q.per = per8; Q.rpg = rpg8;
      const int wp4 = (Q.wu + 3) & ~3;
      static const bool only8 = getenv("SMK_PITCH8") != nullptr;  // (developer: the pitches of rounds 1-3, multiples of 8)
      if (wp4 != wp8 && !only8) {
        int per4, rpg4, gr4;
        shape(wp4, per4, rpg4, gr4);
        if (per4 <= 7 && gr4 * per4 < gr8 * per8) { Q.wp = wp4; Q.per = per4; Q.rpg = rpg4; }
      }
    }
    Q.groups = (Q.wv + Q.rpg - 1) / Q.rpg;
    // small workgroups: a window of whole row groups (its LDS image is that big anyway), see the loader's group loop
    if ((nw + nl) <= SLAB_BIG_WAVES && Q.groups * Q.rpg <= Q.Dv) Q.wv = Q.groups * Q.rpg;
    Q.chunks = Q.groups * Q.per;
    // (a kept shape is probed again the moment its own window changes size -- a narrow tile's window row crosses a pitch
    //  step of 8 units within a degree or two of a turning camera, its DMA count jumps by half, its ring no longer fits
    //  half a CU and the frames fall to the big workgroup: 0.72 ms where the other small shape takes 0.63 -- not only
    //  every 64 frames)
    if (pass == 1 && choose && best >= 0 && ci == 0) {
      if (shape_known && aux->shape_chunks != Q.chunks) aux->shape_age = 64;
      if (!shape_known) aux->shape_chunks = Q.chunks;
    }
    if (pass == 0) {  // probing: DMA instructions per ray
      // (the ten-wave shape leaves four of a CU's wave slots idle: with the brick flags on it measures 0.83 ms on the cfg 3
      //  frame where the twelve-wave shapes take 0.59-0.61, although it needs the fewest DMA instructions per ray at some
      //  poses -- a camera turning through such a pose got 0.79 ms frames for 0.63.  It has to win by 40 % now.)
      const double score = (double)Q.chunks / (nw * 64) * (nw + nl < 12 ? 1.4 : 1.0);
      // (a tie goes to the shape with the LONGER window rows -- the same DMA count in fewer, longer runs of memory: on the
      //  cfg 3 frame the wide shape, whose rows lie along the image's x there; a view turned a quarter about its axis has
      //  them along y)
      if (score < best_score || (score == best_score && Q.wu > best_wu)) { best_score = score; best = ci; best_wu = Q.wu; }
      continue;
    }
    Q.slot_bytes = Q.chunks * 1024;
    // per-slice extents (and with them the table-occupancy bitmap) from four chunks per slice up:
    // re-measured with two slices in flight, 512^3 f32 1.66 -> 1.58 ms, 512^3 u8 1.72 -> 1.68,
    // 256^3 at 1024^2 1.72 -> 1.55 (the first threshold, 12 chunks, dated from five slices in flight)
    Q.mask_need = Q.chunks >= 4 ? 1 : 0;
    // loaders of one slice (see the kernel: NLG groups of LPG loaders)
    const int nlg = !SLAB_ALT ? 1 : ((nw + nl) > SLAB_BIG_WAVES ? ((nl % SLAB_BIG_NLG) == 0 ? SLAB_BIG_NLG : 1) : nl);
    const int lpg = nl / nlg;
    if ((Q.groups + lpg - 1) / lpg * Q.per > 63) { if (ci + 1 < nlist) continue; *why = "window needs more than 63 DMA chunks per loader"; return hipErrorNotSupported; }
    // light enough for this configuration?  otherwise try the next (heavier-duty) one
    if (ci + 1 < nlist && (double)Q.chunks * 1024.0 / (nw * 64) > 16.0 * nl) {
      if (pass == 1 && alt >= 0 && ci == 0) ci = 1;  // (a stream this heavy is too heavy for the other small shape as well: the big one next)
      continue;
    }

    Q.use_ah = (tf_mode == 1 && P.third_axis && P.nelts <= 3 && P.sv >= 2 && P.sv <= 2048) ? 1 : 0;
    if (tf_mode == 1 && (P.sv < 2 || P.sg < 2)) { *why = "transfer function smaller than 2x2"; return hipErrorNotSupported; }
    const size_t occ_bytes = tf_mode == 1 ? (size_t)P.occ_roww * P.sg * 4 : tf_mode == 2 ? (size_t)P.occ_roww * P.s3g * 4 : 0;
    Q.fast_tf = (tf_mode == 1 && (!P.third_axis || Q.use_ah)) ? 1 : 0;
    // (measured: 5.99 -> 5.61 ms on 1024^3, where the texel gathers share the texture path with a
    //  heavy stream; no gain at 512^3, where the 8 KB are worth more as ring slots)
    Q.use_occ = ((Q.fast_tf || tf_mode == 2) && Q.mask_need && P.tf_occ && occ_bytes > 0 && occ_bytes <= 8192) ? 1 : 0;
    {  // brick flags (EMPTY LAYERS in the kernel): model-axis strides -> the kernel's (U, V, S)
      Q.bricks = (tf_mode == 1 || tf_mode == 2) ? P.bricks : nullptr;
      const int bst[3] = {1, P.nbr[0], P.nbr[0] * P.nbr[1]};
      Q.bsu = bst[Q.au];
      Q.bsv = bst[Q.av];
      Q.bss = bst[Q.as];
    }
    const size_t fixed = (size_t)Q.Ds * sizeof(SlabEnt) + (8 + 32) * 4 + 64 + (Q.use_ah ? (size_t)P.sv * 4 : 0) + (Q.use_occ ? occ_bytes : 0);
    // ring: as many slots as fit two workgroups per CU (small tiles) or one (big tiles)
    size_t budget = (nw + nl) > SLAB_BIG_WAVES ? 158 * 1024 : 78 * 1024;
    if (budget <= fixed) { *why = "slice table does not fit LDS"; return hipErrorNotSupported; }
    int ns = (int)((budget - fixed) / (size_t)Q.slot_bytes);
    if (ns > 24) ns = 24;
    // a wave holds ceil(slices per plane) + 1 slices while it works and the loaders want a few in flight
    const int band = (int)ceil(fabs(Bc[as])) + 2;
    if (ns < band + 2 && 158 * 1024 > fixed) {
      // the ring of a small workgroup does not fit half a CU: one workgroup per CU it is -- then
      // rather the big tile with 16 waves than this one with 10
      if (ci + 1 < nlist && budget < 158 * 1024) continue;
      ns = (int)((158 * 1024 - fixed) / (size_t)Q.slot_bytes);
      if (ns > band + 4) ns = band + 4;
    }
    if (opt_ns >= 3 && ns > opt_ns) ns = opt_ns;  // (experiment knob: cap the ring)
    if (ns < 3) { if (ci + 1 < nlist) continue; *why = "window does not fit LDS"; return hipErrorNotSupported; }
    Q.nslots = ns;
    if (pass == 1 && choose && alt >= 0 && ci == 1) {  // the alternative shape it is: kept from the next frame on
      aux->shape_choice = alt;
      aux->shape_chunks = Q.chunks;
      aux->shape_age = 0;
    }
    // (the set-up keeps a 16-byte brick mask per layer of bricks in the ring's memory before the stream starts)
    if (Q.bricks && (size_t)ns * Q.slot_bytes < ((size_t)((Q.Ds - 1) >> SMK_BRICK_LOG2) + 1) * 16) Q.bricks = nullptr;
    const int mych = (Q.groups + lpg - 1) / lpg * Q.per;  // most DMA instructions one loader issues per slice
    // Slices a loader keeps in flight.  TWO: a loader publishes a slice as landed only when it stops
    // issuing and waits for the oldest one, so a deep issue window delays every consumer that polls
    // for that slice -- and two slices per loader already cover the memory latency (4 loaders x 2 x
    // ~7 KiB per CU).  Measured (frames per setting 5/4/3/2/1 on a 5-slot ring): 1024^3 f32 4.27 /
    // 4.19 / 4.03 / 3.84 / 4.36 ms; 512^3 f32 (12 slots) 1.81 at 12, 1.66 at 2-4, 1.80 at 1; 1024^3 u8
    // 3.54 -> 2.94 ms.
    // (small workgroups since their loaders take whole slices in turn: ONE slice in flight per loader -- the other
    //  loader's is in flight beside it; 1 / 2: cfg 3 1.42-1.44 / 1.45-1.46 ms, other poses and a 256^3 frame -0.3 to -3 %)
    Q.maxfly = std::max(1, std::min(std::min(ns, 63 / mych + 1), opt_fly > 0 ? opt_fly : (nlg > 1 && (nw + nl) <= SLAB_BIG_WAVES ? 1 : 2)));
    // a deep ring lets the whole band step together (every lane active); on a short one a wave that
    // waits for its whole band leaves the loaders nothing to overlap with (measured, 1024^3: 5 slots,
    // wstep 0 / 1 / 2 -> 5.5 / 5.8 / 6.8 ms)
    // (re-measured with two slices in flight per loader, 5 slots: wstep 0 / 1 / 2 -> 3.80 / 3.68 / 4.68 ms)
    Q.wstep = std::max(0, std::min((int)ceil(fabs(Bc[as])), ns - 4));
    if (opt_T > 0) Q.wstep = std::max(0, std::min(opt_T - 1, ns - 3));  // (experiment knob: slab_T = wstep + 1)
    // (small workgroups, 10 slots against a band of 4: every other turn 1.87 ms, every turn 1.90, every 4th / 8th 2.0 / 2.2)
    Q.pmask = ns >= 2 * band ? 1 : 0;
    const size_t lds = (size_t)ns * Q.slot_bytes + fixed;
    if (getenv("SMK_DEBUG"))
      fprintf(stderr, "[smk] slice-ring plan: tile %dx%d, %d+%d waves, window %d units x %d rows (pitch %d units), %d chunks/slice, %d slots of %d B, table+ctl %zu B, LDS %zu B, band %d, wstep %d, pmask %d, maxfly %d\n",
              tw, th, nw, nl, Q.wu, Q.wv, Q.wp, Q.chunks, ns, Q.slot_bytes, fixed, lds, band, Q.wstep, Q.pmask, Q.maxfly);
    // ---- schedule.  Blocks are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8, in
    // order of b), each XCD with its own L2.  Every XCD gets one contiguous run of image tiles
    // (row-major: neighbours that walk neighbouring voxels share an L2) cut so that all runs
    // stream about the same number of slices -- tiles at the image border cross less of the
    // volume than central ones -- and starts its long tiles first (shortest tail).
    int nblocks = 0;
    long long ticks_sig_now = 0;
    int ticks_n_now = 0;
    {
      const int nt = P.ntx * P.nty;
      // measured weights: the previous frame's per-tile workgroup durations, when they are of this
      // very tiling and marching direction.  (The geometric estimate above -- slices streamed --
      // misses what consumers cost: on 1024^3 the XCDs holding the image's top and bottom rows ran
      // 1.6x longer per slice than the central ones and the frame waited for them.)
      const long long tsig = (((long long)P.ntx * 4096 + P.nty) * 64 + tw) * 64 + th + ((long long)(Q.perm * 2 + (Q.dir > 0)) << 48) +
                             ((long long)nw << 52) + ((long long)dtype << 56);
      ++aux->ticks_age;
      if (aux->ticks_pending && hipEventQuery(aux->ticks_ev) == hipSuccess) {  // a copy has come back
        // (adopted at once for a new tiling, refined after 4, 8 and 16 frames -- a new order changes
        //  who runs beside whom and with it the durations -- then every 32 frames: durations of a
        //  steady view barely move, and every new table is an upload and a host touch of the stream)
        const bool fresh = aux->ticks_good_sig != aux->ticks_pending_sig;
        const int due = aux->ticks_adopted < 3 ? (4 << aux->ticks_adopted) : 32;
        if (fresh || aux->ticks_age >= due) {
          if (fresh || (int)aux->ticks_good.size() != aux->ticks_pending_n) {
            aux->ticks_good.assign(aux->h_ticks, aux->h_ticks + aux->ticks_pending_n);
            aux->ticks_adopted = 0;
          } else {
            // damped: half the old weight, half the new measurement.  (Workgroups of a few tens of microseconds -- an
            // opaque table -- measure mostly who ran beside them: averaged, the schedule oscillated between two plans,
            // 0.11 and 0.20 ms for the same frame; there the SHORTEST duration seen is the tile's own cost.)
            unsigned longest_new = 0;
            for (int t = 0; t < aux->ticks_pending_n; ++t) longest_new = std::max(longest_new, aux->h_ticks[t]);
            for (int t = 0; t < aux->ticks_pending_n; ++t)
              aux->ticks_good[t] = longest_new < 10000u ? (aux->h_ticks[t] ? std::min(std::max(aux->ticks_good[t], 1u), aux->h_ticks[t]) : aux->ticks_good[t])
                                                        : (aux->ticks_good[t] + aux->h_ticks[t] + 1) / 2;
            ++aux->ticks_adopted;
          }
          aux->ticks_good_sig = aux->ticks_pending_sig;
          aux->ticks_age = 0;
          // the pieces' own durations, and the cuts they were measured under (DEPTH SEGMENTS: the next cuts come from them)
          if (aux->h_pticks && (int)aux->cuts_pending.size() == aux->ticks_pending_n * 10) {
            aux->pticks_good.assign(aux->h_pticks, aux->h_pticks + (size_t)aux->ticks_pending_n * 8);
            aux->cuts_good = aux->cuts_pending;
          }
          aux->recut = true;
        }
        aux->ticks_pending = false;
      }
      (void)hipGetLastError();
      if (aux->ticks_good_sig == tsig && (int)aux->ticks_good.size() == nt) {
        for (int t = 0; t < nt; ++t)
          if (aux->ticks_good[t] > 0) work[t] = (int)std::min<unsigned>(aux->ticks_good[t], 1u << 30);
      }
      if (nt > aux->ticks_cap) {
        if (aux->ticks_pending) (void)hipEventSynchronize(aux->ticks_ev);
        aux->ticks_pending = false;
        if (aux->d_ticks) (void)hipFree(aux->d_ticks);
        if (aux->h_ticks) (void)hipHostFree(aux->h_ticks);
        aux->d_ticks = nullptr;
        aux->h_ticks = nullptr;
        aux->ticks_cap = 0;
        if (aux->d_pticks) (void)hipFree(aux->d_pticks);
        if (aux->h_pticks) (void)hipHostFree(aux->h_pticks);
        aux->d_pticks = nullptr;
        aux->h_pticks = nullptr;
        hipError_t e = hipMalloc((void **)&aux->d_ticks, (size_t)nt * 20);
        if (e != hipSuccess) return e;
        e = hipMemset(aux->d_ticks, 0, (size_t)nt * 20);
        if (e != hipSuccess) return e;
        e = hipHostMalloc((void **)&aux->h_ticks, (size_t)nt * 4, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        e = hipMalloc((void **)&aux->d_pticks, (size_t)nt * 32);
        if (e != hipSuccess) return e;
        e = hipMemset(aux->d_pticks, 0, (size_t)nt * 32);
        if (e != hipSuccess) return e;
        e = hipHostMalloc((void **)&aux->h_pticks, (size_t)nt * 32, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        aux->ticks_cap = nt;
      }
      if (!aux->ticks_ev) {
        hipError_t e = hipEventCreateWithFlags(&aux->ticks_ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
      }
      Q.tile_ticks = aux->d_ticks;
      Q.piece_ticks = aux->d_pticks;
      Q.ntiles = nt;
      aux->ticks_n_last = nt;
      ticks_sig_now = tsig;
      ticks_n_now = nt;
      const int slots = (nw + nl) | (opt_sched << 8) | (opt_split << 12);  // (part of the cached plan.s key)
      std::vector<int2> order;
      // DEPTH SEGMENTS: how many workgroups render each tile.  From MEASURED durations only (the geometric estimate says
      // nothing about what a tile's samples cost): a tile longer than half the mean load of a workgroup slot is cut so that
      // no piece is; tiles under 60 us are never cut (a segment costs its own set-up, ~13 us).  Small scenes therefore run
      // unsplit, bit-identical to the gather kernel; option "slab_split" 1 turns it off, 2.. forces that many everywhere.
      // cuts[t] = {pieces K, cut_0 = 0, ..., cut_K = 255}: the tile's slice positions in 255ths
      const bool measured = aux->ticks_good_sig == tsig && (int)aux->ticks_good.size() == nt;
      if ((int)aux->cuts.size() != nt * 10 || aux->cuts_split != opt_split || aux->cuts_sig != tsig) {
        aux->cuts.assign((size_t)nt * 10, 0);
        for (int t = 0; t < nt; ++t) { aux->cuts[(size_t)t * 10] = 1; aux->cuts[(size_t)t * 10 + 2] = 255; }
        aux->cuts_split = opt_split;
        aux->cuts_sig = tsig;
        aux->recut = true;
      }
      if (aux->recut) {
        aux->recut = false;
        auto equal_cuts = [&](int t, int K) {
          unsigned char *c = &aux->cuts[(size_t)t * 10];
          c[0] = (unsigned char)K;
          for (int k = 0; k <= K; ++k) c[1 + k] = (unsigned char)(255 * k / K);
        };
        if (opt_split >= 2) {
          for (int t = 0; t < nt; ++t) equal_cuts(t, std::min(opt_split, 8));
        } else if (opt_split == 0 && measured) {
          long long total = 0;
          for (int t = 0; t < nt; ++t) total += work[t];
          const double wg_slots = 256.0 * ((nw + nl) > SLAB_BIG_WAVES ? 1 : 2);
          // a piece should take about half the mean load of a workgroup slot, never under 60 us (a piece costs its own
          // set-up, ~13 us); frames whose slots carry under 50 us each -- small scenes -- are not cut at all
          // ... and only frames whose longest tile stands well above the mean load of a slot: where the slots' summed load is
          // the bound (cfg 3 on one GPU: longest tile 0.52 ms, mean load 0.49 ms, frame 0.61 ms with every tile cut in two --
          // 0.62 uncut) pieces only add their set-up; on a shard of 1/8 of that volume (longest 0.24, mean 0.08) they are the gain
          const double mean_load = (double)total / wg_slots;
          double longest_tile = 0;
          for (int t = 0; t < nt; ++t) longest_tile = std::max(longest_tile, (double)work[t]);
          const double piece = std::max(0.75 * mean_load, 6000.0);  // 100 MHz ticks
          bool big_frame = mean_load >= 5000.0 && longest_tile > 1.5 * mean_load;
          if (aux->cuts_engaged && mean_load >= 5000.0 && longest_tile > 1.2 * mean_load) big_frame = true;  // (hysteresis)
          aux->cuts_engaged = big_frame;
          const bool have_pieces = (int)aux->pticks_good.size() == nt * 8 && (int)aux->cuts_good.size() == nt * 10;
          for (int t = 0; t < nt; ++t) {
            unsigned char *c = &aux->cuts[(size_t)t * 10];
            int Kw = (big_frame && (double)work[t] > 1.25 * piece) ? (int)std::min(8.0, ceil((double)work[t] / piece)) : 1;
            if (big_frame && c[0] >= 2 && Kw >= 1 && abs(Kw - (int)c[0]) <= 1 && (double)work[t] > piece) Kw = c[0];  // (a tile keeps its count while the wish is a neighbour of it)
            if (Kw == 1) { equal_cuts(t, 1); continue; }
            // the pieces this tile was last measured in: work per 255th of depth, piecewise constant
            const unsigned char *g = have_pieces ? &aux->cuts_good[(size_t)t * 10] : nullptr;
            const int Kg = g ? g[0] : 1;
            if (!g || Kg < 2) {
              if (c[0] != Kw) equal_cuts(t, Kw);
              continue;
            }
            double d[8], tot = 0, longest = 0;
            for (int k = 0; k < Kg; ++k) {
              d[k] = std::max(1.0, (double)aux->pticks_good[(size_t)t * 8 + k] - 1300.0);  // (less the piece's own set-up)
              tot += d[k];
              longest = std::max(longest, d[k]);
            }
            const bool same = !memcmp(g, c, 10);
            if (same && Kg == Kw && longest <= 1.3 * tot / Kg) continue;  // balanced enough: keep (no flip-flopping)
            // new cuts: equal shares of the measured cumulative work
            unsigned char nc[10] = {(unsigned char)Kw, 0};
            int k = 0;
            double acc = 0;  // work before piece k
            for (int j = 1; j < Kw; ++j) {
              const double want = tot * j / Kw;
              while (k < Kg - 1 && acc + d[k] < want) acc += d[k++];
              const double f = d[k] > 0 ? (want - acc) / d[k] : 0.5;
              int x = (int)lround(g[1 + k] + f * (g[2 + k] - g[1 + k]));
              x = std::max(x, (int)nc[j] + 1);
              x = std::min(x, 255 - (Kw - j));
              nc[1 + j] = (unsigned char)x;
            }
            nc[1 + Kw] = 255;
            memcpy(c, nc, 10);
          }
        } else {
          for (int t = 0; t < nt; ++t) equal_cuts(t, 1);
        }
      }
      std::vector<unsigned char> ksplit((size_t)nt, 1);
      for (int t = 0; t < nt; ++t) ksplit[t] = aux->cuts[(size_t)t * 10];
      // a piece's weight: its own measured duration when it was measured under these very cuts, else an equal share
      auto piece_weight = [&](int t, int k) -> int {
        const unsigned char *c = &aux->cuts[(size_t)t * 10];
        if ((int)aux->pticks_good.size() == nt * 8 && (int)aux->cuts_good.size() == nt * 10 && !memcmp(&aux->cuts_good[(size_t)t * 10], c, 10) &&
            aux->pticks_good[(size_t)t * 8 + k] > 0)
          return (int)std::min<unsigned>(aux->pticks_good[(size_t)t * 8 + k], 1u << 30);
        return work[t] / std::max<int>(c[0], 1);
      };
      if (aux->plan_slots == slots && aux->plan_work == work && aux->plan_cuts == aux->cuts && !aux->plan_order.empty()) {
        order = aux->plan_order;  // same weights, same schedule (planning stays off the per-frame path)
      } else {
        long long total = 0;
        for (int t = 0; t < nt; ++t) total += work[t];
        // The sequence the runs are cut from: tile rows interleaved top half / bottom half (row 0, row
        // h, row 1, row h+1, ...), so that every XCD holds TWO bands of neighbouring rows, one nearer
        // the image border and one nearer its centre.  An XCD runs 32 (or 64) workgroups at a time,
        // longest first; one band of the image centre is ~130 equally long workgroups -- four full
        // rounds and three stragglers in a fifth -- while a mixed run ends on short workgroups that
        // fill the last round.  Bands per XCD 1 / 2 / 3 / 4 / 6: 1024^3 f32 3.65 / 3.51 / 3.58 / 3.60 /
        // 3.56 ms, 1024^3 u8 3.01 / 2.90 / - / 2.96; 512^3 within noise (1.52-1.60).  More bands mix
        // better but share less of their windows' overlap in the XCD's L2.
        std::vector<int> seq;
        seq.reserve((size_t)nt);
        {
          const int F = 2, h = (P.nty + F - 1) / F;
          for (int r = 0; r < h; ++r)
            for (int f = 0; f < F; ++f)
              if (r + f * h < P.nty)
                for (int c = 0; c < P.ntx; ++c) seq.push_back((r + f * h) * P.ntx + c);
        }
        // runs = consecutive tiles of that sequence [cut[x], cut[x+1]): equal accumulated weight
        int cut[9];
        {
          long long acc = 0;
          int x = 0;
          cut[0] = 0;
          for (int t = 0; t < nt; ++t) {
            const int want = (int)std::min<long long>(7, (acc + work[seq[t]] / 2) * 8 / std::max<long long>(total, 1));
            while (x < want) cut[++x] = t;
            acc += work[seq[t]];
          }
          while (x < 8) cut[++x] = nt;
        }
        // (Measured and dropped: refining the cuts against a simulated list schedule of each XCD -- an
        //  XCD runs 32 or 64 workgroups at a time, longest first, so with 4-7 workgroups per slot its
        //  finishing time comes in steps and equal weight can leave three stragglers after the last
        //  full round.  A local search over the cuts won 2 % on 1024^3 and cost tens of milliseconds
        //  of host time whenever new weights arrived.)
        std::vector<std::vector<int>> run(8);
        size_t longest = 0;
        const bool dealt = opt_sched == 0;
        if (dealt) {
          // DEALT (round 3; the runs above stay as option slab_sched 5): tiles in order of falling weight, each to the XCD
          // that carries the least so far -- every XCD gets the same mix of long and short workgroups.  The dispatcher hands
          // blocks out in index order, block b to XCD b % 8, and a block whose XCD has no free slot holds back every block
          // behind it: the XCDs' lists advance in step, entry k of all eight together.  With a contiguous run of the image
          // per XCD the lists differ (166-240 tiles, the centre's runs hold more long tiles than an XCD has slots: two of
          // them must share a slot) and slots stood idle for a mean 10 us per turnover while work was pending elsewhere
          // (tools/timeline.py).  Dealt: cfg 3 0.609 -> 0.587 ms, the 1024^3 frame 1.176 -> 1.109; with every slice streamed
          // (equal tiles) no change.
          // What is dealt is a BLOCK of neighbouring tiles, not a single tile: neighbours
          // weigh about the same, so they also sit next to each other in their XCD's list and stream the same slices at
          // about the same time -- the window fringes they share are then fetched once per block and hit in that XCD's L2.
          // Single tiles scatter every tile's neighbours over the other seven XCDs: HBM-side traffic of the cfg 3 frame
          // 1.18 -> 1.46 GB (2 x 2 blocks: 1.31), of the north star with every slice streamed 22.4 -> 24.0 GB (4 x 4: 22.2).
          // Block size: 2 x 2 tiles where the tiles' weights differ (brick flags on: bigger blocks deal the work coarser --
          // measured 4 x 4: cfg 3 0.595 -> 0.604 ms, the 1024^3 frame 1.110 -> 1.136), 4 x 4 where they are nearly equal
          // (every slice streamed: 4.035 -> 4.01 ms and the traffic of the contiguous runs, 22.2 GB).
          std::vector<int> ws;
          ws.reserve((size_t)nt);
          for (int t = 0; t < nt; ++t)
            if (work[t] > 0) ws.push_back(work[t]);
          std::sort(ws.begin(), ws.end());
          // "nearly equal": the heavier half of the tiles within 1.25 x of one another (95th percentile against the median;
          // the tiles along the volume's silhouette are short whatever the table)
          const bool even = ws.size() >= 16 && (long long)ws[ws.size() * 95 / 100] * 4 <= (long long)ws[ws.size() / 2] * 5;
          const int SLAB_DEAL_W = even ? 4 : 2, SLAB_DEAL_H = even ? 4 : 2;
          const int gbx = (P.ntx + SLAB_DEAL_W - 1) / SLAB_DEAL_W, gby = (P.nty + SLAB_DEAL_H - 1) / SLAB_DEAL_H;
          std::vector<long long> gw((size_t)gbx * gby, 0);
          for (int t = 0; t < nt; ++t) gw[(size_t)((t / P.ntx) / SLAB_DEAL_H) * gbx + (t % P.ntx) / SLAB_DEAL_W] += work[t];
          std::vector<int> idx((size_t)gbx * gby);
          for (int g = 0; g < gbx * gby; ++g) idx[g] = g;
          std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return gw[a] > gw[b]; });
          long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int g : idx) {
            int x = 0;
            for (int k = 1; k < 8; ++k)
              if (load[k] < load[x]) x = k;
            const int gy = g / gbx, gx = g - gy * gbx;
            for (int dy = 0; dy < SLAB_DEAL_H; ++dy)
              for (int dx = 0; dx < SLAB_DEAL_W; ++dx) {
                const int ty = gy * SLAB_DEAL_H + dy, tx = gx * SLAB_DEAL_W + dx;
                if (ty < P.nty && tx < P.ntx) run[x].push_back(ty * P.ntx + tx);
              }
            load[x] += gw[g];
          }
        }
        for (int x = 0; x < 8; ++x) {
          if (!dealt)
            for (int t = cut[x]; t < cut[x + 1]; ++t) run[x].push_back(seq[t]);
          if (opt_sched == 0 || opt_sched == 5) {
            // (a split tile's pieces weigh a share each: they sort behind the unsplit tiles of their tile's full weight)
            std::stable_sort(run[x].begin(), run[x].end(), [&](int a, int b) { return work[a] / ksplit[a] > work[b] / ksplit[b]; });
          } else {
            // spatially coherent dispatch: tiles that share window fringes should stream the same slices at
            // the same time on this XCD, so that the fringe is fetched from HBM once and hit in L2 after
            // that.  Weights only in coarse classes (longest class first), inside a class the tiles of a
            // band column by column: 32 consecutive workgroups = a compact block of neighbours.
            int wmax = 1;
            for (int t : run[x]) wmax = std::max(wmax, work[t]);
            const int classes = opt_sched == 1 ? 6 : (opt_sched == 3 ? 3 : (opt_sched == 4 ? 12 : 1));
            const int hrows = (P.nty + 1) / 2;
            auto key = [&](int t) -> long long {
              const int ty = t / P.ntx, tx = t - ty * P.ntx;
              const int cls = classes > 1 ? std::min(classes - 1, (int)((long long)work[t] * classes / ((long long)wmax + 1))) : 0;
              const int band = ty >= hrows ? 1 : 0;
              return (((long long)(classes - 1 - cls) * 2 + band) * 4096 + tx) * 4096 + ty;
            };
            std::stable_sort(run[x].begin(), run[x].end(), [&](int a, int b) { return key(a) < key(b); });
          }
          longest = std::max(longest, run[x].size());
        }
        // a run's tiles become its workgroups {tile | piece << 20 | pieces << 26, cuts}, longest first by their own weights
        longest = 0;
        std::vector<std::vector<std::pair<int, int2>>> items(8);
        for (int x = 0; x < 8; ++x) {
          for (int t : run[x]) {
            const unsigned char *c = &aux->cuts[(size_t)t * 10];
            for (int k = 0; k < c[0]; ++k)
              items[x].push_back({piece_weight(t, k), make_int2(t | (k << 20) | ((int)c[0] << 26), (int)c[1 + k] | ((int)c[2 + k] << 8))});
          }
          if (opt_sched == 0 || opt_sched == 5) std::stable_sort(items[x].begin(), items[x].end(), [](const std::pair<int, int2> &a, const std::pair<int, int2> &b) { return a.first > b.first; });
          longest = std::max(longest, items[x].size());
        }
        order.assign(longest * 8, make_int2(-1, 0));
        for (int x = 0; x < 8; ++x)
          for (size_t k = 0; k < items[x].size(); ++k) order[k * 8 + x] = items[x][k].second;
        // ... followed by the list of split tiles for the merge pass (tile | pieces << 20), and their number last
        int nsplit = 0;
        for (int t = 0; t < nt; ++t)
          if (ksplit[t] > 1) { order.push_back(make_int2(t | ((int)ksplit[t] << 20), 0)); ++nsplit; }
        order.push_back(make_int2(nsplit, 0));
        aux->plan_work = work;
        aux->plan_cuts = aux->cuts;
        aux->plan_order = order;
        aux->plan_slots = slots;
      }
      const int nsplit = order.back().x;
      nblocks = (int)order.size() - 1 - nsplit;
      aux->ksplit_last.assign((size_t)nt, 1);
      for (int k = 0; k < nsplit; ++k) {
        const int code = order[order.size() - 1 - nsplit + k].x;
        aux->ksplit_last[code & 0xfffff] = (unsigned char)(code >> 20);
      }
      int maxseg = 1;
      for (int k = 0; k < nsplit; ++k) maxseg = std::max(maxseg, order[(size_t)nblocks + k].x >> 20);
      aux->nsplit_last = nsplit;
      aux->nblocks_last = nblocks;
      if (maxseg > 1) {
        // (room for the largest piece count at once: growing the buffer when a tile's count rises is a hipFree + hipMalloc,
        //  ~1 ms in the middle of a session -- 7 partial frames of a 1024^2 viewport are 112 MB of 288 GB)
        const size_t need = (size_t)(8 - 1) * P.W * P.H * 16;
        if (need > aux->seg_cap) {
          if (aux->d_seg) (void)hipFree(aux->d_seg);
          aux->d_seg = nullptr;
          aux->seg_cap = 0;
          hipError_t e = hipMalloc(&aux->d_seg, need);
          if (e != hipSuccess) return e;
          aux->seg_cap = need;
        }
      }
      Q.seg_out = (float4 *)aux->d_seg;
      if (dbg_time) {
        dbg_t[0] += dbg_scan;
        dbg_t[1] += dbg_now() - dbg_t0;
        if (++dbg_n % 60 == 0) {
          fprintf(stderr, "[smk] planning per frame: scans %.3f ms, all of it up to the launch %.3f ms, whole launcher (previous 60) %.3f ms\n", dbg_t[0] / 60, dbg_t[1] / 60, dbg_t[2] / 60);
          dbg_t[0] = dbg_t[1] = dbg_t[2] = 0;
        }
      }
      if (aux->frame_ev0) {  // the frame's kernel-time bracket opens here: planning is done
        hipError_t e = hipEventRecord(aux->frame_ev0, s);
        if (e != hipSuccess) return e;
      }
      auto same_order = [&]() { return aux->order_host.size() == order.size() && (order.empty() || !memcmp(aux->order_host.data(), order.data(), order.size() * sizeof(int2))); };
      if (!same_order()) {  // unchanged camera: the table on the device is still right
        if ((int)order.size() > aux->order_cap) {
          if (aux->d_order) (void)hipFree(aux->d_order);
          aux->d_order = nullptr;
          for (int k = 0; k < 4; ++k) {
            if (aux->order_ev[k]) (void)hipEventSynchronize(aux->order_ev[k]);
            if (aux->h_order[k]) (void)hipHostFree(aux->h_order[k]);
            aux->h_order[k] = nullptr;
          }
          aux->order_cap = 0;
          // (with headroom: the table grows by a few entries whenever a tile's piece count rises, and every regrowth is a
          //  device free + allocation and four pinned ones -- a 1 ms hiccup every few dozen frames on a shard)
          const size_t cap = order.size() * 2 + 1024;
          hipError_t e = hipMalloc((void **)&aux->d_order, cap * sizeof(int2));
          if (e != hipSuccess) return e;
          for (int k = 0; k < 4; ++k) {
            e = hipHostMalloc((void **)&aux->h_order[k], cap * sizeof(int2), hipHostMallocDefault);
            if (e != hipSuccess) return e;
          }
          aux->order_cap = (int)cap;
        }
        // pinned staging + a copy ON THE LAUNCH STREAM: a copy from pageable memory is not
        // stream-ordered against the kernel that follows (seen as wrong tiles when several
        // contexts render at once).  Four staging buffers in turn, each rewritten only after its
        // own last copy: the host does not wait for the stream unless it is four tables ahead.
        const int k = aux->order_next;
        aux->order_next = (k + 1) & 3;
        if (!aux->order_ev[k]) {
          hipError_t e = hipEventCreateWithFlags(&aux->order_ev[k], hipEventDisableTiming);
          if (e != hipSuccess) return e;
        } else {
          hipError_t e = hipEventSynchronize(aux->order_ev[k]);
          if (e != hipSuccess) return e;
        }
        memcpy(aux->h_order[k], order.data(), order.size() * sizeof(int2));
        hipError_t e = hipMemcpyAsync(aux->d_order, aux->h_order[k], order.size() * sizeof(int2), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
        e = hipEventRecord(aux->order_ev[k], s);
        if (e != hipSuccess) return e;
        aux->order_host.swap(order);
      }
      Q.order = aux->d_order;
      Q.trace = nullptr;
      if (P.lockstep & 32) {
        if (nblocks > aux->trace_cap) {
          if (aux->d_trace) (void)hipFree(aux->d_trace);
          aux->d_trace = nullptr;
          aux->trace_cap = 0;
          hipError_t e = hipMalloc((void **)&aux->d_trace, (size_t)nblocks * 32);
          if (e != hipSuccess) return e;
          aux->trace_cap = nblocks;
        }
        hipError_t e = hipMemsetAsync(aux->d_trace, 0, (size_t)nblocks * 32, s);
        if (e != hipSuccess) return e;
        aux->trace_n = nblocks;
        Q.trace = aux->d_trace;
      }
    }
    // developer diagnostics (option lockstep bits 2..64) live in separate instances of the f32 +
    // R8k kernels only: compiled into the product kernels they cost SGPRs (spills) in every frame
    const bool diag = (P.lockstep & ~1) != 0 && dtype == 1 && shade_kind == 1;
    const int nsplit_now = aux->nsplit_last, nblocks_now = aux->nblocks_last;
    if (nsplit_now > 0) {  // (their tick words are sums over the pieces)
      hipError_t e = hipMemsetAsync(aux->d_ticks, 0, (size_t)ticks_n_now * 20, s);
      if (e != hipSuccess) return e;
    }
    const int mtw = tw, mth = th;
    // (the merge pass's first launch costs the host a few milliseconds of code loading, which lands between the frame's
    //  events: paid here, in a context's first slice-ring frame -- not in the frame auto mode happens to be timing when
    //  the first tiles are cut.  One block whose entry names tile 0 with ONE piece: it rewrites that tile's pixels with
    //  themselves, before this frame's kernel writes them.)
    if (!aux->merge_warm && nblocks_now >= 1) {
      aux->merge_warm = true;
      static const int2 one = make_int2(0 | (1 << 20), 0);
      int2 *d_one = nullptr;
      if (hipMalloc((void **)&d_one, sizeof one) == hipSuccess) {
        if (hipMemcpyAsync(d_one, &one, sizeof one, hipMemcpyHostToDevice, s) == hipSuccess)
          hipLaunchKernelGGL(smk_k_slab_merge, dim3(1), dim3(256), 0, s, (const int2 *)d_one, mtw, mth, P.ntx, P.W, P.H, (const float4 *)P.out, P.out, 0);
        (void)hipStreamSynchronize(s);
        (void)hipFree(d_one);
      }
      (void)hipGetLastError();
    }
    auto after_launch = [&](hipError_t e) -> hipError_t {
      if (e == hipSuccess && nsplit_now > 0) {
        hipLaunchKernelGGL(smk_k_slab_merge, dim3(nsplit_now), dim3(256), 0, s, (const int2 *)aux->d_order + nblocks_now, mtw, mth, P.ntx, P.W, P.H,
                           (const float4 *)aux->d_seg, P.out, P.blend == SMK_BLEND_MAX ? 1 : 0);
        e = hipGetLastError();
      }
      if (e != hipSuccess || aux->ticks_pending) return e;
      // fetch this frame's per-tile durations (one copy in flight at a time)
      hipError_t e2 = hipMemcpyAsync(aux->h_ticks, aux->d_ticks, (size_t)ticks_n_now * 4, hipMemcpyDeviceToHost, s);
      if (e2 == hipSuccess && nsplit_now > 0) e2 = hipMemcpyAsync(aux->h_pticks, aux->d_pticks, (size_t)ticks_n_now * 32, hipMemcpyDeviceToHost, s);
      if (nsplit_now > 0) aux->cuts_pending = aux->cuts; else aux->cuts_pending.clear();
      if (e2 == hipSuccess) e2 = hipEventRecord(aux->ticks_ev, s);
      if (e2 != hipSuccess) return e2;
      aux->ticks_pending = true;
      aux->ticks_pending_sig = ticks_sig_now;
      aux->ticks_pending_n = ticks_n_now;
      return hipSuccess;
    };
    return after_launch(P.sh.on     ? smk_slab_dispatch_shadow(P, Q, dtype, tf_mode, shade_kind, nw, nl, lds, nblocks, why, s)
                        : dtype == 0 ? smk_slab_dispatch_u8(P, Q, tf_mode, shade_kind, nw, nl, diag, lds, nblocks, why, s)
                                     : smk_slab_dispatch_f32(P, Q, tf_mode, shade_kind, nw, nl, diag, lds, nblocks, why, s));
  }
  }  // pass
  (void)forced;
  *why = "no configuration fits";
  return hipErrorNotSupported;
}
#endif  // SLAB_PART == 0
