// smk_cols.hip -- kernel C: the column-stream ray-marcher (object order).
//
// What it replaces in the reference is what the other two ray-marchers replace: the per-slice polygon loop and the
// fragment pipeline of VolumeRenderer::render3DVA (VolumeRenderer.cpp:507-741) / NV20VolRen3D::render3DVA
// (NV20VolRen3D.cpp:852-1083) / R8kVolRen3D::render3DVA (R8kVolRen3D.cpp:1257-1604), with the brick-wise draw of
// renderBricks (NV20VolRen3D.cpp:190-231) turned into what it is on a 288 GB part: a memory layout.
//
// The slice-ring kernel (smk_slab.hip) gives a workgroup a PIXEL tile and streams the window of every slice the tile's
// ray bundle crosses: windows of neighbouring tiles overlap (1.4 x the algorithmic bytes on the 1024^3 frame), their
// row pieces are scattered, and one workgroup owns a tile for all planes.  Here the decomposition is by VOLUME:
//
//   * layout: for the view's principal axis S the stored box is cut into columns of CW x CH cells along (U, V); a
//     column's slice image -- (CW+1) x (CH+1) voxels, the bilinear halo included -- is contiguous in memory and the
//     images of consecutive slices follow one another: a column is ONE sequential stream, read exactly once, by
//     whole-KiB LDS-DMA instructions with no masks, no per-slice addresses, no fringe that depends on the view
//     (the stream is (CW+1)(CH+1)/(CW CH) ~ 1.05-1.08 x the stored bytes; built lazily per principal axis, 288 GB);
//   * job = one column x one chunk of CL slice positions; a workgroup per job: NL loader waves stream the chunk into an
//     LDS ring, NW consumer waves sample it.  All jobs are the same length, there are thousands of them, none depends
//     on another: no tile schedule, no longest workgroup;
//   * rays pass THROUGH columns.  In its set-up a job lists the rays that cross its cell box, each with the exact
//     interval of planes whose samples belong to the box, sorted by the position they enter at; a consumer lane
//     without a ray takes the next one of the list, marches it (no membership test in the loop: every sample it takes
//     is the ray's), and writes the ray's partial composite -- the "segment" of (ray, job) -- when the ray leaves;
//   * segments of a ray are keyed by the job's place along the ray (monotone in column indices and chunk), stored in
//     layers[key][pixel] with a bit per key in a per-pixel mask; the resolve pass blends a pixel's segments in key
//     order (front to back "over", or max) and clears the mask.
//
// Sample placement, membership, interpolation order, classification and shading are EXACTLY those of the gather
// kernel (same fma chains, smk_device.h), so every sample's source colour is bit-identical; what differs is the
// association of the blend: ((s1 over s2) over (s3 over s4)) instead of (((s1 over s2) over s3) over s4).  "Over" is
// associative in exact arithmetic; in fp32 the frames agree to a few ulp per segment (tests: <= 2e-5 against the
// gather kernel, <= 1e-4 against the CPU checker, the project's stated tolerance).  Exact early termination holds
// inside a segment only.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "smk_device.h"

#define COL_DONE 0x3fffffff
#define COL_MAX_CL 256   // positions per job at most (the per-slice table lives in LDS)
#define COL_BOX_MARGIN 0.05f

// wave-uniform description of one launch
struct ColParams {
  const char *lay;            // layout base: [cv][cu][s][(CH+1)][(CW+1)] voxels, slice images of slice_bytes
  int CW, CH, ncu, ncv;       // cells per column along U, V; columns
  int Ou, Ov, Os;             // stored-box origin (global voxel index) along U, V, S
  int Du, Dv, Ds;             // stored-box dims
  int slice_bytes;            // (CW+1)(CH+1) voxels, rounded up to 16 bytes
  int n_ch;                   // DMA wave-instructions per slice = ceil(slice_bytes / 1024)
  unsigned long long last_mask;  // lanes of the last one
  int nslots, maxfly, wstep;
  int take_min, take_wait;    // a wave takes new rays when this many lanes are free, or after this many turns
  int ring_bytes;             // LDS bytes in front of the tables: the ring, at least the set-up's scratch (the unsorted rays)
  int CL, nck;                // positions per chunk, chunks
  int dir;                    // +1: rays advance towards +S
  float Mx[4], My[4], Mw[4];  // voxel (global coordinates) -> continuous pixel: x = Mx.(X,1) / Mw.(X,1)
  float4 *layers;             // [nkeys][npix]
  unsigned long long *masks;  // [npix][mask_words]
  int nkeys, mask_words;
  int use_ah, use_occ, fast_tf;
  int *status;                // host-visible: status_tag | (1 protocol time-out, 3 a job's rays do not fit lanes or list, 5 a ray's plane count
                              // does not fit its list entry)
  int status_tag;             // the frame's id << 8
  unsigned *job_ticks;        // [njobs] duration of each job's workgroup in 100 MHz ticks, or null
  unsigned long long *counts; // [8] samples taken | visible | slices streamed | segments written | consumer wave-iterations | lanes with a sample to
                              // take in them | iterations in which some lane changes rays | lanes changing rays (developer statistics)
};

typedef float c_v4f __attribute__((ext_vector_type(4)));
typedef unsigned c_v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const void *c_lds_cptr_t;

#define COL_READ8(INS, OFF)                                                                                              \
  asm volatile(INS " %0, %8\n\t" INS " %1, %8 offset:" OFF "\n\t" INS " %2, %9\n\t" INS " %3, %9 offset:" OFF "\n\t"      \
               INS " %4, %10\n\t" INS " %5, %10 offset:" OFF "\n\t" INS " %6, %11\n\t" INS " %7, %11 offset:" OFF "\n\t" \
               "s_waitcnt lgkmcnt(0)"                                                                                    \
               : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])  \
               : "v"(a), "v"(ap), "v"(b), "v"(bp)                                                                        \
               : "memory")
// the 8 corners of one sample, whole voxels (normals included), ONE batch of LDS reads behind ONE wait (left to hipcc
// the reads are waited for one by one: smk_slab.hip)
__device__ __forceinline__ void col_read8(unsigned a, unsigned ap, unsigned b, unsigned bp, c_v4f (&q)[8]) { COL_READ8("ds_read_b128", "16"); }
__device__ __forceinline__ void col_read8(unsigned a, unsigned ap, unsigned b, unsigned bp, c_v2u (&q)[8]) { COL_READ8("ds_read_b64", "8"); }
#undef COL_READ8

__device__ __forceinline__ void col_wait_vmcnt(int n) {
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

__device__ __forceinline__ int col_lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void col_lds_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// LDS accesses of a LOADER wave go through asm: with an LDS-DMA in flight hipcc puts s_waitcnt vmcnt(0) in front of
// every LDS read it can see (smk_slab.hip, cdna_hip_programming.md 5.7)
__device__ __forceinline__ int col_raw_lds_b32(const void *p) {
  int v;
  unsigned a = (unsigned)(size_t)(c_lds_cptr_t)p;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ void col_raw_lds_st_b32(void *p, int v) {
  unsigned a = (unsigned)(size_t)(c_lds_cptr_t)p;
  asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
}
// minimum over the 64 lanes of a fully active wave on the DPP network: v_min with a DPP operand (left to hipcc each step is
// v_mov + s_nop + v_mov_dpp + v_min)
__device__ __forceinline__ int col_wave_min(int v) {
  asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
               : "+v"(v));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

struct ColTexel4 {
  uint32_t a, b, c, d;
  float fs, ft;
};
__device__ __forceinline__ ColTexel4 col_tex2d_fetch(const uint32_t *tex, int ss, int s0, int t0, float fs, float ft) {
  ColTexel4 o;
  o.fs = fs;
  o.ft = ft;
  const unsigned off = (unsigned)(t0 * ss + s0) * 4u;
  const char *tb = reinterpret_cast<const char *>(tex);
  uint2 lo, hi;
  __builtin_memcpy(&lo, tb + off, 8);
  __builtin_memcpy(&hi, tb + (off + (unsigned)ss * 4u), 8);
  o.a = lo.x; o.b = lo.y; o.c = hi.x; o.d = hi.y;
  return o;
}
__device__ __forceinline__ float col_tex_chan(const ColTexel4 &x, int k) {
  return smk_lerp(smk_lerp(smk_ub(x.a, k), smk_ub(x.b, k), x.fs), smk_lerp(smk_ub(x.c, k), smk_ub(x.d, k), x.fs), x.ft) * SMK_INV255;
}

#define COL_MAX_RAYS 4096   // rays one job can list (8 bytes each in LDS)

// one ray's marching state (registers)
struct ColRay {
  int i, j;
  float A[3], B[3];
  int m, m_out;  // plane of the next sample, last plane of the ray inside this job
  int pb;        // position of the next sample relative to the job's first (marching order); COL_DONE = the lane has no ray
  float sc;      // clamped principal-axis coordinate of the next sample
  int bi;        // and its base slice (global index)
  float C0, C1, C2, C3;
};

// NW consumer waves, NL loader waves.
//
// Set-up, per job: every pixel whose ray can cross the job's cell box (the bounding box of its eight projected corners)
// gets the exact interval [m_in, m_out] of planes whose sample has its base cell in the box and passes the gather
// kernel's membership test -- coordinates are monotone in the plane index, so it IS an interval, found by bracketing
// and testing the ends with the very fma chains the samples use -- and the rays that have one are listed in LDS in
// the order of their first position.  Marching: a lane without a ray takes the next one of the list (the list is
// sorted by entry, so everything taken later enters later: the head's entry bounds the ring's tail); every sample a
// lane takes is one of its ray's, no test in the loop; a ray that ends writes its segment.
template <int DT, int SH, int PERM, int TF, int NW, int NL>
__global__ __launch_bounds__((NW + NL) * 64) void smk_k_cols(const RenderParams P, const ColParams Q) {
  constexpr int VB = DT == 0 ? 8 : 16;
  constexpr int NTH = (NW + NL) * 64;
  constexpr int AS = PERM == 0 ? 2 : (PERM == 1 ? 1 : 0);
  constexpr int AU = PERM == 2 ? 1 : 0;
  constexpr int AV = PERM == 0 ? 1 : 2;
  extern __shared__ __align__(16) unsigned char smem[];
  const int slot_stride = Q.slice_bytes;
  // LDS: ring | ray list [COL_MAX_RAYS] uint2 | entry position of each listed ray [COL_MAX_RAYS] u8 | slot address of each of
  // the job's slices [COL_MAX_CL + 1] | histogram / cursors [COL_MAX_CL + 1] | control words [32] | alpha_H | occupancy bitmap
  uint2 *list = reinterpret_cast<uint2 *>(smem + (size_t)Q.ring_bytes);
  unsigned char *epos = reinterpret_cast<unsigned char *>(list + COL_MAX_RAYS);
  int *slot_addr = reinterpret_cast<int *>(epos + COL_MAX_RAYS);
  int *hist = slot_addr + (COL_MAX_CL + 4);
  // control words: [0] error flag  [1] list head  [2] rays listed  [3] candidates' scratch count  [4..8) landed per loader
  // [8..24) progress per consumer wave
  int *hist2 = hist + (COL_MAX_CL + 4);  // exits per position (set-up: the lanes must hold every ray that is inside at one position)
  int *ctl = hist2 + (COL_MAX_CL + 4);
  float *ah = reinterpret_cast<float *>(ctl + 32);
  const uint32_t *occ = reinterpret_cast<const uint32_t *>(ah + (Q.use_ah ? P.sv : 0));
  uint4 *tmp = reinterpret_cast<uint4 *>(smem);  // set-up only (the ring is not in use yet): unsorted rays {ij, m_in, m_out, entry}

  const unsigned t_begin = (unsigned)__builtin_amdgcn_s_memrealtime();
  const int job = blockIdx.x;
  const int cu = job % Q.ncu, cv = (job / Q.ncu) % Q.ncv, ck = job / (Q.ncu * Q.ncv);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave >= NW;
  const int NS = P.N[AS], NU = P.N[AU], NV = P.N[AV];
  const int dir = Q.dir;
  const smk_raycoef &rc = P.rc;
  // positions (marching order) of the whole stored box: 0 .. Ds-2; base slice (stored) of position g = dir > 0 ? g : Ds-2-g
  const int pa = ck * Q.CL;                                  // first global position of the job
  const int npos = min(Q.CL, Q.Ds - 1 - pa);                 // positions of the job; loads 0 .. npos
  const int slo = dir > 0 ? pa : Q.Ds - 1 - (pa + npos);     // lowest stored slice the job touches
  const int u0 = cu * Q.CW, v0 = cv * Q.CH;                  // first cell of the column (stored coordinates)
  const unsigned ring_addr = (unsigned)(size_t)(c_lds_cptr_t)smem;
  const unsigned pitch_b = (unsigned)(Q.CW + 1) * VB;
  const int psgn = dir > 0 ? 1 : -1;
  const int poff = dir > 0 ? -(Q.Os + pa) : (Q.Os + Q.Ds - 2 - pa);  // relative position of global base slice b: psgn * b + poff
  const int eoff = -Q.Os - slo;                                      // slot-table entry of global slice b: b + eoff

  // ---- the job's cell box as coordinate bounds (inclusive), per model axis.  Base cell i0 = min((int)clamp(p, 0, N-1), N-2)
  // is monotone in p: i0 >= g  <=>  p >= g (g >= 1), i0 < h  <=>  p < h (h <= N-2); at a face of the volume the cell range
  // is open-ended and the membership bound [lo, hin] of the region takes over.
  float Lo[3], Hi[3];
  {
    const int gu0 = u0 + Q.Ou, gv0 = v0 + Q.Ov, gs0 = slo + Q.Os;  // global index of the first cell / lowest base slice
    auto below = [](float x) -> float { return __uint_as_float(__float_as_uint(x) - 1u); };  // (x >= 1)
    Lo[AU] = gu0 > 0 ? fmaxf(P.lo[AU], (float)gu0) : P.lo[AU];
    Hi[AU] = gu0 + Q.CW <= NU - 2 ? fminf(P.hin[AU], below((float)(gu0 + Q.CW))) : P.hin[AU];
    Lo[AV] = gv0 > 0 ? fmaxf(P.lo[AV], (float)gv0) : P.lo[AV];
    Hi[AV] = gv0 + Q.CH <= NV - 2 ? fminf(P.hin[AV], below((float)(gv0 + Q.CH))) : P.hin[AV];
    Lo[AS] = gs0 > 0 ? fmaxf(P.lo[AS], (float)gs0) : P.lo[AS];
    Hi[AS] = gs0 + npos <= NS - 2 ? fminf(P.hin[AS], below((float)(gs0 + npos))) : P.hin[AS];
  }
  const bool empty_job = !(Lo[0] <= Hi[0] && Lo[1] <= Hi[1] && Lo[2] <= Hi[2]) || npos <= 0;

  // ---- set-up (1): control words, slot table, tables of the classification fast path, histogram
  if (tid < 32) ctl[tid] = (tid >= 4 && tid < 8) ? (tid - 4 < NL ? tid - 4 : COL_DONE) : (tid >= 8 && tid < 8 + NW ? 0 : (tid >= 8 ? COL_DONE : 0));
  for (int e = tid; e <= npos; e += NTH) {
    const int r = dir > 0 ? e : npos - e;  // load index of this slice
    slot_addr[e] = (int)ring_addr + (r % Q.nslots) * slot_stride;
  }
  for (int e = tid; e < 2 * (COL_MAX_CL + 4); e += NTH) hist[e] = 0;  // (hist and hist2 are adjacent)
  if (Q.use_ah)
    for (int e = tid; e < P.sv; e += NTH) ah[e] = smk_ub(P.tf_h[e], 3);
  if (Q.use_occ) {
    uint32_t *occ_w = const_cast<uint32_t *>(occ);
    for (int e = tid; e < P.occ_roww * (TF == 2 ? P.s3g : P.sg); e += NTH) occ_w[e] = P.tf_occ[e];
  }
  // ---- set-up (2): candidate pixels = the bounding box of the job box's eight projected corners (a sample in the box
  // projects inside their convex hull).  Every thread computes it (wave-uniform arithmetic).
  int ci0 = 0, ci1 = -1, cj0 = 0, cj1 = -1;
  if (!empty_job) {
    float xl = 1e30f, xh = -1e30f, yl = 1e30f, yh = -1e30f;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float X[3];
      const float eps = 0.01f;
      X[0] = (c & 1) ? Hi[0] + eps : Lo[0] - eps;
      X[1] = (c & 2) ? Hi[1] + eps : Lo[1] - eps;
      X[2] = (c & 4) ? Hi[2] + eps : Lo[2] - eps;
      const float wq = Q.Mw[0] * X[0] + Q.Mw[1] * X[1] + Q.Mw[2] * X[2] + Q.Mw[3];
      const float xn = Q.Mx[0] * X[0] + Q.Mx[1] * X[1] + Q.Mx[2] * X[2] + Q.Mx[3];
      const float yn = Q.My[0] * X[0] + Q.My[1] * X[1] + Q.My[2] * X[2] + Q.My[3];
      bad = bad || !(wq > 1e-12f);
      const float iw = 1.0f / wq;
      xl = fminf(xl, xn * iw); xh = fmaxf(xh, xn * iw);
      yl = fminf(yl, yn * iw); yh = fmaxf(yh, yn * iw);
    }
    if (bad) {
      if (tid == 0) ctl[0] = 3;
    } else {
      // pixels whose centre i + .5 lies in [min, max]
      const float lim = 30000.f;
      ci0 = max((int)ceilf(fminf(fmaxf(xl - 0.5f - COL_BOX_MARGIN, -lim), lim)), 0);
      ci1 = min((int)floorf(fminf(fmaxf(xh - 0.5f + COL_BOX_MARGIN, -lim), lim)), P.W - 1);
      cj0 = max((int)ceilf(fminf(fmaxf(yl - 0.5f - COL_BOX_MARGIN, -lim), lim)), 0);
      cj1 = min((int)floorf(fminf(fmaxf(yh - 0.5f + COL_BOX_MARGIN, -lim), lim)), P.H - 1);
    }
  }
  __syncthreads();
  // base slice (global) and clamped coordinate of plane q on a ray: exactly smk_lin_clamp's (xc, i0) of the principal axis
  auto base_slice = [&](const float (&A)[3], const float (&B)[3], int q, float &sc_out) -> int {
    const float s = __fmaf_rn((float)q, B[AS], A[AS]);
    sc_out = smk_clampf(s, 0.0f, (float)(NS - 1));
    return min((int)sc_out, NS - 2);
  };
  // ---- set-up (3): the rays.  A thread per candidate pixel (rows of 32), the plane interval of its ray in the job box.
  {
    const int cw = ci1 - ci0 + 1, chh = cj1 - cj0 + 1;
    for (int jj = tid >> 5; jj < chh; jj += NTH / 32) {
      for (int i0 = 0; i0 < cw; i0 += 32) {
        const int ii = i0 + (tid & 31);
        bool have = false;
        int m_in = 0, m_out = -1;
        const int i = ci0 + ii, j = cj0 + jj;
        float A[3], B[3];
        if (ii < cw) {
          const float px = __fmaf_rn((float)i + 0.5f, rc.pxs, rc.pxl);
          const float py = __fmaf_rn((float)j + 0.5f, rc.pys, rc.pyl);
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            A[a] = __fmaf_rn(px, rc.Ax[a], __fmaf_rn(py, rc.Ay[a], rc.Ac[a]));
            B[a] = __fmaf_rn(px, rc.Bx[a], __fmaf_rn(py, rc.By[a], rc.Bc[a]));
          }
          // bracket: real-valued plane range per axis, a plane of slack (the quotient is off by far less); the exact test at the ends decides
          float tin = 0.0f, tout = (float)(rc.nplanes - 1);
          bool none = false;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            if (fabsf(B[a]) > 1e-20f) {
              const float inv = 1.0f / B[a];
              const float t1 = (Lo[a] - A[a]) * inv, t2 = (Hi[a] - A[a]) * inv;
              tin = fmaxf(tin, fminf(t1, t2) - 1.0f);
              tout = fminf(tout, fmaxf(t1, t2) + 1.0f);
            } else if (!(A[a] >= Lo[a] && A[a] <= Hi[a])) {
              none = true;
            }
          }
          if (!none && tin <= tout) {
            int qa = (int)floorf(fmaxf(tin, 0.0f)), qb = (int)ceilf(fminf(tout, (float)(rc.nplanes - 1)));
            auto in = [&](int q) -> bool {
              const float qf = (float)q;
              const float p0 = __fmaf_rn(qf, B[0], A[0]), p1 = __fmaf_rn(qf, B[1], A[1]), p2 = __fmaf_rn(qf, B[2], A[2]);
              return ((int)(smk_clampf(p0, Lo[0], Hi[0]) == p0) & (int)(smk_clampf(p1, Lo[1], Hi[1]) == p1) & (int)(smk_clampf(p2, Lo[2], Hi[2]) == p2)) != 0;
            };
            // the ends move inwards until they are inside (a few steps: the slack is two planes; a ray that only grazes
            // the box ends empty).  Coordinates are monotone in q, so everything between two inside planes is inside.
#pragma unroll 1
            for (int k = 0; k < 8 && qa <= qb && !in(qa); ++k) ++qa;
#pragma unroll 1
            for (int k = 0; k < 8 && qa <= qb && !in(qb); ++k) --qb;
            if (qa <= qb && in(qa) && in(qb)) {
              have = true;
              m_in = qa;
              m_out = qb;
            }
          }
        }
        // append to the unsorted list, count its entry position
        const unsigned long long bal = __ballot(have);
        if (bal) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&ctl[3], (int)__popcll(bal));
          base = __builtin_amdgcn_readfirstlane(base);
          if (have) {
            const int idx = base + (int)__popcll(bal & ((1ull << lane) - 1ull));
            float sc;
            const int ep = psgn * base_slice(A, B, m_in, sc) + poff;  // in [0, npos)
            if (idx < COL_MAX_RAYS) {
              tmp[idx] = make_uint4((unsigned)i | ((unsigned)j << 16), (unsigned)m_in, (unsigned)m_out, (unsigned)ep);
              atomicAdd(&hist[min(max(ep, 0), COL_MAX_CL)], 1);
              const int xp = psgn * base_slice(A, B, m_out, sc) + poff;
              atomicAdd(&hist2[min(max(xp, 0), COL_MAX_CL)], 1);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const int nrays = min(ctl[3], COL_MAX_RAYS);
  if (tid == 0 && ctl[3] > COL_MAX_RAYS) ctl[0] = 3;  // more rays than the list holds: reported, the frame is rendered another way
  // ---- set-up (4): order by entry position (counting sort: exclusive prefix of the histogram, then scatter)
  if (wave == 0) {
    int carry = 0;
    for (int b0 = 0; b0 <= COL_MAX_CL; b0 += 64) {
      const int e = b0 + lane;
      const int v = e <= COL_MAX_CL ? hist[e] : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      if (e <= COL_MAX_CL) hist[e] = carry + incl - v;
      carry += __builtin_amdgcn_readlane(incl, 63);
    }
    // rays inside at position T = entered at <= T minus left before T: all of them want a lane at once
    int carry_in = 0, carry_out = 0, worst = 0;
    for (int b0 = 0; b0 <= COL_MAX_CL; b0 += 64) {
      const int e = b0 + lane;
      const int nin = e < COL_MAX_CL ? hist[e + 1] : nrays;           // entries at positions <= e (exclusive prefix of e + 1)
      const int vo = e <= COL_MAX_CL ? hist2[e] : 0;
      int incl = vo;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const int nout = carry_out + incl - vo;                         // exits at positions < e
      int alive = e <= COL_MAX_CL ? nin - nout : 0;
      for (int o = 32; o > 0; o >>= 1) alive = max(alive, __shfl_xor(alive, o));
      worst = max(worst, alive);
      carry_out += __builtin_amdgcn_readlane(incl, 63);
      (void)carry_in;
    }
    if (lane == 0 && worst > NW * 64) ctl[0] = 3;  // more rays inside at once than lanes: reported before anything is streamed
  }
  __syncthreads();
  for (int n = tid; n < nrays; n += NTH) {
    const uint4 t = tmp[n];
    const int at = atomicAdd(&hist[min(max((int)t.w, 0), COL_MAX_CL)], 1);
    const unsigned cnt = t.z - t.y + 1u;
    if (cnt > 4095u || t.y >= (1u << 20)) ctl[0] = 5;
    list[at] = make_uint2(t.x, t.y | (cnt << 20));
    epos[at] = (unsigned char)t.w;
  }
  if (tid == 0) ctl[2] = nrays;
  __syncthreads();  // list visible; the ring's memory is free from here on.  LAST workgroup barrier before the end
  const bool failed_setup = ctl[0] != 0;
  const unsigned t_setup = (unsigned)__builtin_amdgcn_s_memrealtime();

  // developer statistics, wave-scalar
  unsigned n_samples = 0, n_visible = 0, n_segments = 0, n_iters = 0, n_act = 0, n_swit = 0, n_swl = 0;

  if (!failed_setup && nrays > 0) {
    if (is_loader) {
      // ================================ loader wave ============================================
      // Loads r = 0 .. npos in marching order, mine are r = lid, lid + NL, ...; the slot of r is r % nslots and may be
      // rewritten once every consumer is past position r - nslots AND no ray still to be taken enters before it.  A slice
      // is n_ch whole-KiB LDS-DMA instructions from ONE contiguous run of memory (the last one with fewer lanes), the same
      // count for every slice, so the in-order vmcnt tells which slices have landed.
      __builtin_amdgcn_s_setprio(3);
      const int lid = wave - NW;
      const char *col_base = Q.lay + ((size_t)(cv * Q.ncu + cu) * Q.Ds) * (size_t)Q.slice_bytes;
      const unsigned voff = (unsigned)lane * 16u, voff1 = voff + 1024u, voff2 = voff + 2048u, voff3 = voff + 3072u;
      int r = lid, inflight = 0, landed = lid, idle = 0, minp = 0;
      auto poll_progress = [&]() -> int {
        int v = COL_DONE;
        if (lane < 16) v = col_raw_lds_b32(&ctl[8 + lane]);
        if (lane == 16) {  // the first position of the next ray nobody has taken yet
          const int head = col_raw_lds_b32(&ctl[1]);
          if (head < nrays) {
            const unsigned a = (unsigned)(size_t)(c_lds_cptr_t)(epos + head);
            unsigned b;
            asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(a) : "memory");
            v = (int)b;
          }
        }
        v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
        v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
        v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
        v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
        return min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16));
      };
      unsigned keep_m0;
      unsigned n_loaded = 0;
      while (landed <= npos) {
        bool stop = false;
        while (r <= npos && inflight < Q.maxfly) {
          if (r - Q.nslots >= minp) {
            minp = poll_progress();
            if (minp >= COL_DONE) { stop = true; break; }  // every consumer finished: the rest is not needed
            if (r - Q.nslots >= minp) break;
          }
          const int sl = dir > 0 ? slo + r : slo + npos - r;  // stored slice
          const char *src = col_base + (size_t)sl * (size_t)Q.slice_bytes;
          unsigned dst = ring_addr + (unsigned)((r % Q.nslots) * slot_stride);
          int k = 0;
          // four whole instructions per statement: M0 steps through their LDS images
          for (; k + 4 < Q.n_ch; k += 4) {
            asm volatile("s_mov_b32 %[km], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %[v0], %[src]\n\ts_add_u32 m0, m0, 0x400\n\t"
                         "global_load_lds_dwordx4 %[v1], %[src]\n\ts_add_u32 m0, m0, 0x400\n\t"
                         "global_load_lds_dwordx4 %[v2], %[src]\n\ts_add_u32 m0, m0, 0x400\n\t"
                         "global_load_lds_dwordx4 %[v3], %[src]\n\t"
                         "s_mov_b32 m0, %[km]"
                         : [km] "=&s"(keep_m0)
                         : [dst] "s"(dst), [src] "s"(src), [v0] "v"(voff), [v1] "v"(voff1), [v2] "v"(voff2), [v3] "v"(voff3)
                         : "memory", "scc");
            src += 4096;
            dst += 4096u;
          }
          for (; k + 1 < Q.n_ch; ++k) {
            asm volatile("s_mov_b32 %[km], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[src]\n\ts_mov_b32 m0, %[km]"
                         : [km] "=&s"(keep_m0)
                         : [dst] "s"(dst), [src] "s"(src), [v] "v"(voff)
                         : "memory");
            src += 1024;
            dst += 1024u;
          }
          {
            unsigned long long keep_exec;
            asm volatile("s_mov_b32 %[km], m0\n\ts_mov_b64 %[ke], exec\n\ts_mov_b32 m0, %[dst]\n\ts_mov_b64 exec, %[em]\n\t"
                         "global_load_lds_dwordx4 %[v], %[src]\n\ts_mov_b64 exec, %[ke]\n\ts_mov_b32 m0, %[km]"
                         : [km] "=&s"(keep_m0), [ke] "=&s"(keep_exec)
                         : [dst] "s"(dst), [src] "s"(src), [v] "v"(voff), [em] "s"(Q.last_mask)
                         : "memory");
          }
          ++n_loaded;
          r += NL;
          ++inflight;
        }
        if (stop) break;
        if (inflight > 0) {
          col_wait_vmcnt(Q.n_ch * (inflight - 1));  // retire the oldest slice in flight
          --inflight;
          landed += NL;
          col_raw_lds_st_b32(&ctl[4 + lid], landed);
          idle = 0;
        } else {
          const int flagged = col_raw_lds_b32(&ctl[0]);
          if (++idle > (1 << 22) || flagged) {  // bounded spin: never hang the GPU on a protocol bug
            if (!flagged) col_raw_lds_st_b32(&ctl[0], 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      col_wait_vmcnt(0);
      if (Q.counts && lane == 0 && n_loaded) atomicAdd(&Q.counts[2], (unsigned long long)n_loaded);
    } else {
      // ================================ consumer waves ==========================================
      // (the 18 ray coefficients are wanted every time a lane takes a ray; as scalars they do not fit beside the loop's
      //  own -- the compiler reloaded them from the kernel arguments each time: kept in vector registers instead)
      float kAc[3], kAx[3], kAy[3], kBc[3], kBx[3], kBy[3], kps[4];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        asm volatile("v_mov_b32 %0, %1" : "=v"(kAc[a]) : "s"(rc.Ac[a]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(kAx[a]) : "s"(rc.Ax[a]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(kAy[a]) : "s"(rc.Ay[a]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(kBc[a]) : "s"(rc.Bc[a]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(kBx[a]) : "s"(rc.Bx[a]));
        asm volatile("v_mov_b32 %0, %1" : "=v"(kBy[a]) : "s"(rc.By[a]));
      }
      asm volatile("v_mov_b32 %0, %1" : "=v"(kps[0]) : "s"(rc.pxs));
      asm volatile("v_mov_b32 %0, %1" : "=v"(kps[1]) : "s"(rc.pxl));
      asm volatile("v_mov_b32 %0, %1" : "=v"(kps[2]) : "s"(rc.pys));
      asm volatile("v_mov_b32 %0, %1" : "=v"(kps[3]) : "s"(rc.pyl));
      ColRay y;
      y.C0 = y.C1 = y.C2 = y.C3 = 0.f;
      y.i = y.j = 0;
      y.m = 0;
      y.m_out = -1;
      y.sc = 0.f;
      y.bi = 0;
      y.pb = COL_DONE;
#pragma unroll
      for (int a = 0; a < 3; ++a) y.A[a] = y.B[a] = 0.f;
      bool exhausted = false;  // the list has no ray left for this lane
      auto landed_all = [&]() -> int {
        int4 v;
        const unsigned a = (unsigned)(size_t)(c_lds_cptr_t)(ctl + 4);
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        return min(min(v.x, v.y), min(v.z, v.w));
      };
      // a ray's partial composite for this job -> layers[key][pixel], key = the job's place along the ray
      auto flush = [&]() {
        if (y.C0 != 0.f || y.C1 != 0.f || y.C2 != 0.f || y.C3 != 0.f) {
          const int key = (y.B[AU] >= 0.f ? cu : Q.ncu - 1 - cu) + (y.B[AV] >= 0.f ? cv : Q.ncv - 1 - cv) + ck;
          const size_t pix = (size_t)y.j * P.W + y.i;
          Q.layers[(size_t)key * ((size_t)P.W * P.H) + pix] = make_float4(y.C0, y.C1, y.C2, y.C3);
          atomicOr(&Q.masks[pix * Q.mask_words + (key >> 6)], 1ull << (key & 63));
          n_segments += 1;
        }
        y.C0 = y.C1 = y.C2 = y.C3 = 0.f;
      };
      int pos = 0, have = 0, since_take = 0;
      bool first_turn = true;
      for (;;) {
        bool took = false;
        // ---- lanes without a ray take the next ones of the list (one counter bump per wave)
        {
          const bool want = y.pb >= COL_DONE && !exhausted;
          unsigned long long bal = __ballot(want);
          // (a take costs the wave ~60 vector instructions whatever the number of lanes: free lanes wait for company
          //  unless nobody works, or they have waited take_wait turns)
          const int nfree = (int)__popcll(bal);
          // (never while the wave is about to wait for data: the ray at the list's head may be what holds the ring's tail)
          if (bal && nfree < Q.take_min && since_take < Q.take_wait && __any(y.pb < COL_DONE) && have >= min(pos + 2 + Q.wstep, npos + 1)) {
            bal = 0;
            ++since_take;
          }
          if (bal) {
            since_take = 0;
            took = true;
            // the word this wave has published must cover whatever it takes: every ray from the head on enters at or
            // behind the head's entry (sorted list), so that entry goes into the word BEFORE the head moves -- a loader
            // that sees the new head also sees the lowered word
            {
              const int h0 = col_lds_ld(&ctl[1]);
              if (h0 < nrays) {
                const int e0 = (int)epos[h0];
                if (e0 < pos) {
                  pos = e0;
                  if (lane == 0) col_lds_st(&ctl[8 + wave], pos);
                }
              }
            }
            int base = 0;
            if (lane == 0) base = atomicAdd(&ctl[1], (int)__popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);
            if (want) {
              const int n = base + (int)__popcll(bal & ((1ull << lane) - 1ull));
              if (n < nrays) {
                const uint2 e = list[n];
                y.i = (int)(e.x & 0xffffu);
                y.j = (int)(e.x >> 16);
                y.m = (int)(e.y & 0xfffffu);
                y.m_out = y.m + (int)(e.y >> 20) - 1;
                const float px = __fmaf_rn((float)y.i + 0.5f, kps[0], kps[1]);
                const float py = __fmaf_rn((float)y.j + 0.5f, kps[2], kps[3]);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                  y.A[a] = __fmaf_rn(px, kAx[a], __fmaf_rn(py, kAy[a], kAc[a]));
                  y.B[a] = __fmaf_rn(px, kBx[a], __fmaf_rn(py, kBy[a], kBc[a]));
                }
                y.bi = base_slice(y.A, y.B, y.m, y.sc);
                y.pb = psgn * y.bi + poff;
              } else {
                exhausted = true;
              }
            }
            if (Q.counts) {
              n_swit += 1;
              n_swl += (unsigned)__popcll(bal);
            }
          }
        }
        if (!__any(y.pb < COL_DONE)) break;  // (a lane without a ray here is exhausted: the list is sorted, nothing is left)
        // progress = position of the slowest lane (known from the release below unless rays were taken just now)
        if (took || first_turn) {
          first_turn = false;
          const int plo = col_wave_min(y.pb);
          if (plo != pos) {
            pos = plo;
            if (lane == 0) col_lds_st(&ctl[8 + wave], pos);
          }
        }
        const int need = min(pos + 2 + Q.wstep, npos + 1);
        if (have < need) {
          have = landed_all();
          for (int spins = 0; have < need; ++spins) {
            const int flagged = col_lds_ld(&ctl[0]);
            if (spins > (1 << 22) || flagged) {
              if (!flagged) col_lds_st(&ctl[0], 1);
              have = 0x3ffffff0;
              y.pb = COL_DONE;
              exhausted = true;
              break;
            }
            __builtin_amdgcn_s_sleep(2);
            have = landed_all();
          }
          have = __builtin_amdgcn_readfirstlane(have);
        }
        asm volatile("" ::: "memory");
        // ---- part A: where the sample is, its eight corners; then where the next one is
        const bool act = y.pb < COL_DONE && y.pb + 2 <= have;
        float fx = 0.f, fy = 0.f, fz = 0.f, nsc = y.sc;
        int nbi = y.bi, npb = y.pb;
        bool last = false;
        typename std::conditional<DT == 0, c_v2u, c_v4f>::type rq[8];
        if (act) {
          const float mf = (float)y.m;
          int x0 = 0, x1, y0 = 0, y1, z0 = 0, z1;
          if constexpr (AS != 0) smk_lin_clamp(__fmaf_rn(mf, y.B[0], y.A[0]), P.N[0], x0, x1, fx);
          else { x0 = y.bi; fx = y.sc - (float)y.bi; }
          if constexpr (AS != 1) smk_lin_clamp(__fmaf_rn(mf, y.B[1], y.A[1]), P.N[1], y0, y1, fy);
          else { y0 = y.bi; fy = y.sc - (float)y.bi; }
          if constexpr (AS != 2) smk_lin_clamp(__fmaf_rn(mf, y.B[2], y.A[2]), P.N[2], z0, z1, fz);
          else { z0 = y.bi; fz = y.sc - (float)y.bi; }
          (void)x1; (void)y1; (void)z1;
          const int iu = (AU == 0 ? x0 : y0) - Q.Ou - u0, iv = (AV == 1 ? y0 : z0) - Q.Ov - v0;  // cell inside the column
          const int *te = slot_addr + (y.bi + eoff);
          const unsigned off = __umul24((unsigned)iv, pitch_b) + (unsigned)iu * VB;
          const unsigned a0 = (unsigned)te[0] + off, b0 = (unsigned)te[1] + off;
          col_read8(a0, a0 + pitch_b, b0, b0 + pitch_b, rq);
          if (y.m < y.m_out) {
            nbi = base_slice(y.A, y.B, y.m + 1, nsc);
            npb = psgn * nbi + poff;
          } else {
            last = true;
            npb = COL_DONE;
          }
        }
        // everything this iteration needs of the ring is in registers: release the slots now (a lane whose ray ends takes
        // its next one from the list's head, which the loaders count in themselves)
        {
          const int plo = col_wave_min(npb);
          if (plo != pos && plo < COL_DONE) {
            pos = plo;
            if (lane == 0) col_lds_st(&ctl[8 + wave], pos);
          }
        }
        // ---- part B: interpolate, classify, shade, blend -- the gather kernel's operations in its order
        bool hit = false;
        if (act) {
#define QI(dx, dy, dz) (PERM == 0 ? ((dz) * 4 + (dy) * 2 + (dx)) : PERM == 1 ? ((dy) * 4 + (dz) * 2 + (dx)) : ((dx) * 4 + (dz) * 2 + (dy)))
#define TRI(E)                                                                                                             \
  smk_lerp(smk_lerp(smk_lerp(E(0, 0, 0), E(1, 0, 0), fx), smk_lerp(E(0, 1, 0), E(1, 1, 0), fx), fy),                       \
           smk_lerp(smk_lerp(E(0, 0, 1), E(1, 0, 1), fx), smk_lerp(E(0, 1, 1), E(1, 1, 1), fx), fy), fz)
          float ch0, ch1 = 0.f, ch2 = 0.f, ch3 = 0.f;
          const bool lazy_h = (TF == 1 && Q.fast_tf) || (TF == 2 && Q.use_occ);
          auto tri_h = [&]() -> float {
            if constexpr (DT == 0) {
#define E2(dx, dy, dz) smk_ub(rq[QI(dx, dy, dz)].x, 2)
              return TRI(E2) * SMK_INV255;
#undef E2
            } else {
#define E2(dx, dy, dz) rq[QI(dx, dy, dz)].z
              return TRI(E2);
#undef E2
            }
          };
          if constexpr (DT == 1) {
#define E0(dx, dy, dz) rq[QI(dx, dy, dz)].x
#define E1(dx, dy, dz) rq[QI(dx, dy, dz)].y
            ch0 = TRI(E0);
            if (TF != 0 || SH != 0) ch1 = TRI(E1);
            if ((TF == 2 || (TF == 1 && P.third_axis)) && !lazy_h) ch2 = tri_h();
#undef E1
#undef E0
          } else {
#define E0(dx, dy, dz) smk_ub(rq[QI(dx, dy, dz)].x, 0)
#define E1(dx, dy, dz) smk_ub(rq[QI(dx, dy, dz)].x, 1)
#define E3(dx, dy, dz) smk_ub(rq[QI(dx, dy, dz)].x, 3)
            ch0 = TRI(E0) * SMK_INV255;
            if (TF != 0 || SH != 0) ch1 = TRI(E1) * SMK_INV255;
            if ((TF == 2 || (TF == 1 && P.third_axis)) && !lazy_h) {
              ch2 = tri_h();
              if (P.nelts == 4) ch3 = TRI(E3) * SMK_INV255;
            }
#undef E3
#undef E1
#undef E0
          }
          float4 col;
          ColTexel4 tx4 = {0, 0, 0, 0, 0.f, 0.f};
          if (TF == 1 && Q.fast_tf) {
            int s0, s1, t0, t1;
            float fs, ft;
            smk_lin_clamp(__fmaf_rn(ch0, (float)P.sv, -0.5f), P.sv, s0, s1, fs);
            smk_lin_clamp(__fmaf_rn(ch1, (float)P.sg, -0.5f), P.sg, t0, t1, ft);
            bool maybe = true;
            if (Q.use_occ) maybe = (occ[__mul24(t0, P.occ_roww) + (s0 >> 5)] >> (s0 & 31)) & 1u;
            col.w = 0.0f;
            if (maybe) {
              tx4 = col_tex2d_fetch(P.tf_vg, P.sv, s0, t0, fs, ft);
              col.w = col_tex_chan(tx4, 3);
              if (Q.use_ah) {
                ch2 = tri_h();
                int h0, h1;
                float fh;
                smk_lin_clamp(__fmaf_rn(ch2, (float)P.sv, -0.5f), P.sv, h0, h1, fh);
                col.w *= smk_lerp(ah[h0], ah[h0 + 1], fh) * SMK_INV255;
              }
              col.w = smk_sat(col.w);
            }
            hit = col.w != 0.0f;
          } else if (TF == 2 && Q.use_occ) {
            int s0, s1, t0, t1;
            float fs, ft;
            smk_lin_clamp(__fmaf_rn(ch0, (float)P.s3v, -0.5f), P.s3v, s0, s1, fs);
            smk_lin_clamp(__fmaf_rn(ch1, (float)P.s3g, -0.5f), P.s3g, t0, t1, ft);
            col.w = 0.0f;
            if ((occ[__mul24(t0, P.occ_roww) + (s0 >> 5)] >> (s0 & 31)) & 1u) {
              ch2 = tri_h();
              hit = smk_classify<DT, TF>(P, ch0, ch1, ch2, ch3, col);
            }
          } else {
            hit = smk_classify<DT, TF>(P, ch0, ch1, ch2, ch3, col);
          }
          if (hit) {
            if (TF == 1 && Q.fast_tf) {
              col.x = col_tex_chan(tx4, 0);
              col.y = col_tex_chan(tx4, 1);
              col.z = col_tex_chan(tx4, 2);
            }
            float4 src;
            if (TF == 0) {
              src = col;
            } else if (SH == 0) {
              src = smk_shade_sample<0>(P, col, 0.f, 0.f, 0.f, 0.f);
            } else {
              uint32_t nb[8];
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                if constexpr (DT == 1) nb[k] = __float_as_uint(rq[k].w);
                else nb[k] = rq[k].y;
              }
#define NB(dx, dy, dz) nb[QI(dx, dy, dz)]
              float n0 = smk_nrm(NB(0, 0, 0), NB(1, 0, 0), NB(0, 1, 0), NB(1, 1, 0), NB(0, 0, 1), NB(1, 0, 1), NB(0, 1, 1), NB(1, 1, 1), 0, fx, fy, fz);
              float n1 = smk_nrm(NB(0, 0, 0), NB(1, 0, 0), NB(0, 1, 0), NB(1, 1, 0), NB(0, 0, 1), NB(1, 0, 1), NB(0, 1, 1), NB(1, 1, 1), 1, fx, fy, fz);
              float n2 = smk_nrm(NB(0, 0, 0), NB(1, 0, 0), NB(0, 1, 0), NB(1, 1, 0), NB(0, 0, 1), NB(1, 0, 1), NB(0, 1, 1), NB(1, 1, 1), 2, fx, fy, fz);
#undef NB
              src = smk_shade_sample<SH>(P, col, n0, n1, n2, ch1);
            }
            if (P.blend == SMK_BLEND_MAX) {
              y.C0 = fmaxf(y.C0, src.x);
              y.C1 = fmaxf(y.C1, src.y);
              y.C2 = fmaxf(y.C2, src.z);
              y.C3 = fmaxf(y.C3, src.w);
            } else {
              const float w = 1.0f - y.C3;
              y.C0 = __fmaf_rn(w, src.x, y.C0);
              y.C1 = __fmaf_rn(w, src.y, y.C1);
              y.C2 = __fmaf_rn(w, src.z, y.C2);
              y.C3 = __fmaf_rn(w, src.w, y.C3);
              // exact early termination inside the segment: once A == 1.0f no later sample of this job can change it
              if (y.C3 == 1.0f && !last) {
                last = true;
                npb = COL_DONE;
              }
            }
          }
#undef TRI
#undef QI
        }
        if (Q.counts) {
          n_iters += 1;
          n_act += (unsigned)__popcll(__ballot(y.pb < COL_DONE));
          n_samples += (unsigned)__popcll(__ballot(act));
          n_visible += (unsigned)__popcll(__ballot(hit));
        }
        if (act) {
          y.m += 1;
          y.pb = npb;
          y.sc = nsc;
          y.bi = nbi;
          if (last) flush();  // the ray leaves the job (or is saturated): its segment; the lane is free
        }
      }
      if (lane == 0) col_lds_st(&ctl[8 + wave], COL_DONE);
      if (Q.counts) {
        if (lane == 0) {
          if (n_samples) atomicAdd(&Q.counts[0], (unsigned long long)n_samples);
          if (n_visible) atomicAdd(&Q.counts[1], (unsigned long long)n_visible);
          atomicAdd(&Q.counts[4], (unsigned long long)n_iters);
          atomicAdd(&Q.counts[5], (unsigned long long)n_act);
          atomicAdd(&Q.counts[6], (unsigned long long)n_swit);
          atomicAdd(&Q.counts[7], (unsigned long long)n_swl);
        }
        unsigned seg = n_segments;
        for (int o = 32; o > 0; o >>= 1) seg += __shfl_xor(seg, o);
        if (lane == 0 && seg) atomicAdd(&Q.counts[3], (unsigned long long)seg);
      }
    }
  }
  __syncthreads();
  if (tid == 0 && ctl[0]) *(volatile int *)Q.status = Q.status_tag | ctl[0];
  if (tid == 0 && Q.job_ticks) {
    Q.job_ticks[job] = max((unsigned)__builtin_amdgcn_s_memrealtime() - t_begin, 1u);
    Q.job_ticks[gridDim.x + job] = t_setup - t_begin;   // (developer statistics: the set-up's share, rays listed)
    Q.job_ticks[2 * gridDim.x + job] = (unsigned)nrays;
  }
}

// ---- resolve: a pixel's segments in key order (front to back), mask cleared for the next frame
__global__ __launch_bounds__(256) void smk_k_cols_resolve(const float4 *layers, unsigned long long *masks, int mask_words, size_t npix,
                                                          float4 *out, int use_max) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  float4 C = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int w = 0; w < mask_words; ++w) {
    unsigned long long m = masks[p * mask_words + w];
    if (!m) continue;
    masks[p * mask_words + w] = 0ull;
    while (m) {
      const int k = __builtin_ctzll(m);
      m &= m - 1;
      const float4 s = layers[(size_t)(w * 64 + k) * npix + p];
      if (use_max) {
        C = make_float4(fmaxf(C.x, s.x), fmaxf(C.y, s.y), fmaxf(C.z, s.z), fmaxf(C.w, s.w));
      } else {
        const float wgt = 1.0f - C.w;
        C.x = __fmaf_rn(wgt, s.x, C.x);
        C.y = __fmaf_rn(wgt, s.y, C.y);
        C.z = __fmaf_rn(wgt, s.z, C.z);
        C.w = __fmaf_rn(wgt, s.w, C.w);
      }
    }
  }
  out[p] = C;
}

// ---- layout build: one thread per voxel of the column layout, gathered from the native [z][y][x] stored box
template <class V, int PERM>
__global__ __launch_bounds__(256) void smk_k_cols_build(const V *src, char *dst, int Dx, int Dy, int Dz, int CW, int CH, int ncu, int ncv,
                                                        int slice_bytes) {
  constexpr int AS = PERM == 0 ? 2 : (PERM == 1 ? 1 : 0);
  constexpr int AU = PERM == 2 ? 1 : 0;
  constexpr int AV = PERM == 0 ? 1 : 2;
  const int D[3] = {Dx, Dy, Dz};
  const int Ds = D[AS], Du = D[AU], Dv = D[AV];
  const int iw = CW + 1, ih = CH + 1;
  const size_t per_slice = (size_t)iw * ih;
  const size_t total = (size_t)ncu * ncv * Ds * per_slice;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int cx = (int)(t % iw);
  const int cy = (int)((t / iw) % ih);
  const size_t sl_col = t / per_slice;  // (column * Ds + slice)
  const int s = (int)(sl_col % Ds);
  const int col = (int)(sl_col / Ds);
  const int cu = col % ncu, cv = col / ncu;
  const int u = min(cu * CW + cx, Du - 1), v = min(cv * CH + cy, Dv - 1);  // (beyond the box: a copy of the edge, never sampled)
  int X[3];
  X[AU] = u; X[AV] = v; X[AS] = s;
  const V val = src[((size_t)X[2] * Dy + X[1]) * Dx + X[0]];
  *reinterpret_cast<V *>(dst + sl_col * (size_t)slice_bytes + ((size_t)cy * iw + cx) * sizeof(V)) = val;
}

// ------------------------------------------------------------------------------- host side

template <int DT, int SH, int PERM, int TF, int NW, int NL>
static hipError_t launch_cols(const RenderParams &P, const ColParams &Q, size_t lds, int njobs, hipStream_t s) {
  auto k = smk_k_cols<DT, SH, PERM, TF, NW, NL>;
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(k, dim3(njobs), dim3((NW + NL) * 64), lds, s, P, Q);
  return hipGetLastError();
}

// workgroup shapes: {consumer waves, loader waves}
struct ColShape { int nw, nl; };
static const ColShape kColShapes[] = {{15, 1}, {14, 2}};

template <int DT, int SH, int PERM, int TF>
static hipError_t dispatch_shape(const RenderParams &P, const ColParams &Q, int shape, size_t lds, int njobs, hipStream_t s) {
  switch (shape) {
    case 0: return launch_cols<DT, SH, PERM, TF, 15, 1>(P, Q, lds, njobs, s);
    case 1: return launch_cols<DT, SH, PERM, TF, 14, 2>(P, Q, lds, njobs, s);
  }
  return hipErrorInvalidValue;
}

template <int DT, int SH, int TF>
static hipError_t dispatch_perm(const RenderParams &P, const ColParams &Q, int perm, int shape, size_t lds, int njobs, hipStream_t s) {
  switch (perm) {
    case 0: return dispatch_shape<DT, SH, 0, TF>(P, Q, shape, lds, njobs, s);
    case 1: return dispatch_shape<DT, SH, 1, TF>(P, Q, shape, lds, njobs, s);
    case 2: return dispatch_shape<DT, SH, 2, TF>(P, Q, shape, lds, njobs, s);
  }
  return hipErrorInvalidValue;
}

static void cols_free_layout(ColLayout &L) {
  if (L.d) (void)hipFree(L.d);
  L = ColLayout();
}

void smk_cols_free(ColsAux *aux) {
  for (int k = 0; k < 3; ++k) cols_free_layout(aux->lay[k]);
  if (aux->d_layers) (void)hipFree(aux->d_layers);
  if (aux->d_masks) (void)hipFree(aux->d_masks);
  if (aux->d_ticks) (void)hipFree(aux->d_ticks);
  if (aux->d_counts) (void)hipFree(aux->d_counts);
  *aux = ColsAux();
}

void smk_cols_drop_layouts(ColsAux *aux) {
  for (int k = 0; k < 3; ++k) cols_free_layout(aux->lay[k]);
}

static void host_ray_cols(const RenderParams &P, double fi, double fj, double A[3], double B[3]) {
  const smk_raycoef &rc = P.rc;
  const double px = fi * (double)rc.pxs + (double)rc.pxl, py = fj * (double)rc.pys + (double)rc.pyl;
  for (int a = 0; a < 3; ++a) {
    A[a] = px * rc.Ax[a] + py * rc.Ay[a] + rc.Ac[a];
    B[a] = px * rc.Bx[a] + py * rc.By[a] + rc.Bc[a];
  }
}

static bool inv3(const double m[9], double o[9]) {
  const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  if (!(fabs(det) > 1e-300)) return false;
  const double id = 1.0 / det;
  o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return true;
}

// plan + launch; hipErrorNotSupported (and *why) when the frame must use another kernel
hipError_t smk_launch_cols(RenderParams P, int dtype, int tf_mode, int shade_kind, int knobs, const void *vox_native, ColsAux *aux,
                           int *status_word, const char **why, hipStream_t s) {
  *why = nullptr;
  const int opt_shape = knobs & 0xff, opt_ns = (knobs >> 8) & 0xff, opt_cl = (knobs >> 16) & 0xfff, opt_wstep = (knobs >> 28) & 0x7;
  if (tf_mode < 0 || tf_mode > 2) { *why = "no classification mode"; return hipErrorNotSupported; }
  if (tf_mode == 0 && (!P.tlut || P.tlut_size < 1)) { *why = "no colour table"; return hipErrorNotSupported; }
  if (tf_mode == 0) shade_kind = 0;
  if (tf_mode == 1 && (!P.tf_vg || P.sv < 2 || P.sg < 2)) { *why = "transfer function smaller than 2x2"; return hipErrorNotSupported; }
  if (tf_mode == 2 && (!P.tf3d || P.s3v < 1 || P.s3g < 1 || P.s3h < 1)) { *why = "no 3-D table"; return hipErrorNotSupported; }
  if (P.pert_on) { *why = "perturbation"; return hipErrorNotSupported; }
  if (P.blend == SMK_BLEND_BACK_TO_FRONT) { *why = "back-to-front blend (columns stream front to back)"; return hipErrorNotSupported; }
  if (P.depth) { *why = "first-hit depth requested"; return hipErrorNotSupported; }
  if (P.cplane_on) { *why = "free clip plane"; return hipErrorNotSupported; }
  if (dtype == 1 && !P.n_in_w) { *why = "4-channel f32 voxels"; return hipErrorNotSupported; }
  if (P.rc.nplanes <= 0) { *why = "no planes"; return hipErrorNotSupported; }
  for (int a = 0; a < 3; ++a) {
    if (P.D[a] < 2 || P.N[a] < 2) { *why = "volume thinner than 2 voxels"; return hipErrorNotSupported; }
    if (!(P.lo[a] <= P.hin[a])) { *why = "region is empty"; return hipErrorNotSupported; }
  }
  if (P.W > 16384 || P.H > 16384) { *why = "viewport larger than 16384"; return hipErrorNotSupported; }

  // principal axis and marching direction from the central ray; every ray must share them
  double Ac[3], Bc[3];
  host_ray_cols(P, P.W * 0.5, P.H * 0.5, Ac, Bc);
  int as = 0;
  for (int a = 1; a < 3; ++a)
    if (fabs(Bc[a]) > fabs(Bc[as])) as = a;
  const int perm = as == 2 ? 0 : (as == 1 ? 1 : 2);
  const int au = perm == 2 ? 1 : 0, av = perm == 0 ? 1 : 2;
  const int dir = Bc[as] > 0 ? 1 : -1;
  double slope_u = 0, slope_v = 0;
  for (int c = 0; c < 4; ++c) {
    double A[3], B[3];
    host_ray_cols(P, (c & 1) ? P.W : 0.0, (c & 2) ? P.H : 0.0, A, B);
    if (!(B[as] * dir > 0) || fabs(B[as]) < 1e-12) { *why = "rays do not share a marching direction"; return hipErrorNotSupported; }
    slope_u = std::max(slope_u, fabs(B[au] / B[as]));
    slope_v = std::max(slope_v, fabs(B[av] / B[as]));
  }
  if (slope_u > 3.0 || slope_v > 3.0) { *why = "view too oblique for the principal axis"; return hipErrorNotSupported; }

  ColParams Q;
  memset(&Q, 0, sizeof Q);
  Q.dir = dir;
  Q.Ou = P.O[au]; Q.Ov = P.O[av]; Q.Os = P.O[as];
  Q.Du = P.D[au]; Q.Dv = P.D[av]; Q.Ds = P.D[as];
  // voxel -> continuous pixel.  X + .5 = E' + tau d(px, py) with d = Bc + px Bx + py By (per unit of dtau) and
  // A = E + tau0/dtau * B: (px, py, 1) tau/dtau = G^-1 (X - E), G = [Bx By Bc]
  {
    const smk_raycoef &rc = P.rc;
    const double G[9] = {rc.Bx[0], rc.By[0], rc.Bc[0], rc.Bx[1], rc.By[1], rc.Bc[1], rc.Bx[2], rc.By[2], rc.Bc[2]};
    double Gi[9];
    if (!inv3(G, Gi)) { *why = "degenerate projection"; return hipErrorNotSupported; }
    const double k = (double)rc.tau0 / (double)rc.dtau;
    const double E[3] = {rc.Ac[0] - k * rc.Bc[0], rc.Ac[1] - k * rc.Bc[1], rc.Ac[2] - k * rc.Bc[2]};
    double row[3][4];
    for (int r = 0; r < 3; ++r) {
      for (int a = 0; a < 3; ++a) row[r][a] = Gi[3 * r + a];
      row[r][3] = -(Gi[3 * r] * E[0] + Gi[3 * r + 1] * E[1] + Gi[3 * r + 2] * E[2]);
    }
    // the sign of the homogeneous coordinate: positive in front of the eye (tau / dtau has dtau's sign)
    const double sgn = rc.dtau > 0 ? 1.0 : -1.0;
    for (int a = 0; a < 4; ++a) {
      Q.Mx[a] = (float)(sgn * (row[0][a] - (double)rc.pxl * row[2][a]) / (double)rc.pxs);
      Q.My[a] = (float)(sgn * (row[1][a] - (double)rc.pyl * row[2][a]) / (double)rc.pys);
      Q.Mw[a] = (float)(sgn * row[2][a]);
    }
    // scale so that w ~ 1 at the volume's centre (keeps the kernel's "w > 0" test well away from rounding)
    const double cx = 0.5 * P.N[0], cy = 0.5 * P.N[1], cz = 0.5 * P.N[2];
    const double wc = Q.Mw[0] * cx + Q.Mw[1] * cy + Q.Mw[2] * cz + Q.Mw[3];
    if (!(wc > 0)) { *why = "volume centre behind the eye"; return hipErrorNotSupported; }
    for (int a = 0; a < 4; ++a) {
      Q.Mx[a] = (float)(Q.Mx[a] / wc);
      Q.My[a] = (float)(Q.My[a] / wc);
      Q.Mw[a] = (float)(Q.Mw[a] / wc);
    }
  }
  // ---- workgroup shape and column size.  Every ray that is inside a column at one slice position wants a lane: cells x
  // rays per cell (largest where the volume is nearest to the eye) must stay below the consumer lanes.
  const int vb = dtype == 0 ? 8 : 16;
  auto project = [&](double u, double v, double sc, double &x, double &y) -> bool {
    double X[3];
    X[au] = u; X[av] = v; X[as] = sc;
    const double w = Q.Mw[0] * X[0] + Q.Mw[1] * X[1] + Q.Mw[2] * X[2] + Q.Mw[3];
    if (!(w > 1e-9)) return false;
    x = (Q.Mx[0] * X[0] + Q.Mx[1] * X[1] + Q.Mx[2] * X[2] + Q.Mx[3]) / w;
    y = (Q.My[0] * X[0] + Q.My[1] * X[1] + Q.My[2] * X[2] + Q.My[3]) / w;
    return true;
  };
  // pixels per (u, v) cell of a slice at the volume's corners and centre
  double dens = 0, flux = 0;
  for (int c = 0; c < 9; ++c) {
    const double u = c == 8 ? 0.5 * P.N[au] : ((c & 1) ? P.N[au] - 0.5 : -0.5), v = c == 8 ? 0.5 * P.N[av] : ((c & 2) ? P.N[av] - 0.5 : -0.5),
                 sc = c == 8 ? 0.5 * P.N[as] : ((c & 4) ? P.N[as] - 0.5 : -0.5);
    double x0, y0, x1, y1, x2, y2, x3, y3;
    if (!project(u, v, sc, x0, y0) || !project(u + 1, v, sc, x1, y1) || !project(u, v + 1, sc, x2, y2) || !project(u, v, sc + 1, x3, y3)) {
      *why = "volume reaches behind the eye";
      return hipErrorNotSupported;
    }
    dens = std::max(dens, fabs((x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0)));
    flux = std::max(flux, std::max(fabs(x3 - x0), fabs(y3 - y0)));
  }
  (void)flux;
  if (!(dens > 1e-9)) { *why = "degenerate projection"; return hipErrorNotSupported; }
  int shape = opt_shape ? opt_shape - 1 : 0;
  if (shape < 0 || shape >= (int)(sizeof kColShapes / sizeof kColShapes[0])) { *why = "no such workgroup shape"; return hipErrorNotSupported; }
  const ColShape &S = kColShapes[shape];
  const int lanes = S.nw * 64;
  // LDS: ring + ray list + entry positions + slot table + two histograms + control + alpha_H + occupancy bitmap
  const bool three = P.third_axis && P.tf_h;
  const int use_ah = (tf_mode == 1 && three && P.nelts <= 3 && P.sv >= 2 && P.sv <= 1024) ? 1 : 0;
  const int fast_tf = (tf_mode == 1 && (!three || use_ah)) ? 1 : 0;
  const size_t occ_bytes = tf_mode == 1 ? (size_t)P.occ_roww * P.sg * 4 : tf_mode == 2 ? (size_t)P.occ_roww * P.s3g * 4 : 0;
  const int use_occ = (P.tf_occ && occ_bytes > 0 && occ_bytes <= 16384 && (fast_tf || tf_mode == 2)) ? 1 : 0;
  const size_t fixed = (size_t)COL_MAX_RAYS * 9 + (size_t)3 * (COL_MAX_CL + 4) * 4 + 32 * 4 + (use_ah ? (size_t)P.sv * 4 : 0) + (use_occ ? occ_bytes : 0) + 64;
  const size_t lds_cap = 160 * 1024;
  const int want_slots = opt_ns ? opt_ns : 6;
  ColLayout &LY = aux->lay[perm];
  const int cells_u = Q.Du - 1, cells_v = Q.Dv - 1;
  const double fill = (aux->opt_fill > 0 ? aux->opt_fill : 92) * 0.01;  // of the lanes, at the densest place (the set-up checks every job exactly and reports, see the kernel)
  auto fits = [&](int cw, int ch, int slots) -> bool {
    if ((double)cw * ch * dens > fill * lanes) return false;
    const size_t sb = (((size_t)(cw + 1) * (ch + 1) * vb) + 15) & ~(size_t)15;
    if (sb * slots + fixed > lds_cap) return false;
    if ((sb + 1023) / 1024 * 2 > 63) return false;  // two slices in flight per loader within the vmcnt range
    return true;
  };
  bool reuse = LY.d && LY.Du == Q.Du && LY.Dv == Q.Dv && LY.Ds == Q.Ds && LY.src == vox_native && LY.vb == vb;
  if (reuse) {
    // an existing layout is kept while its columns fit the lanes and are not wastefully small for the view
    reuse = fits(LY.CW, LY.CH, 3) && ((double)LY.CW * LY.CH * dens > 0.45 * lanes || (LY.CW >= cells_u && LY.CH >= cells_v));
  }
  if (!reuse) {
    // columns as balanced divisions of the box: the squarest pair that fits, most cells first (least halo)
    int bw = 0, bh = 0;
    double best = 1e300;
    for (int ncu = 1; ncu <= cells_u; ++ncu) {
      const int cw = (cells_u + ncu - 1) / ncu;
      if (cw > 255) continue;
      if (ncu > 1 && (cells_u + ncu - 2) / (ncu - 1) == cw) continue;  // (same width as with one column fewer)
      for (int ncv = 1; ncv <= cells_v; ++ncv) {
        const int ch = (cells_v + ncv - 1) / ncv;
        if (ch > 255) continue;
        if (ncv > 1 && (cells_v + ncv - 2) / (ncv - 1) == ch) continue;
        if (!fits(cw, ch, want_slots)) continue;
        const double over = (double)ncu * (cw + 1) * (double)ncv * (ch + 1) / ((double)cells_u * cells_v);
        if (over < best) { best = over; bw = cw; bh = ch; }
      }
    }
    if (!bw) { *why = "no column size fits the lanes (view too close)"; return hipErrorNotSupported; }
    const int ncu = (cells_u + bw - 1) / bw, ncv = (cells_v + bh - 1) / bh;
    const size_t sb = (((size_t)(bw + 1) * (bh + 1) * vb) + 15) & ~(size_t)15;
    const size_t bytes = (size_t)ncu * ncv * Q.Ds * sb;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (LY.d) cols_free_layout(LY);
    if (bytes + ((size_t)2 << 30) > free_b + 0) {
      // make room: the other axes' layouts go first
      for (int k = 0; k < 3; ++k)
        if (k != perm) cols_free_layout(aux->lay[k]);
      (void)hipMemGetInfo(&free_b, &total_b);
      if (bytes + ((size_t)1 << 30) > free_b) { *why = "no memory for the column layout"; return hipErrorNotSupported; }
    }
    void *d = nullptr;
    if (hipMalloc(&d, bytes + 4096) != hipSuccess) { (void)hipGetLastError(); *why = "no memory for the column layout"; return hipErrorNotSupported; }
    LY.d = d; LY.bytes = bytes; LY.CW = bw; LY.CH = bh; LY.ncu = ncu; LY.ncv = ncv; LY.Du = Q.Du; LY.Dv = Q.Dv; LY.Ds = Q.Ds;
    LY.slice_bytes = (int)sb; LY.src = vox_native; LY.vb = vb;
    const size_t total = (size_t)ncu * ncv * Q.Ds * (size_t)(bw + 1) * (bh + 1);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (total / 256 > 0x7fffffffull) { *why = "volume too large for the layout builder"; return hipErrorNotSupported; }
#define BUILD(V, R) hipLaunchKernelGGL((smk_k_cols_build<V, R>), dim3(blocks), dim3(256), 0, s, (const V *)vox_native, (char *)d, P.D[0], P.D[1], P.D[2], bw, bh, ncu, ncv, (int)sb)
    if (dtype == 0) { if (perm == 0) BUILD(uint2, 0); else if (perm == 1) BUILD(uint2, 1); else BUILD(uint2, 2); }
    else { if (perm == 0) BUILD(float4, 0); else if (perm == 1) BUILD(float4, 1); else BUILD(float4, 2); }
#undef BUILD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    ++aux->builds;
  }
  Q.lay = (const char *)LY.d;
  Q.CW = LY.CW; Q.CH = LY.CH; Q.ncu = LY.ncu; Q.ncv = LY.ncv;
  Q.slice_bytes = LY.slice_bytes;
  Q.n_ch = (Q.slice_bytes + 1023) / 1024;
  {
    const int last_units = Q.slice_bytes / 16 - 64 * (Q.n_ch - 1);
    Q.last_mask = last_units >= 64 ? ~0ull : ((1ull << last_units) - 1ull);
  }
  int nslots = (int)((lds_cap - fixed) / (size_t)Q.slice_bytes);
  if (opt_ns) nslots = std::min(nslots, opt_ns);
  nslots = std::min(nslots, 12);
  if (nslots < 3) { *why = "column slice does not fit LDS three times"; return hipErrorNotSupported; }
  Q.nslots = nslots;
  Q.maxfly = std::max(1, std::min(aux->opt_fly > 0 ? aux->opt_fly : 2, (nslots - 2) / S.nl));
  while (Q.maxfly > 1 && Q.n_ch * Q.maxfly > 63) --Q.maxfly;  // (the counted vmcnt wait takes an immediate < 64)
  if (Q.n_ch > 63) { *why = "column slice needs more than 63 DMA instructions"; return hipErrorNotSupported; }
  Q.wstep = opt_wstep ? opt_wstep - 1 : (nslots >= 7 ? 2 : nslots >= 5 ? 1 : 0);
  Q.take_min = aux->opt_take_min > 0 ? aux->opt_take_min : 16;
  Q.take_wait = aux->opt_take_wait > 0 ? aux->opt_take_wait - 1 : 3;
  const int npos_total = Q.Ds - 1;
  int clmax = opt_cl ? std::min(opt_cl, COL_MAX_CL) : 128;
  clmax = std::max(clmax, 4);
  Q.nck = (npos_total + clmax - 1) / clmax;
  Q.CL = (npos_total + Q.nck - 1) / Q.nck;
  Q.nck = (npos_total + Q.CL - 1) / Q.CL;
  Q.nkeys = Q.ncu + Q.ncv + Q.nck - 2;
  Q.mask_words = (Q.nkeys + 63) / 64;
  if (Q.mask_words > 8) { *why = "more than 512 segment keys"; return hipErrorNotSupported; }
  Q.use_ah = use_ah; Q.use_occ = use_occ; Q.fast_tf = fast_tf;
  const size_t npix = (size_t)P.W * P.H;
  const size_t lay_bytes = (size_t)Q.nkeys * npix * 16;
  if (lay_bytes > aux->layers_cap) {
    if (aux->d_layers) (void)hipFree(aux->d_layers);
    aux->d_layers = nullptr; aux->layers_cap = 0;
    if (hipMalloc(&aux->d_layers, lay_bytes) != hipSuccess) { (void)hipGetLastError(); *why = "no memory for the segment layers"; return hipErrorNotSupported; }
    aux->layers_cap = lay_bytes;
  }
  const size_t mask_bytes = npix * Q.mask_words * 8;
  if (mask_bytes > aux->masks_cap || aux->masks_dirty) {
    if (mask_bytes > aux->masks_cap) {
      if (aux->d_masks) (void)hipFree(aux->d_masks);
      aux->d_masks = nullptr; aux->masks_cap = 0;
      if (hipMalloc(&aux->d_masks, mask_bytes) != hipSuccess) { (void)hipGetLastError(); *why = "no memory for the segment masks"; return hipErrorNotSupported; }
      aux->masks_cap = mask_bytes;
    }
    hipError_t e = hipMemsetAsync(aux->d_masks, 0, aux->masks_cap, s);
    if (e != hipSuccess) return e;
    aux->masks_dirty = false;
  }
  aux->mask_words_last = Q.mask_words;
  const int njobs = Q.ncu * Q.ncv * Q.nck;
  if (njobs > aux->ticks_cap) {
    if (aux->d_ticks) (void)hipFree(aux->d_ticks);
    aux->d_ticks = nullptr; aux->ticks_cap = 0;
    if (hipMalloc((void **)&aux->d_ticks, (size_t)njobs * 12) != hipSuccess) { (void)hipGetLastError(); *why = "no memory"; return hipErrorNotSupported; }
    aux->ticks_cap = njobs;
  }
  if (!aux->d_counts) {
    if (hipMalloc((void **)&aux->d_counts, 8 * 8) != hipSuccess) { (void)hipGetLastError(); *why = "no memory"; return hipErrorNotSupported; }
  }
  Q.layers = (float4 *)aux->d_layers;
  Q.masks = (unsigned long long *)aux->d_masks;
  Q.status = status_word;
  Q.status_tag = aux->status_tag;
  Q.job_ticks = aux->d_ticks;
  Q.counts = aux->want_counts ? aux->d_counts : nullptr;
  if (Q.counts) {
    hipError_t e = hipMemsetAsync(aux->d_counts, 0, 64, s);
    if (e != hipSuccess) return e;
  }
  aux->njobs_last = njobs;
  aux->last = Q.CW | (Q.CH << 8) | (nslots << 16) | (shape << 24);
  aux->last_stream_bytes = (double)njobs / Q.nck * ((double)npos_total + Q.nck) * Q.slice_bytes;
  Q.ring_bytes = (int)std::max((size_t)nslots * Q.slice_bytes, (size_t)COL_MAX_RAYS * 16);  // (set-up scratch: the unsorted rays)
  const size_t lds = (size_t)Q.ring_bytes + fixed;
  if (lds > lds_cap) { *why = "column job does not fit LDS"; return hipErrorNotSupported; }
  if (aux->frame_ev0) {
    hipError_t e = hipEventRecord(aux->frame_ev0, s);
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipErrorInvalidValue;
#define CASE(D, T, SHK) \
  if (dtype == D && tf_mode == T && shade_kind == SHK) e = dispatch_perm<D, SHK, T>(P, Q, perm, shape, lds, njobs, s);
  CASE(0, 1, 0) CASE(0, 1, 1) CASE(1, 1, 0) CASE(1, 1, 1) CASE(0, 2, 1) CASE(1, 2, 1) CASE(0, 0, 0)
#undef CASE
  if (e == hipErrorInvalidValue) { *why = "no column-stream instance for this mode"; return hipErrorNotSupported; }
  if (e != hipSuccess) return e;
  aux->masks_dirty = true;  // (until the resolve pass below has run; it clears what it reads)
  hipLaunchKernelGGL(smk_k_cols_resolve, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, (const float4 *)aux->d_layers,
                     (unsigned long long *)aux->d_masks, Q.mask_words, npix, P.out, P.blend == SMK_BLEND_MAX ? 1 : 0);
  e = hipGetLastError();
  if (e == hipSuccess) aux->masks_dirty = false;
  return e;
}
