// smk_slab.hip -- kernel S: the slice-ring ray-marcher (the fast path for 2-D / separable
// classification without perturbation).
//
// Idea.  The gather kernel (smk_gather.hip) pulls 8 corners per sample through TA/L1: every
// 128-B line is re-requested from L2 several times and waves spend ~87 % of their time in
// s_waitcnt (profiles/r01_*).  Here the volume is streamed instead:
//
//   * the principal axis S of the view (largest |ray direction| component in voxel space) is
//     chosen on the host; U is the memory-contiguous axis, V the third one;
//   * a workgroup owns a TW x TH pixel tile: NW consumer waves (one lane = one ray, a wave = a
//     compact 8x8 sub-tile) plus ONE loader wave;
//   * the loader wave streams, front to back, the (u,v) window of every S-slice the tile's ray
//     bundle crosses -- whole contiguous row pieces, 16 B per lane -- from HBM straight into
//     an LDS ring with LDS-DMA (global_load_lds_dwordx4: no VGPR round trip).  It keeps
//     several slices in flight behind a COUNTED s_waitcnt vmcnt(N) and publishes a `landed`
//     counter; the consumers' transfer-function gathers never wait behind the stream because
//     vmcnt is per wave;
//   * every consumer wave advances on its own (no workgroup barrier in the main loop): it
//     waits for `landed`, takes the samples whose base slice is resident, reading the 8
//     corners from LDS (ds_read_b128 / b64), and publishes its progress; the loader reuses a
//     ring slot once every wave is past it;
//   * RGBA stays in registers front to back; 16 B per pixel leave the kernel.
//
// Each voxel row piece a tile needs is read once per tile; neighbouring tiles share only the
// 1-3 voxel fringe (served by the XCD's L2 because tiles are dealt to XCDs in contiguous runs).
//
// Sample placement, membership and interpolation order are EXACTLY those of the gather kernel
// (same fma chains), so the two kernels and the CPU checker agree bit for bit on positions.
// Reference semantics: see smk_device.h.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "smk_device.h"

// wave-uniform description of one launch
struct SlabParams {
  int perm;                    // 0: S=z (U=x,V=y)  1: S=y (U=x,V=z)  2: S=x (U=y,V=z; x-major copy)
  int au, av, as;              // model-axis index of U, V, S
  long long strideV, strideS;  // voxel strides of the layout in use (U stride is 1)
  int Ou, Ov, Os;              // stored-box origin along U,V,S (global voxel index)
  int Du, Dv, Ds;              // stored-box dims along U,V,S
  int slot_vox;                // LDS voxels per ring slot (a whole number of 64-lane DMA chunks)
  int chunks;                  // DMA wave-instructions per slice (= slot_vox / (64*UPV)), uniform
  int gmax;                    // base slices a consumer may take per step
  int nslots;                  // ring size
  int maxfly;                  // slices the loader keeps in flight (chunks*(maxfly-1) <= 63)
  int dir;                     // +1: rays advance towards +S, -1: towards -S
  int tw, th;                  // pixel tile
  const void *vox;             // layout base (native or x-major)
  int use_ah;                  // third-axis alpha served from a 1-D LDS table (<= 3 channels)
};

#define SLAB_EPS 0.02f
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

template <int DT>
__device__ __forceinline__ SmkCorner slab_corner(const typename VoxT<DT>::type &v) {
  SmkCorner k;
  if (DT == 0) {
    const uint2 &q = reinterpret_cast<const uint2 &>(v);
    k.c0 = smk_ub(q.x, 0);
    k.c1 = smk_ub(q.x, 1);
    k.c2 = smk_ub(q.x, 2);
    k.c3 = smk_ub(q.x, 3);
    k.nb = q.y;
  } else {
    const float4 &q = reinterpret_cast<const float4 &>(v);
    k.c0 = q.x;
    k.c1 = q.y;
    k.c2 = q.z;
    k.c3 = 0.0f;
    k.nb = __float_as_uint(q.w);
  }
  return k;
}

// window of one slice for this tile, in stored-box voxel coordinates; the slot holds it flat,
// row-major with pitch w (so one DMA wave-instruction = 64 consecutive 16-byte units)
struct SlabWin {
  short u0, v0;
  unsigned char w, h;  // window dims (<= 255, host-checked)
  short slot;          // ring slot of the slice
};

// one LDS voxel read as a single ds_read_b128 / b64: the empty asm keeps hipcc from splitting
// the vector load into partial ds_read2_b32 pieces per consumer
__device__ __forceinline__ float4 lds_vox(const float4 *p) {
  float4 v = *p;
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
  return v;
}
__device__ __forceinline__ uint2 lds_vox(const uint2 *p) {
  uint2 v = *p;
  asm volatile("" : "+v"(v.x), "+v"(v.y));
  return v;
}

typedef __attribute__((address_space(3))) void *lds_ptr_t;
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

// NW = consumer waves (64 rays each), NL = loader waves; the block has (NW+NL)*64 threads, the
// last NL waves are loaders: loader l streams DMA chunks l, l+NL, ... of every slice
// second launch-bound = waves per SIMD wanted: workgroups of 5/9/10 waves only double up on a CU
// (2 x 9 waves = 5 on one SIMD) if the kernel stays within 96 VGPRs
template <int DT, int SH, int PERM, int NW, int NL>
__global__ __launch_bounds__((NW + NL) * 64, ((NW + NL) == 9 || (NW + NL) == 5 || (NW + NL) == 10) ? 5 : 4) void smk_k_slab(const RenderParams P, const SlabParams Q) {
  typedef typename VoxT<DT>::type Vox;
  constexpr int UPV = DT == 0 ? 2 : 1;  // voxels per 16-byte DMA unit
  constexpr int NTH = (NW + NL) * 64;
  extern __shared__ __align__(16) unsigned char smem[];
  // LDS carve: ring [nslots][slot_vox] voxels | window table [Ds] | control words | alpha_H
  Vox *ring = reinterpret_cast<Vox *>(smem);
  const int slot_vox = Q.slot_vox;
  SlabWin *wtab = reinterpret_cast<SlabWin *>(smem + (size_t)Q.nslots * slot_vox * sizeof(Vox));
  // control words: [0] smin [1] smax [3] protocol time-out [4..4+NL) landed per loader [8..8+NW) progress
  int *ctl = reinterpret_cast<int *>(wtab + Q.Ds);
  // third-axis alpha as a 1-D table: with <= 3 channels the (H,4th) lookup has t = 0, i.e. row
  // 0 of deptex2 with a zero t-weight, so lerp(row0[s0], row0[s1], fs) is the SAME float
  float *ah = reinterpret_cast<float *>(ctl + 8 + 32);

  int tx, ty;
  if (!smk_tile_of_block(P, blockIdx.x, tx, ty)) return;  // whole workgroup leaves together

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
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
  float A[3], B[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    A[a] = __fmaf_rn(px, rc.Ax[a], __fmaf_rn(py, rc.Ay[a], rc.Ac[a]));
    B[a] = __fmaf_rn(px, rc.Bx[a], __fmaf_rn(py, rc.By[a], rc.Bc[a]));
  }
  // conservative plane range (identical to the gather kernel)
  float tenter = 0.0f, texit = (float)(rc.nplanes - 1);
  bool empty = rc.nplanes <= 0 || !live;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (fabsf(B[a]) > 1e-20f) {
      float inv = 1.0f / B[a];
      float t1 = (P.lo[a] - A[a]) * inv, t2 = (P.hi[a] - A[a]) * inv;
      tenter = fmaxf(tenter, fminf(t1, t2) - 2.0f);
      texit = fminf(texit, fmaxf(t1, t2) + 2.0f);
    } else if (!(A[a] >= P.lo[a] && A[a] <= P.hi[a])) {
      empty = true;
    }
  }
  int m = (int)floorf(fmaxf(tenter, 0.0f));
  int m1 = (int)ceilf(fminf(texit, (float)(rc.nplanes - 1)));
  if (empty || !(tenter <= texit)) m1 = m - 1;

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

  // ---- workgroup slice range
  if (tid == 0) {
    ctl[0] = 0x7fffffff;
    ctl[1] = -0x7fffffff;
    ctl[3] = 0;  // protocol time-out flag (bounded spins)
    ctl[4] = ctl[5] = ctl[6] = ctl[7] = 0;
  }
  if (tid < 32) ctl[8 + tid] = SLAB_DONE;
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
  const int smin = ctl[0], smax = ctl[1];
  const int dir = Q.dir, nslots = Q.nslots;
  // positions p = 0..npos-1 in marching order: base slice b(p) = dir>0 ? smin+p : smax-p;
  // load order q = 0..npos: slice L(q) = dir>0 ? smin+q : smax+1-q; position p reads L(p), L(p+1)
  const int npos = smax - smin + 1;
  // ---- per-slice windows of this tile (every thread fills some table entries): bbox over the
  // tile's 4 corner rays of every position a sample touching slice sl can have (s in
  // [sl-1, sl+1], stretched to the volume faces at the ends)
  if (npos > 0) {
    float cA[4][3], cB[4][3];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int ci = min(tx * Q.tw + ((c & 1) ? Q.tw - 1 : 0), P.W - 1);
      int cj = min(ty * Q.th + ((c & 2) ? Q.th - 1 : 0), P.H - 1);
      float cx = __fmaf_rn((float)ci + 0.5f, rc.pxs, rc.pxl), cy = __fmaf_rn((float)cj + 0.5f, rc.pys, rc.pyl);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        cA[c][a] = __fmaf_rn(cx, rc.Ax[a], __fmaf_rn(cy, rc.Ay[a], rc.Ac[a]));
        cB[c][a] = __fmaf_rn(cx, rc.Bx[a], __fmaf_rn(cy, rc.By[a], rc.Bc[a]));
      }
    }
    if (Q.use_ah)
      for (int e = tid; e < P.sv; e += NTH) ah[e] = smk_ub(P.tf_h[e], 3);
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
      // to stored-box coordinates, clipped to it; rows made of whole 16-byte units
      u0 = max(u0 - Q.Ou, 0);
      v0 = max(v0 - Q.Ov, 0);
      u1 = min(u1 - Q.Ou, Q.Du - 1);
      v1 = min(v1 - Q.Ov, Q.Dv - 1);
      if (UPV == 2) {
        u0 &= ~1;
        u1 |= 1;  // the stored U extent is even (host check)
      }

      int w = max(u1 - u0 + 1, 0), h = max(v1 - v0 + 1, 0);
      if (w * h > slot_vox) h = w > 0 ? slot_vox / w : 0;  // never overrun a slot (host sizes it)
      SlabWin ww;
      ww.u0 = (short)u0;
      ww.v0 = (short)v0;
      ww.w = (unsigned char)w;
      ww.h = (unsigned char)h;
      ww.slot = (short)(q % nslots);
      wtab[e] = ww;
    }
  }
  // every consumer wave announces the first position it needs before anyone moves on
  int bs = (m <= m1) ? base_slice(m) : -0x40000000;
  auto pos_of = [&](int b) -> int { return dir > 0 ? b - smin : smax - b; };
  int pos = SLAB_DONE;
  if (!is_loader && npos > 0) {
    int p0 = (m <= m1) ? pos_of(bs) : SLAB_DONE;
    for (int o = 32; o > 0; o >>= 1) p0 = min(p0, __shfl_xor(p0, o));
    pos = p0;
    if (lane == 0) ctl[8 + wave] = pos;
  }
  __syncthreads();  // table, alpha_H, control words visible; LAST workgroup barrier
  float C0 = 0.f, C1 = 0.f, C2 = 0.f, C3 = 0.f;
  float first = __int_as_float(0x7f800000);

  if (npos > 0) {
    if (is_loader) {
      // ================================ loader wave ============================================
      // Streams load indices q = 0..npos in order.  Slot of q is q % nslots; it may be rewritten
      // once every consumer is past position q - nslots (positions < min progress are done).
      // Every slice issues exactly `chunks` DMA instructions (lanes past the window re-read
      // the window's first unit into the slot's unused tail), so the in-order vmcnt tells
      // which slices have landed: `inflight` slices outstanding <=> vmcnt <= chunks*inflight.
      __builtin_amdgcn_s_setprio(3);  // the stream must never wait for issue slots behind pollers
      const char *gv = reinterpret_cast<const char *>(Q.vox);
      const int chunks = Q.chunks;
      const int mych = (chunks - lid + NL - 1) / NL;  // DMA instructions THIS loader issues per slice
      const unsigned strideVb = (unsigned)(Q.strideV * (long long)sizeof(Vox));  // bytes, < 2^32
      int q = 0, inflight = 0, landed = 0, idle = 0, minp = 0;
      long long t_issue = 0, t_wait = 0, t_idle = 0, t_all = __builtin_amdgcn_s_memtime(), t0_ = 0;
      const bool prof = (P.lockstep & 4) != 0;  // (diagnostic build switch: loader cycle shares)
      auto poll_progress = [&]() -> int {
        int v = SLAB_DONE;
        if (lane < NW) v = raw_lds_b32(&ctl[8 + lane]);
        for (int o = 16; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
        return __builtin_amdgcn_readfirstlane(v);
      };
      while (landed <= npos) {
        // ---- issue while the ring has room (progress is re-polled only when it blocks us)
        bool stop = false;
        if (prof) t0_ = __builtin_amdgcn_s_memtime();
        while (q <= npos && inflight < Q.maxfly) {
          if (q - nslots >= minp) {
            minp = poll_progress();
            if (minp >= SLAB_DONE) {  // every consumer finished: the rest is not needed
              stop = true;
              break;
            }
            if (q - nslots >= minp) break;
          }
          const int sl = (dir > 0 ? smin + q : smax + 1 - q) - Q.Os;
          if (sl >= 0 && sl < Q.Ds) {
            SlabWin w;
            {
              uint2 raw = raw_lds_b64(&wtab[sl]);
              memcpy(&w, &raw, sizeof w);
            }
            const int wu = max(w.w / UPV, 1);  // 16-byte units per window row
            const int n = (w.w / UPV) * w.h;   // units in the window
            Vox *dst = ring + w.slot * slot_vox;
            // wave-uniform 64-bit base of the window + a 32-bit per-lane byte offset
            // (diagnostic bit 8: every slice re-reads slice 0 -> L2-hot stream, isolates issue cost)
            const size_t sl_src = (P.lockstep & 8) ? 0 : (size_t)sl;
            const char *base = gv + (sl_src * Q.strideS + (size_t)w.v0 * Q.strideV + (size_t)w.u0) * sizeof(Vox);
            // this loader's chunks c = lid, lid+NL, ...; lane's unit idx = 64*c + lane =
            // row*wu + col, advanced incrementally by 64*NL units per step
            const int adv = 64 * NL;
            const int qa = adv / wu, ra = adv - qa * wu;
            const int idx0 = lid * 64 + lane;
            const int row0 = idx0 / wu;
            int col = idx0 - row0 * wu;
            unsigned off = (unsigned)row0 * strideVb + (unsigned)col * 16u;
            const unsigned step = (unsigned)qa * strideVb + (unsigned)ra * 16u;
            const unsigned wrap = strideVb - (unsigned)wu * 16u;
            int left = n - idx0;  // > 0 while this lane's unit is inside the window
            for (int c = lid; c < chunks; c += NL) {
              const unsigned o = left > 0 ? off : 0u;
              // LDS address = wave-uniform base + lane*16: the slot image is flat in unit order
              __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + o), (lds_ptr_t)(dst + c * 64 * UPV), 16, 0, 0);
              left -= adv;
              col += ra;
              off += step;
              if (col >= wu) {
                col -= wu;
                off += wrap;
              }
            }
          } else {
            // slice outside the stored box (never read): keep the instruction count uniform
            for (int c = lid; c < chunks; c += NL)
              __builtin_amdgcn_global_load_lds((glb_ptr_t)gv, (lds_ptr_t)(ring + (q % nslots) * slot_vox + c * 64 * UPV), 16, 0, 0);
          }
          ++q;
          ++inflight;
        }
        if (prof) t_issue += __builtin_amdgcn_s_memtime() - t0_;
        if (stop) break;
        if (inflight > 0) {
          // retire the oldest slice in flight: all but the (inflight-1) younger slices' DMAs done
          if (prof) t0_ = __builtin_amdgcn_s_memtime();
          wait_vmcnt(mych * (inflight - 1));
          if (prof) t_wait += __builtin_amdgcn_s_memtime() - t0_;
          --inflight;
          ++landed;
          raw_lds_st_b32(&ctl[4 + lid], landed);
        } else {
          if (prof) t_idle += 200;
          if (++idle > (1 << 22) || raw_lds_b32(&ctl[3])) {  // bounded spin (see consumers)
            raw_lds_st_b32(&ctl[3], 1);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      wait_vmcnt(0);
      if (prof && lane == 0) {
        t_all = __builtin_amdgcn_s_memtime() - t_all;
        P.out[P.W * (size_t)P.H - 1 - blockIdx.x] = make_float4((float)t_issue, (float)t_wait, (float)t_idle, (float)t_all);
      }
    } else {
      // ================================ consumer waves ==========================================
      // slices landed = the slowest loader's count
      auto landed_all = [&]() -> int {
        int v = lds_ld(&ctl[4]);
#pragma unroll
        for (int l = 1; l < NL; ++l) v = min(v, lds_ld(&ctl[4 + l]));
        return v;
      };
      int have = 0;  // cached copy of `landed` (monotonic): re-polled only when it is too small
      while (pos < npos) {
        if (!__any(m <= m1)) break;  // every ray of this wave is finished
        // wait until the slices of position pos (load indices pos, pos+1) have landed; take up
        // to gmax positions if more are already resident
        if (have < pos + 1 + Q.gmax) have = landed_all();
        for (int spins = 0; have < pos + 2; ++spins) {
          if (spins > (1 << 22) || lds_ld(&ctl[3])) {  // bounded: never hang the GPU on a protocol bug
            lds_st(&ctl[3], 1);
            have = 0x3ffffff0;
            pos = npos;
            break;
          }
          __builtin_amdgcn_s_sleep(4);
          have = landed_all();
        }
        if (pos >= npos) break;
        have = __builtin_amdgcn_readfirstlane(have);
        asm volatile("" ::: "memory");  // slot reads stay behind the poll
        const int G = min(min(Q.gmax, have - pos - 1), npos - pos);
        const int b0 = dir > 0 ? smin + pos : smax - pos - G + 1, b1 = b0 + G - 1;
        // windows of slices b0 .. b1+1 (at most 4): wave-uniform, fetched once per step into SGPRs
        // so a sample's address does not start with a dependent LDS round trip
        uint2 we[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint2 raw = *reinterpret_cast<const uint2 *>(&wtab[min(b0 + k, smax + 1) - Q.Os]);
          we[k].x = __builtin_amdgcn_readfirstlane(raw.x);
          we[k].y = __builtin_amdgcn_readfirstlane(raw.y);
        }
        // ---- all samples of this ray whose base slice lies in [b0, b1]
        // (a two-phase variant -- walk to the next alpha>0 sample, then shade all lanes together --
        //  was measured 1.45x SLOWER on the LevWidget scene: hits are not sparse enough there)
        while (true) {
          const bool act = bs >= b0 && bs <= b1 && !(P.lockstep & 2);  // (debug bit 2: stream only)
          if (!__any(act)) break;
          if (act) {
            float p[3];
            p[0] = __fmaf_rn((float)m, B[0], A[0]);
            p[1] = __fmaf_rn((float)m, B[1], A[1]);
            p[2] = __fmaf_rn((float)m, B[2], A[2]);
            // same membership predicate as the gather kernel, evaluated without short-circuit
            // branches (bitwise on lane masks)
            const bool t0 = P.top[0] != 0, t1 = P.top[1] != 0, t2 = P.top[2] != 0;
            bool in = ((p[0] >= P.lo[0]) & ((p[0] < P.hi[0]) | (t0 & (p[0] <= P.hi[0])))) &
                      ((p[1] >= P.lo[1]) & ((p[1] < P.hi[1]) | (t1 & (p[1] <= P.hi[1])))) &
                      ((p[2] >= P.lo[2]) & ((p[2] < P.hi[2]) | (t2 & (p[2] <= P.hi[2]))));
            if (in) {
              int x0, x1, y0, y1, z0, z1;
              float fx, fy, fz;
              smk_lin_clamp(p[0], P.N[0], x0, x1, fx);
              smk_lin_clamp(p[1], P.N[1], y0, y1, fy);
              smk_lin_clamp(p[2], P.N[2], z0, z1, fz);
              const int iu = (AU == 0 ? x0 : y0) - Q.Ou, iv = (AV == 1 ? y0 : z0) - Q.Ov;
              const int is = (AS == 2 ? z0 : (AS == 1 ? y0 : x0));
              const int kk = is - b0;  // 0 .. G-1 (G <= 3)
              const uint2 ea = kk == 0 ? we[0] : (kk == 1 ? we[1] : we[2]);
              const uint2 eb = kk == 0 ? we[1] : (kk == 1 ? we[2] : we[3]);
              SlabWin wa, wb;
              memcpy(&wa, &ea, sizeof wa);
              memcpy(&wb, &eb, sizeof wb);
              // clamp into the resident windows (never alters a result: windows cover the bundle)
              const int ca = min(max(iu - wa.u0, 0), wa.w - 2), ra = min(max(iv - wa.v0, 0), wa.h - 2);
              const int cb = min(max(iu - wb.u0, 0), wb.w - 2), rb = min(max(iv - wb.v0, 0), wb.h - 2);
              const Vox *sa = ring + (wa.slot * slot_vox + ra * wa.w + ca);
              const Vox *sb = ring + (wb.slot * slot_vox + rb * wb.w + cb);
              // corners q[ds][dv][du]
              SmkCorner q000 = slab_corner<DT>(lds_vox(sa)), q001 = slab_corner<DT>(lds_vox(sa + 1));
              SmkCorner q010 = slab_corner<DT>(lds_vox(sa + wa.w)), q011 = slab_corner<DT>(lds_vox(sa + wa.w + 1));
              SmkCorner q100 = slab_corner<DT>(lds_vox(sb)), q101 = slab_corner<DT>(lds_vox(sb + 1));
              SmkCorner q110 = slab_corner<DT>(lds_vox(sb + wb.w)), q111 = slab_corner<DT>(lds_vox(sb + wb.w + 1));
              // back to model order k<dx><dy><dz>: the lerp order (x, y, z) is the gather kernel's
#define KX(dx, dy, dz)                                                                                   \
  (PERM == 0 ? (dz ? (dy ? (dx ? q111 : q110) : (dx ? q101 : q100)) : (dy ? (dx ? q011 : q010) : (dx ? q001 : q000))) \
   : PERM == 1 ? (dy ? (dz ? (dx ? q111 : q110) : (dx ? q101 : q100)) : (dz ? (dx ? q011 : q010) : (dx ? q001 : q000))) \
               : (dx ? (dz ? (dy ? q111 : q110) : (dy ? q101 : q100)) : (dz ? (dy ? q011 : q010) : (dy ? q001 : q000))))
              const SmkCorner &k000 = KX(0, 0, 0), &k100 = KX(1, 0, 0), &k010 = KX(0, 1, 0), &k110 = KX(1, 1, 0);
              const SmkCorner &k001 = KX(0, 0, 1), &k101 = KX(1, 0, 1), &k011 = KX(0, 1, 1), &k111 = KX(1, 1, 1);
#undef KX
              const float sc = DT == 0 ? SMK_INV255 : 1.0f;
              float ch0 = SMK_TRI(c0), ch1 = SMK_TRI(c1), ch2 = 0.f, ch3 = 0.f;
              if (DT == 0) {
                ch0 *= sc;
                ch1 *= sc;
              }
              if (P.third_axis) {
                ch2 = SMK_TRI(c2);
                if (DT == 0) ch2 *= sc;
                if (DT == 0 && P.nelts == 4) ch3 = SMK_TRI(c3) * sc;
              }
              float4 col;
              bool hit;
              if (Q.use_ah) {
                col = smk_tex2d(P.tf_vg, P.sv, P.sg, ch0, ch1);
                int h0, h1;
                float fh;
                smk_lin_clamp(__fmaf_rn(ch2, (float)P.sv, -0.5f), P.sv, h0, h1, fh);
                col.w *= smk_lerp(ah[h0], ah[h1], fh) * SMK_INV255;
                col.w = smk_sat(col.w);
                hit = col.w != 0.0f;
              } else {
                hit = smk_classify<DT, 1>(P, ch0, ch1, ch2, ch3, col);
              }
              if (hit) {
                float4 src;
                if (SH == 0) {
                  src = smk_shade_sample<0>(P, col, 0.f, 0.f, 0.f, 0.f);
                } else {
                  float n0 = smk_nrm(k000.nb, k100.nb, k010.nb, k110.nb, k001.nb, k101.nb, k011.nb, k111.nb, 0, fx, fy, fz);
                  float n1 = smk_nrm(k000.nb, k100.nb, k010.nb, k110.nb, k001.nb, k101.nb, k011.nb, k111.nb, 1, fx, fy, fz);
                  float n2 = smk_nrm(k000.nb, k100.nb, k010.nb, k110.nb, k001.nb, k101.nb, k011.nb, k111.nb, 2, fx, fy, fz);
                  src = smk_shade_sample<SH>(P, col, n0, n1, n2, ch1);
                }
                float w = 1.0f - C3;
                if (first == __int_as_float(0x7f800000)) first = __fmaf_rn((float)m, rc.dtau, rc.tau0) * P.znear;
                C0 = __fmaf_rn(w, src.x, C0);
                C1 = __fmaf_rn(w, src.y, C1);
                C2 = __fmaf_rn(w, src.z, C2);
                C3 = __fmaf_rn(w, src.w, C3);
                // exact early termination: once A == 1.0f every later weight (1-A) is exactly 0,
                // so no later sample can change C or A (nor the first-hit depth)
                if (C3 == 1.0f) m1 = m;
              }
            }
            ++m;
            bs = (m <= m1) ? base_slice(m) : -0x40000000;
          }
        }
        // done with positions [pos, pos+G): their lower slices may be recycled
        pos += G;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every slot read has returned
        if (lane == 0) lds_st(&ctl[8 + wave], pos);
      }
      if (lane == 0) lds_st(&ctl[8 + wave], SLAB_DONE);
    }
  }
  if (live) {
    size_t o = (size_t)j * P.W + i;
    P.out[o] = make_float4(C0, C1, C2, C3);
    if (P.depth) P.depth[o] = first;
  }
}

// ------------------------------------------------------------------------------- host side

static void host_ray(const RenderParams &P, int i, int j, double A[3], double B[3]) {
  const smk_raycoef &rc = P.rc;
  float px = fmaf((float)i + 0.5f, rc.pxs, rc.pxl), py = fmaf((float)j + 0.5f, rc.pys, rc.pyl);
  for (int a = 0; a < 3; ++a) {
    A[a] = fmaf(px, rc.Ax[a], fmaf(py, rc.Ay[a], rc.Ac[a]));
    B[a] = fmaf(px, rc.Bx[a], fmaf(py, rc.By[a], rc.Bc[a]));
  }
}

template <int DT, int SH, int PERM, int NW, int NL>
static hipError_t launch_slab(const RenderParams &P, const SlabParams &Q, size_t lds, hipStream_t s) {
  auto k = smk_k_slab<DT, SH, PERM, NW, NL>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(k, dim3(8 * P.tiles_per_xcd), dim3((NW + NL) * 64), lds, s, P, Q);
  return hipGetLastError();
}

// plan + launch; returns hipErrorNotSupported when the configuration must use the gather kernel
hipError_t smk_launch_slab(RenderParams P, int dtype, int shade_kind, int opt_T, int opt_tile, int forced,
                           const void *vox_native, const void *vox_xmajor, const char **why, hipStream_t s) {
  *why = nullptr;
  if (P.pert_on) { *why = "perturbation"; return hipErrorNotSupported; }
  if (P.N[0] < 2 || P.N[1] < 2 || P.N[2] < 2) { *why = "volume thinner than 2 voxels"; return hipErrorNotSupported; }
  if (dtype == 1 && !P.n_in_w) { *why = "4-channel f32 voxels"; return hipErrorNotSupported; }
  if (P.rc.nplanes <= 0) { *why = "no planes"; return hipErrorNotSupported; }

  // principal axis from the central ray
  double Ac[3], Bc[3];
  host_ray(P, P.W / 2, P.H / 2, Ac, Bc);
  int as = 0;
  for (int a = 1; a < 3; ++a)
    if (fabs(Bc[a]) > fabs(Bc[as])) as = a;
  SlabParams Q;
  memset(&Q, 0, sizeof Q);
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
  if (Q.Ds > 4096) { *why = "more than 4096 slices"; return hipErrorNotSupported; }
  // u8 voxels are 8 B: the DMA moves 16-B units, so rows must start and end on even voxels
  if (dtype == 0 && ((Q.Du & 1) || (Q.strideV & 1) || (Q.strideS & 1))) { *why = "odd U extent for 8-byte voxels"; return hipErrorNotSupported; }

  // workgroup shape: consumer waves are 8x8 pixel sub-tiles; NL loader waves.
  //   light windows (<= ~16 B per ray and slice): 32x16 tile, 8+1 waves, two workgroups per CU
  //   heavy windows (1024^3 f32 at a voxel per pixel: 26 B): 32x24 tile, 12+4 waves, one per CU
  // (one loader wave issues ~1 KiB of LDS-DMA per ~250 cycles incl. its address arithmetic:
  //  profiles/r01_*; so the stream needs several loader waves per CU to approach HBM speed)
  struct Cfg { int tw, th, nl; };
  Cfg cfgs[2] = {{32, 16, 1}, {32, 24, 4}};
  int ncfg = 2;
  if (opt_tile == 1) { cfgs[0] = {16, 16, 1}; ncfg = 1; }
  else if (opt_tile == 2) { cfgs[0] = {32, 24, 4}; ncfg = 1; }
  else if (opt_tile == 3) { cfgs[0] = {24, 16, 2}; ncfg = 1; }
  else if (opt_tile == 4) { cfgs[0] = {64, 8, 1}; ncfg = 1; }
  else if (opt_tile == 5) { cfgs[0] = {32, 16, 2}; ncfg = 1; }
  else if (opt_tile == 6) { cfgs[0] = {32, 16, 1}; ncfg = 1; }
  else if (opt_tile == 8) { cfgs[0] = {32, 16, 4}; ncfg = 1; }
  const int upv = dtype == 0 ? 2 : 1;
  const size_t vb = dtype == 0 ? 8 : 16;
  for (int ci = 0; ci < ncfg; ++ci) {
    const int tw = cfgs[ci].tw, th = cfgs[ci].th, nl = cfgs[ci].nl;
    const int nw = (tw / 8) * (th / 8);
    Q.tw = tw; Q.th = th;
    P.ntx = (P.W + tw - 1) / tw;
    P.nty = (P.H + th - 1) / th;
    P.tiles_per_xcd = (P.ntx * P.nty + 7) / 8;

    // every ray must advance along S in the same direction and not too obliquely; window bound:
    // bundle cross-section extent (corner rays of every tile) at the two S faces + drift over the
    // two-slice interval a window covers + texel pair + eps
    double max_eu = 0, max_ev = 0, max_drift_u = 0, max_drift_v = 0;
    const double sf[2] = {-0.5, (double)P.N[as] - 0.5};
    for (int tyi = 0; tyi < P.nty; ++tyi)
      for (int txi = 0; txi < P.ntx; ++txi) {
        double umin[2] = {1e300, 1e300}, umax[2] = {-1e300, -1e300}, vmin[2] = {1e300, 1e300}, vmax[2] = {-1e300, -1e300};
        for (int c = 0; c < 4; ++c) {
          int cx = std::min(txi * tw + ((c & 1) ? tw - 1 : 0), P.W - 1);
          int cy = std::min(tyi * th + ((c & 2) ? th - 1 : 0), P.H - 1);
          double A[3], B[3];
          host_ray(P, cx, cy, A, B);
          if (!(B[as] * Q.dir > 0) || fabs(B[as]) < 1e-12) { *why = "rays do not share a marching direction"; return hipErrorNotSupported; }
          double du = fabs(B[Q.au] / B[as]), dv = fabs(B[Q.av] / B[as]);
          if (du > 1.5 || dv > 1.5) { *why = "view too oblique for the principal axis"; return hipErrorNotSupported; }
          max_drift_u = std::max(max_drift_u, du);
          max_drift_v = std::max(max_drift_v, dv);
          for (int f = 0; f < 2; ++f) {
            double mm = (sf[f] - A[as]) / B[as];
            double u = A[Q.au] + B[Q.au] * mm, v = A[Q.av] + B[Q.av] * mm;
            umin[f] = std::min(umin[f], u); umax[f] = std::max(umax[f], u);
            vmin[f] = std::min(vmin[f], v); vmax[f] = std::max(vmax[f], v);
          }
        }
        for (int f = 0; f < 2; ++f) {
          max_eu = std::max(max_eu, umax[f] - umin[f]);
          max_ev = std::max(max_ev, vmax[f] - vmin[f]);
        }
      }
    // a window spans s in [j-1, j+1] (2 slices of drift; 2.5 at the faces), + pair + eps + rounding
    int Wu = (int)ceil(max_eu + 2.5 * max_drift_u + 2 * SLAB_EPS) + 3;
    int Wv = (int)ceil(max_ev + 2.5 * max_drift_v + 2 * SLAB_EPS) + 3;
    if (dtype == 0) Wu += 2;  // even alignment of both ends
    Wu = std::min(Wu, Q.Du);
    Wv = std::min(Wv, Q.Dv);
    if (Wu < 2 || Wv < 2) { *why = "degenerate window"; return hipErrorNotSupported; }
    if (Wu > 255 || Wv > 255) { *why = "window too large"; return hipErrorNotSupported; }
    Q.slot_vox = ((Wu * Wv + 64 * upv - 1) / (64 * upv)) * (64 * upv);
    Q.chunks = Q.slot_vox / (64 * upv);
    if (Q.chunks > 63) { *why = "window needs more than 63 DMA chunks"; return hipErrorNotSupported; }
    // light enough for this configuration?  otherwise try the next (heavier-duty) one
    if (ci + 1 < ncfg && (double)Q.chunks * 1024.0 / (nw * 64) > 16.0 * nl) continue;

    Q.use_ah = (P.third_axis && P.nelts <= 3 && P.sv <= 2048) ? 1 : 0;
    Q.gmax = std::min(opt_T > 0 ? opt_T : 3, 3);  // <= 3: a consumer step caches 4 slice windows
    const size_t fixed = (size_t)Q.Ds * sizeof(SlabWin) + (8 + 32) * 4 + 64 + (Q.use_ah ? (size_t)P.sv * 4 : 0);
    // ring: as many slots as fit two workgroups per CU (small tiles) or one (big tiles)
    size_t budget = (nw + nl) * 64 > 640 ? 158 * 1024 : 78 * 1024;
    int ns = (int)((budget - fixed) / ((size_t)Q.slot_vox * vb));
    if (ns > 16) ns = 16;
    if (ns < 4) {
      ns = (int)((158 * 1024 - fixed) / ((size_t)Q.slot_vox * vb));
      if (ns > 8) ns = 8;
    }
    if (ns < 4) { *why = "window does not fit LDS"; return hipErrorNotSupported; }
    Q.nslots = ns;
    if (Q.gmax > ns - 2) Q.gmax = ns - 2;
    const int mych = (Q.chunks + nl - 1) / nl;  // most DMA instructions one loader issues per slice
    Q.maxfly = std::min(ns - 2, 63 / mych + 1);
    if (Q.maxfly < 1) Q.maxfly = 1;
    if (P.wave_w != 8) Q.maxfly = std::max(1, std::min(P.wave_w, Q.maxfly));  // (experiment knob)
    const size_t lds = (size_t)ns * Q.slot_vox * vb + fixed;
#define GO(D, S, R, N, L) \
  if (dtype == D && shade_kind == S && Q.perm == R && nw == N && nl == L) return launch_slab<D, S, R, N, L>(P, Q, lds, s);
#define GO_NW(D, S, R) GO(D, S, R, 4, 1) GO(D, S, R, 6, 2) GO(D, S, R, 8, 1) GO(D, S, R, 8, 2) GO(D, S, R, 8, 4) GO(D, S, R, 12, 4)
#define GO_R(D, S) GO_NW(D, S, 0) GO_NW(D, S, 1) GO_NW(D, S, 2)
    GO_R(0, 0) GO_R(0, 1) GO_R(0, 2) GO_R(1, 0) GO_R(1, 1) GO_R(1, 2)
#undef GO_R
#undef GO_NW
#undef GO
    *why = "no kernel instance for this tile size";
    return hipErrorNotSupported;
  }
  (void)forced;
  *why = "no configuration fits";
  return hipErrorNotSupported;
}
