// smk_shadow.hip -- half-angle-slicing shadows.
//
// Replaces R8kVolRen3D's shadow mode: the slice axis half-way between view and light direction
// (R8kVolRen3D.cpp:296-326), volShadow's two draws per slice polygon (:1651-1868: the slice into the
// frame buffer with the light buffer bound as texture 5, then the same slice into the light buffer), the
// light-buffer fragment shader (:2991-3180) and the eye shader's `1 - shadow` term (:2928-2934).
//
// Three forms of the same arithmetic, sample placement by the fma chains of smk_shadowcoef /
// SmkShadowRays, evaluated identically by the CPU checker:
//   * the default (round 3): the LIGHT MARCH below -- a light-buffer texel depends on itself alone from slice to
//     slice, so the light pass is one march per texel that keeps every slice's buffer -- and the eye pass as an
//     ordinary frame of the ray-marchers (smk_slab.hip / smk_gather.hip, SHD instances) over the half-angle slices,
//     which looks each sample's slice up (smk_shadow_term, smk_device.h).  Two launches.
//   * a launch per slice (option shadow_march 0; rounds 1-2): the eye pass and the light pass of a slice side by
//     side in one grid, the light buffer ping-ponging as the reference's two pbuffers do (the eye pass of slice k
//     reads the buffer as slices < k left it, the light pass writes the other one -- including the texels the slice
//     does not cover, which the reference leaves two slices stale).
//   * all slices in ONE cooperative launch with a grid barrier between them (option shadow_fused; measured slower).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "smk_device.h"

struct ShadowSlice {
  smk_shadowcoef sc;
  const float4 *Lprev;  // [LB][LB] light buffer after the previous slices
  float4 *Lnext;        // ... after this one
  float lnum;           // numerator of this slice's light-ray parameter, fma(k, ldnum, lnum0)
  int k;                // the slice, 1..nslices in the light's order
  int eye_bx, eye_blocks, light_bx;  // 16x16-pixel blocks: eye grid width, eye block count, light grid width
};

// the 8 corners of voxel coordinate p and the interpolated data channels (same arithmetic as kernel G)
template <int DT, int TF, bool SHADE>
__device__ __forceinline__ void shadow_fetch(const RenderParams &P, float p0, float p1, float p2, float &ch0, float &ch1,
                                             float &ch2, float &ch3, float &n0, float &n1, float &n2) {
  int x0, x1, y0, y1, z0, z1;
  float fx, fy, fz;
  smk_lin_clamp(p0, P.N[0], x0, x1, fx);
  smk_lin_clamp(p1, P.N[1], y0, y1, fy);
  smk_lin_clamp(p2, P.N[2], z0, z1, fz);
  const int Dx = P.D[0], Dy = P.D[1];
  size_t r00 = ((size_t)z0 * Dy + y0) * Dx, r10 = ((size_t)z0 * Dy + y1) * Dx;
  size_t r01 = ((size_t)z1 * Dy + y0) * Dx, r11 = ((size_t)z1 * Dy + y1) * Dx;
  const size_t c0 = (size_t)x0, c1 = (size_t)x1;
  SmkCorner k000, k100, k010, k110, k001, k101, k011, k111;
  if constexpr (DT == 0) {
    smk_load_pair_u8(P, r00 + c0, r00 + c1, k000, k100);
    smk_load_pair_u8(P, r10 + c0, r10 + c1, k010, k110);
    smk_load_pair_u8(P, r01 + c0, r01 + c1, k001, k101);
    smk_load_pair_u8(P, r11 + c0, r11 + c1, k011, k111);
  } else {
    k000 = smk_load_corner<DT>(P, r00 + c0); k100 = smk_load_corner<DT>(P, r00 + c1);
    k010 = smk_load_corner<DT>(P, r10 + c0); k110 = smk_load_corner<DT>(P, r10 + c1);
    k001 = smk_load_corner<DT>(P, r01 + c0); k101 = smk_load_corner<DT>(P, r01 + c1);
    k011 = smk_load_corner<DT>(P, r11 + c0); k111 = smk_load_corner<DT>(P, r11 + c1);
  }
  const float sc = DT == 0 ? SMK_INV255 : 1.0f;
  ch0 = SMK_TRI(c0);
  ch1 = SMK_TRI(c1);
  ch2 = ch3 = 0.f;
  if (DT == 0) {
    ch0 *= sc;
    ch1 *= sc;
  }
  if (TF == 2 || (TF == 1 && P.third_axis)) {
    ch2 = SMK_TRI(c2);
    if (DT == 0) ch2 *= sc;
    if (P.nelts == 4) {
      ch3 = SMK_TRI(c3);
      if (DT == 0) ch3 *= sc;
    }
  }
  if (SHADE) {
    n0 = smk_nrm(k000.nb, k100.nb, k010.nb, k110.nb, k001.nb, k101.nb, k011.nb, k111.nb, 0, fx, fy, fz);
    n1 = smk_nrm(k000.nb, k100.nb, k010.nb, k110.nb, k001.nb, k101.nb, k011.nb, k111.nb, 1, fx, fy, fz);
    n2 = smk_nrm(k000.nb, k100.nb, k010.nb, k110.nb, k001.nb, k101.nb, k011.nb, k111.nb, 2, fx, fy, fz);
  }
}

// the table's occupancy bit for this sample's (v, g) base texel (2-D table: smk_api.hip refresh_tf2d; dense 3-D table: folded
// over the sheets, smk_set_tf3d): clear => the lookup returns alpha == 0 exactly, so nine samples in ten skip it
template <int TF>
__device__ __forceinline__ bool shadow_maybe_visible(const RenderParams &P, float ch0, float ch1) {
  if (!P.tf_occ) return true;
  const int sv = TF == 2 ? P.s3v : P.sv, sg = TF == 2 ? P.s3g : P.sg;
  int s0, s1, t0, t1;
  float fs, ft;
  smk_lin_clamp(__fmaf_rn(ch0, (float)sv, -0.5f), sv, s0, s1, fs);
  smk_lin_clamp(__fmaf_rn(ch1, (float)sg, -0.5f), sg, t0, t1, ft);
  return (P.tf_occ[t0 * P.occ_roww + (s0 >> 5)] >> (s0 & 31)) & 1u;
}

// bilinear lookup of the light buffer; texels outside it are 0 (the rest of the pbuffer stays cleared)
template <bool COH = false>
__device__ __forceinline__ void shadow_lookup(const float4 *L, int LB, float lx, float ly, float out[3]) {
  const float fx0 = floorf(lx - 0.5f), fy0 = floorf(ly - 0.5f);
  const float fx = (lx - 0.5f) - fx0, fy = (ly - 0.5f) - fy0;
  out[0] = out[1] = out[2] = 0.0f;
  if (!(fx0 >= -1.0f && fx0 < (float)LB && fy0 >= -1.0f && fy0 < (float)LB)) return;  // (also NaN)
  const int x0 = (int)fx0, y0 = (int)fy0;
  float4 t[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = x0 + (q & 1), y = y0 + (q >> 1);
    t[q] = (x >= 0 && x < LB && y >= 0 && y < LB) ? shadow_ld4<COH>(L + (size_t)y * LB + x) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  out[0] = smk_lerp(smk_lerp(t[0].x, t[1].x, fx), smk_lerp(t[2].x, t[3].x, fx), fy);
  out[1] = smk_lerp(smk_lerp(t[0].y, t[1].y, fx), smk_lerp(t[2].y, t[3].y, fx), fy);
  out[2] = smk_lerp(smk_lerp(t[0].z, t[1].z, fx), smk_lerp(t[2].z, t[3].z, fx), fy);
}

// brick flags (smk_bricks.hip): a sample whose cell lies in a brick with a clear flag is exactly transparent under the
// current table -- nothing for the eye, nothing for the light buffer -- so its eight corners are not fetched
__device__ __forceinline__ bool shadow_brick_empty(const RenderParams &P, float p0, float p1, float p2) {
  if (P.bricks == nullptr) return false;
  int x0, x1, y0, y1, z0, z1;
  float fx, fy, fz;
  smk_lin_clamp(p0, P.N[0], x0, x1, fx);
  smk_lin_clamp(p1, P.N[1], y0, y1, fy);
  smk_lin_clamp(p2, P.N[2], z0, z1, fz);
  return !P.bricks[((size_t)(z0 >> SMK_BRICK_LOG2) * P.nbr[1] + (size_t)(y0 >> SMK_BRICK_LOG2)) * P.nbr[0] + (size_t)(x0 >> SMK_BRICK_LOG2)];
}

// one eye pixel of one slice: the sample on the pixel's ray in this slice, shaded under the light buffer as the previous
// slices left it, blended into the frame
// The fused kernel's slices meet at a barrier INSIDE one launch: what one workgroup writes of the frame and the light buffer
// another, possibly on another XCD with its own L2, reads a slice later.  Those two buffers are therefore accessed with
// device-scope (relaxed atomic) loads and stores, which go past the non-coherent cache levels -- the voxels and tables keep
// their cache lines across the barrier (a device-wide cache write-back + invalidate per slice, which is what a plain
// grid.sync() does, cost 130 us per slice: every slice re-read its voxels from memory).
template <bool COH>
__device__ __forceinline__ float4 shadow_ld4(const float4 *p) {
  if (!COH) return *p;
  const float *f = reinterpret_cast<const float *>(p);
  float4 v;
  v.x = __hip_atomic_load(f + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.y = __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.z = __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.w = __hip_atomic_load(f + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
template <bool COH>
__device__ __forceinline__ void shadow_st4(float4 *p, float4 v) {
  if (!COH) { *p = v; return; }
  float *f = reinterpret_cast<float *>(p);
  __hip_atomic_store(f + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(f + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(f + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(f + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int DT, int TF, int SH, bool COH = false>
__device__ __forceinline__ void shadow_eye_pixel(const RenderParams &P, const ShadowSlice &Q, int i, int j) {
  const smk_shadowcoef &sc = Q.sc;
  if (i >= P.W || j >= P.H) return;
  // the sample of this pixel's ray in slice k: plane m of the ray, counted from the eye (SmkShadowRays, smk_ray_AB) -- the
  // very chain the ray-marchers evaluate
  const float px = __fmaf_rn((float)i + 0.5f, P.rc.pxs, P.rc.pxl), py = __fmaf_rn((float)j + 0.5f, P.rc.pys, P.rc.pyl);
  float A[3], B[3], tauA, dtau;
  if (!smk_ray_AB(P, px, py, A, B, tauA, dtau)) return;
  const int m = P.sh.dk > 0 ? Q.k - P.sh.k0 : P.sh.k0 - Q.k;
  if (!smk_tau_ok(tauA, dtau, m)) return;
  float p[3];
  bool in = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p[a] = __fmaf_rn((float)m, B[a], A[a]);
    in = in && p[a] >= P.lo[a] && p[a] <= P.hi[a];  // (the box, 2^-10 voxels wide: smk_api.hip)
  }
  if (!in) return;
  if (P.cplane_on && !(__fmaf_rn(p[0], P.cplane[0], __fmaf_rn(p[1], P.cplane[1], __fmaf_rn(p[2], P.cplane[2], P.cplane[3]))) >= 0.0f)) return;
  const size_t o = (size_t)j * P.W + i;
  float4 C = shadow_ld4<COH>(P.out + o);
  if (sc.front_to_back && C.w == 1.0f) return;  // exact: every later weight (1-A) is 0
  if (shadow_brick_empty(P, p[0], p[1], p[2])) return;
  float ch0, ch1, ch2, ch3, n0 = 0.f, n1 = 0.f, n2 = 0.f;
  shadow_fetch<DT, TF, SH != 0>(P, p[0], p[1], p[2], ch0, ch1, ch2, ch3, n0, n1, n2);
  float4 col;
  if (!shadow_maybe_visible<TF>(P, ch0, ch1)) return;
  if (!smk_classify<DT, TF>(P, ch0, ch1, ch2, ch3, col)) return;
  // (the light-buffer texels depend on the position alone and could be requested before the voxels are classified,
  //  shortening the chain of dependent gathers; measured: 12.28 vs 10.95 ms per 512-slice frame -- nine lookups in
  //  ten are then made for transparent samples, and the slice is bound by gather throughput, not by latency)
  const float lw = __fmaf_rn(p[0], sc.Wm[0], __fmaf_rn(p[1], sc.Wm[1], __fmaf_rn(p[2], sc.Wm[2], sc.Wm[3])));
  const float lxx = __fmaf_rn(p[0], sc.Xm[0], __fmaf_rn(p[1], sc.Xm[1], __fmaf_rn(p[2], sc.Xm[2], sc.Xm[3])));
  const float lyy = __fmaf_rn(p[0], sc.Ym[0], __fmaf_rn(p[1], sc.Ym[1], __fmaf_rn(p[2], sc.Ym[2], sc.Ym[3])));
  float shadow[3];
  shadow_lookup<COH>(Q.Lprev, sc.LB, __fmaf_rn(__fdiv_rn(lxx, lw), sc.lscale, sc.lbias),
                     __fmaf_rn(__fdiv_rn(lyy, lw), sc.lscale, sc.lbias), shadow);
  const float4 src = smk_shade_sample<SH>(P, col, n0, n1, n2, ch1, shadow);
  if (sc.front_to_back) {
    const float w = 1.0f - C.w;
    C.x = __fmaf_rn(w, src.x, C.x);
    C.y = __fmaf_rn(w, src.y, C.y);
    C.z = __fmaf_rn(w, src.z, C.z);
    C.w = __fmaf_rn(w, src.w, C.w);
  } else {
    const float w = 1.0f - src.w;
    C.x = __fmaf_rn(w, C.x, src.x);
    C.y = __fmaf_rn(w, C.y, src.y);
    C.z = __fmaf_rn(w, C.z, src.z);
    C.w = __fmaf_rn(w, C.w, src.w);
  }
  shadow_st4<COH>(P.out + o, C);
}

// one light-buffer texel of one slice: carried over, with the slice's sample on the texel's light ray laid over it
template <int DT, int TF, bool COH = false>
__device__ __forceinline__ void shadow_light_texel(const RenderParams &P, const ShadowSlice &Q, int x, int y) {
  const smk_shadowcoef &sc = Q.sc;
  if (x >= sc.LB || y >= sc.LB) return;
  const size_t o = (size_t)y * sc.LB + x;
  float4 L = shadow_ld4<COH>(Q.Lprev + o);
  const float a = __fmaf_rn((float)x + 0.5f, sc.las, sc.lal), bb = __fmaf_rn((float)y + 0.5f, sc.las, sc.lal);
  const float nG = __fmaf_rn(a, sc.nGx, __fmaf_rn(bb, sc.nGy, sc.nGc));
  const float w = __fdiv_rn(Q.lnum, nG);
  bool in = w > 0.0f && !isinf(w);
  float p[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float G = __fmaf_rn(a, sc.Gx[q], __fmaf_rn(bb, sc.Gy[q], sc.Gc[q]));
    p[q] = __fmaf_rn(w, G, sc.Lc[q]);
    in = in && p[q] >= P.sh.llo[q] && p[q] <= P.sh.lhi[q];
  }
  if (in && P.cplane_on) in = __fmaf_rn(p[0], P.cplane[0], __fmaf_rn(p[1], P.cplane[1], __fmaf_rn(p[2], P.cplane[2], P.cplane[3]))) >= 0.0f;
  if (in && !shadow_brick_empty(P, p[0], p[1], p[2])) {
    float ch0, ch1, ch2, ch3, n0, n1, n2;
    shadow_fetch<DT, TF, false>(P, p[0], p[1], p[2], ch0, ch1, ch2, ch3, n0, n1, n2);
    float4 col;
    if (shadow_maybe_visible<TF>(P, ch0, ch1) && smk_classify<DT, TF>(P, ch0, ch1, ch2, ch3, col)) {
      // LERP r0.a, r0, r5 (saturated); alpha = sat((1 - a) r5.a + a)   (R8kVolRen3D.cpp:3150-3165)
      const float al = col.w;
      L.x = smk_sat(__fmaf_rn(al, smk_sat(col.x) - L.x, L.x));
      L.y = smk_sat(__fmaf_rn(al, smk_sat(col.y) - L.y, L.y));
      L.z = smk_sat(__fmaf_rn(al, smk_sat(col.z) - L.z, L.z));
      L.w = smk_sat(__fmaf_rn(1.0f - al, L.w, al));
    }
  }
  shadow_st4<COH>(Q.Lnext + o, L);
}

template <int DT, int TF, int SH>
__global__ __launch_bounds__(256) void smk_k_shadow_slice(const RenderParams P, const ShadowSlice Q) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a workgroup = 16x16 pixels, each wave an 8x8 sub-tile (compact footprints in the volume)
  const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
  if ((int)blockIdx.x < Q.eye_blocks) {
    shadow_eye_pixel<DT, TF, SH>(P, Q, ((int)blockIdx.x % Q.eye_bx) * 16 + lx, ((int)blockIdx.x / Q.eye_bx) * 16 + ly);
  } else {
    const int b = (int)blockIdx.x - Q.eye_blocks;
    shadow_light_texel<DT, TF>(P, Q, (b % Q.light_bx) * 16 + lx, (b / Q.light_bx) * 16 + ly);
  }
}

// ALL slices in one launch: a grid that fits the chip at once (cooperative launch: the runtime refuses one that does
// not), every workgroup takes its share of each slice's eye and light blocks, and a grid-wide barrier stands where the
// launch boundary was -- the light buffer slice k reads is complete, and visible on every XCD, before anybody starts
// slice k + 1.  The same arithmetic per pixel and texel as the per-slice launches: identical frames and light buffers.
// 512 launches of ~11 us each were the larger part of a frame with shadows (5.7 ms); the barrier is a fence and one
// atomic round per workgroup.
template <int DT, int TF, int SH>
__global__ __launch_bounds__(256) void smk_k_shadow_fused(const RenderParams P, ShadowSlice Q, float4 *L0, float4 *L1, unsigned *barrier) {
  const smk_shadowcoef &sc = Q.sc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lx = (wave & 1) * 8 + (lane & 7), ly = (wave >> 1) * 8 + (lane >> 3);
  const int nlb = Q.light_bx * Q.light_bx, nblocks = Q.eye_blocks + nlb;
  for (int k = 1; k <= sc.nslices; ++k) {
    Q.k = k;
    Q.lnum = __fmaf_rn((float)k, sc.ldnum, sc.lnum0);
    Q.Lprev = (k & 1) ? L0 : L1;
    Q.Lnext = (k & 1) ? L1 : L0;
    // (the light blocks first: every one of them has work -- a texel is always carried over -- while most eye blocks of a
    //  slice leave at once)
    for (int b = (int)blockIdx.x; b < nblocks; b += (int)gridDim.x) {
      if (b < nlb) shadow_light_texel<DT, TF, true>(P, Q, (b % Q.light_bx) * 16 + lx, (b / Q.light_bx) * 16 + ly);
      else {
        const int e = b - nlb;
        shadow_eye_pixel<DT, TF, SH, true>(P, Q, (e % Q.eye_bx) * 16 + lx, (e / Q.eye_bx) * 16 + ly);
      }
    }
    // grid barrier: every wave's device-scope stores have been acknowledged (vmcnt 0) before its workgroup signs in; one
    // counter, monotone (slice k waits for k * gridDim.x arrivals); the cooperative launch guarantees that every workgroup
    // is resident, the spin is bounded all the same (a barrier that cannot complete leaves the frame wrong, never the GPU hung)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      // two levels: the workgroups of an XCD (blockIdx % 8) count into their own word, the last of each XCD into the common
      // one, and everybody polls the common word -- 8 + gridDim.x / 8 arrivals per word instead of gridDim.x on one
      const unsigned xcd = blockIdx.x & 7u, per_xcd = (gridDim.x + 7u - xcd) / 8u;
      const unsigned mine = __hip_atomic_fetch_add(barrier + 16 * (1 + xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      if (mine == (unsigned)k * per_xcd) __hip_atomic_fetch_add(barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)k * min(gridDim.x, 8u);
      for (int spins = 0; spins < (1 << 22); ++spins) {
        if (__hip_atomic_load(barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
  }
}

// ---- the two marches (round 3, the default).  A texel of the light buffer depends on ITSELF alone from slice to slice
// (shadow_light_texel reads Lprev[o], writes Lnext[o]); only the eye pass reads the buffer at an arbitrary position.  So the
// recurrence over slices is a ray-march per TEXEL along its light ray -- no launch boundary needed -- provided every slice's
// light buffer is kept for the eye pass to look up: hist[k][LB][LB], k = 0..nslices (288 GB of HBM: 512 slices of a 512^2
// buffer are 2.1 GB).  With the history in memory an eye pixel depends on itself alone, too: the eye pass is a ray-march per
// PIXEL -- an ordinary frame of the ray-marchers (slice-ring or gather kernel) over the half-angle slices, whose shading looks
// the light buffer of the sample's slice up (smk_shadow_term).  Light buffers are bit-identical to the per-slice form's;
// frames too where the eye pass keeps the blend's order (a light behind the viewer, no depth segments), and differ by the
// association of the blend otherwise.
//
// Conservative range of slices whose sample E + tau(k) D, tau(k) = fma(k, dnum, num0) / den, can lie inside the volume with
// tau > 0: solved in real arithmetic on a box widened by 0.05 voxels, +- 2 slices; anything doubtful gives the whole range.
// It only brackets: the exact per-sample tests of the per-slice form decide inside it.
__device__ __forceinline__ void shadow_k_range(const RenderParams &P, float num0, float dnum, float den, const float D[3],
                                               const float E[3], int n, int &k0, int &k1) {
  k0 = 1;
  k1 = n;
  float ta = 0.0f, tb = __int_as_float(0x7f800000);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float lo = P.sh.llo[a] - 0.05f, hi = P.sh.lhi[a] + 0.05f;
    if (fabsf(D[a]) > 1e-20f) {
      const float inv = 1.0f / D[a];
      const float t1 = (lo - E[a]) * inv, t2 = (hi - E[a]) * inv;
      ta = fmaxf(ta, fminf(t1, t2));
      tb = fminf(tb, fmaxf(t1, t2));
    } else if (!(E[a] >= lo && E[a] <= hi)) {
      k1 = 0;  // never inside
      return;
    }
  }
  if (!(ta <= tb)) {  // (NaN included: but then nothing passes the per-sample test either... keep the whole range for NaN)
    if (ta > tb) k1 = 0;
    return;
  }
  const float mg = 1e-4f * (fabsf(ta) + (isinf(tb) ? 0.0f : fabsf(tb))) + 1e-30f;
  ta -= mg;
  tb += mg;
  if (!(fabsf(dnum) > 0.0f) || !(fabsf(den) > 0.0f) || isinf(den) || isinf(dnum)) return;
  // k = (tau den - num0) / dnum
  const float ka = (ta * den - num0) / dnum;
  const float kb = isinf(tb) ? ((den / dnum) > 0.0f ? __int_as_float(0x7f800000) : -__int_as_float(0x7f800000)) : (tb * den - num0) / dnum;
  if (ka != ka || kb != kb) return;
  const float klo = fminf(ka, kb) - 2.0f, khi = fmaxf(ka, kb) + 2.0f;
  if (klo > (float)n || khi < 1.0f) {
    k1 = 0;
    return;
  }
  k0 = max(1, (int)floorf(fmaxf(klo, 1.0f)));
  k1 = min(n, (int)ceilf(fminf(khi, (float)n)));
}

// blocks -> 16x16 tiles, XCD-aware: block b runs on XCD b % 8, which takes a contiguous run of the tiles
__device__ __forceinline__ bool shadow_tile_of_block(int bid, int ntiles, int &tile) {
  const int per = (ntiles + 7) >> 3;
  const int k = bid >> 3;
  tile = (bid & 7) * per + k;
  return k < per && tile < ntiles;
}

// The light march: hist[k] = the light buffer after slices 1..k (hist[0] = cleared).  A texel's recurrence over the slices
// is sequential, but what it blends -- the classified sample of its light ray in each slice -- is not: a wave takes 8 texels
// x 8 CONSECUTIVE slices (lane = texel + 8 * slice), fetches and classifies its 64 samples at once (one chain of dependent
// gathers per 8 slices instead of one per slice: a thread per texel walking 512 slices took 1.8 ms for a 512^2 buffer), then
// lays the 8 slices' samples over the running value in order -- 8 short steps on values passed between lanes, every lane of a
// texel computing the same running value, lane s keeping it as slice s leaves it -- and stores 8 buffers' worth.  The same
// operations in the same order per texel as a launch per slice: bit-identical light buffers.
template <int DT, int TF>
__global__ __launch_bounds__(256) void smk_k_shadow_light_march(const RenderParams P, const ShadowSlice Q, float4 *hist, long long hstride) {
  const smk_shadowcoef &sc = Q.sc;
  // blocks of 8 x 4 texels (a wave = 8 texels of one row), dealt to the XCDs in contiguous runs
  // (8 waves side by side -- 64 x 1 texels, a block's stores into one buffer one run of memory -- measured slower: 1.25 vs 1.0 ms)
  const int bx = (sc.LB + 7) >> 3, by = (sc.LB + 3) >> 2;
  int tile;
  if (!shadow_tile_of_block((int)blockIdx.x, bx * by, tile)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = lane & 7, sl = lane >> 3;
  const int x = (tile % bx) * 8 + t, y = (tile / bx) * 4 + wave;
  const bool live = x < sc.LB && y < sc.LB;
  const size_t nl = (size_t)hstride, o = (size_t)y * sc.LB + x;  // (buffers `hstride` texels apart)
  const float a = __fmaf_rn((float)x + 0.5f, sc.las, sc.lal), bb = __fmaf_rn((float)y + 0.5f, sc.las, sc.lal);
  const float nG = __fmaf_rn(a, sc.nGx, __fmaf_rn(bb, sc.nGy, sc.nGc));
  float G[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) G[q] = __fmaf_rn(a, sc.Gx[q], __fmaf_rn(bb, sc.Gy[q], sc.Gc[q]));
  int k0, k1;
  shadow_k_range(P, sc.lnum0, sc.ldnum, nG, G, sc.Lc, sc.nslices, k0, k1);
  if (!live) k1 = 0;
  float4 L = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && sl == 0) hist[o] = L;
  // where this lane's sample of slice k is, and -- requested one turn of the loop ahead, so that the first of the turn's
  // dependent round trips is over when the turn begins -- the flag of its brick (shadow_brick_empty's test)
  float p[3] = {0.f, 0.f, 0.f};
  auto place = [&](int k) -> bool {
    if (!(k >= k0 && k <= k1 && k <= sc.nslices)) return false;
    const float w = __fdiv_rn(__fmaf_rn((float)k, sc.ldnum, sc.lnum0), nG);
    bool in = w > 0.0f && !isinf(w);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      p[q] = __fmaf_rn(w, G[q], sc.Lc[q]);
      in = in && p[q] >= P.sh.llo[q] && p[q] <= P.sh.lhi[q];
    }
    if (in && P.cplane_on) in = __fmaf_rn(p[0], P.cplane[0], __fmaf_rn(p[1], P.cplane[1], __fmaf_rn(p[2], P.cplane[2], P.cplane[3]))) >= 0.0f;
    return in;
  };
  auto flag_of = [&](bool in) -> unsigned char {
    if (!in) return 0;
    if (P.bricks == nullptr) return 1;
    int x0, x1, y0, y1, z0, z1;
    float fx, fy, fz;
    smk_lin_clamp(p[0], P.N[0], x0, x1, fx);
    smk_lin_clamp(p[1], P.N[1], y0, y1, fy);
    smk_lin_clamp(p[2], P.N[2], z0, z1, fz);
    return P.bricks[((size_t)(z0 >> SMK_BRICK_LOG2) * P.nbr[1] + (size_t)(y0 >> SMK_BRICK_LOG2)) * P.nbr[0] + (size_t)(x0 >> SMK_BRICK_LOG2)];
  };
  // the slices some texel of this wave can have a sample in: before them every buffer is the cleared one, behind them the
  // last one -- those turns of the loop only store
  int klo = k1 >= k0 ? k0 : 0x7fffffff, khi = k1 >= k0 ? k1 : -0x7fffffff;
  for (int off = 32; off > 0; off >>= 1) {
    klo = min(klo, __shfl_xor(klo, off));
    khi = max(khi, __shfl_xor(khi, off));
  }
  const int kb_first = klo > khi ? sc.nslices + 1 : 1 + ((klo - 1) & ~7);  // first turn with work (turns start at 1 + 8 n)
  for (int kb = 1; kb < kb_first && kb <= sc.nslices; kb += 8)
    if (live && kb + sl <= sc.nslices) hist[(size_t)(kb + sl) * nl + o] = L;
  // Four turns' worth (32 slices) of positions and brick flags at a time: the flag loads are in flight together -- one round
  // trip per 32 slices where a turn of its own costs one per 8.  (Measured: no gain, 1.00-1.03 ms either way -- the turns that
  // cost are those with samples in flagged bricks, a chain of corner, occupancy and table round trips each; forcing 8 waves
  // per SIMD to overlap more of them spills: 1.0 -> 1.45 ms.)
  int kb = kb_first;
  for (; kb <= sc.nslices && kb <= khi; kb += 32) {
    float pu[4][3];
    unsigned char fl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      fl[u] = flag_of(place(kb + 8 * u + sl));
      pu[u][0] = p[0]; pu[u][1] = p[1]; pu[u][2] = p[2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + 8 * u + sl;
      float4 col = make_float4(0.f, 0.f, 0.f, 0.f);  // this lane's sample: saturated colour and alpha, alpha 0 = nothing to lay over
      if (fl[u]) {
        float ch0, ch1, ch2, ch3, n0, n1, n2;
        shadow_fetch<DT, TF, false>(P, pu[u][0], pu[u][1], pu[u][2], ch0, ch1, ch2, ch3, n0, n1, n2);
        float4 cc;
        if (shadow_maybe_visible<TF>(P, ch0, ch1) && smk_classify<DT, TF>(P, ch0, ch1, ch2, ch3, cc))
          col = make_float4(smk_sat(cc.x), smk_sat(cc.y), smk_sat(cc.z), cc.w);
      }
      float4 mine = L;  // the running value as slice k leaves it (lanes of slices nobody lays anything over: unchanged)
      if (__any(col.w != 0.0f)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int src = t + 8 * s;
          const float al = __shfl(col.w, src), cx = __shfl(col.x, src), cy = __shfl(col.y, src), cz = __shfl(col.z, src);  // (every lane active)
          // LERP r0.a, r0, r5 (saturated); alpha = sat((1 - a) r5.a + a)   (R8kVolRen3D.cpp:3150-3165, as shadow_light_texel)
          if (al != 0.0f) {
            L.x = smk_sat(__fmaf_rn(al, cx - L.x, L.x));
            L.y = smk_sat(__fmaf_rn(al, cy - L.y, L.y));
            L.z = smk_sat(__fmaf_rn(al, cz - L.z, L.z));
            L.w = smk_sat(__fmaf_rn(1.0f - al, L.w, al));
          }
          if (s == sl) mine = L;
        }
      }
      if (live && k <= sc.nslices) hist[(size_t)k * nl + o] = mine;
    }
  }
  for (; kb <= sc.nslices; kb += 8)
    if (live && kb + sl <= sc.nslices) hist[(size_t)(kb + sl) * nl + o] = L;
}

template <int DT, int TF>
static hipError_t run_march(const RenderParams &P, ShadowSlice Q, float4 *hist, long long hstride, hipStream_t s) {
  const smk_shadowcoef &sc = Q.sc;
  const int tiles = ((sc.LB + 7) / 8) * ((sc.LB + 3) / 4);
  const int lblocks = 8 * ((tiles + 7) / 8);
  hipLaunchKernelGGL((smk_k_shadow_light_march<DT, TF>), dim3(lblocks), dim3(256), 0, s, P, Q, hist, hstride);
  return hipGetLastError();
}

// The light march alone: hist = nslices + 1 buffers of [LB][LB] texels, `hstride` texels apart; hist[k] = the light buffer after slices 1..k.  The eye pass is
// then an ordinary frame of the ray-marchers over the half-angle slices (P.sh, smk_api.hip).
hipError_t smk_launch_shadow_march(const RenderParams &P, const smk_shadowcoef &sc, int dtype, int tf_mode, float4 *hist, long long hstride,
                                   hipStream_t s) {
  ShadowSlice Q;
  memset(&Q, 0, sizeof Q);
  Q.sc = sc;
#define CASE(D, T) \
  if (dtype == D && tf_mode == T) return run_march<D, T>(P, Q, hist, hstride, s);
  CASE(0, 1) CASE(0, 2) CASE(1, 1) CASE(1, 2)
#undef CASE
  return hipErrorNotSupported;
}

template <int DT, int TF, int SH>
static hipError_t run(const RenderParams &P, ShadowSlice Q, float4 *L0, float4 *L1, unsigned *barrier, hipStream_t s) {
  const smk_shadowcoef &sc = Q.sc;
  Q.eye_bx = (P.W + 15) / 16;
  Q.eye_blocks = Q.eye_bx * ((P.H + 15) / 16);
  Q.light_bx = (sc.LB + 15) / 16;
  const int blocks = Q.eye_blocks + Q.light_bx * Q.light_bx;
  // one cooperative launch for all slices where the device offers it (SMK_SHADOW_FUSED=0: the per-slice launches)
  // Measured on one MI355X (cfg 3 with shadows, 512 slices): per-slice launches 5.56 ms per frame; the fused launch 12.8 /
  // 16.3 / 20.2 ms with 2 / 4 / 8 workgroups per CU (two-level barrier with back-off; a single counter: 26 / 40 / 50 ms;
  // cooperative_groups' grid.sync() with its device-wide cache write-back and invalidate: 67 ms).  Few resident workgroups
  // walk a slice's 5120 blocks one after the other, each a chain of dependent gathers, where a launch has them all in
  // flight; many make the barrier -- every workgroup polling one word through the fabric -- cost more than the ~11 us launch
  // it replaces.  The hardware's dispatcher IS the cheaper barrier here: the fused form stays an option ("shadow_fused").
  const bool want_fused = (P.lockstep & 256) != 0;
  if (want_fused) {
    int dev = 0, coop = 0, per_cu = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (coop && cus > 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, smk_k_shadow_fused<DT, TF, SH>, 256, 0) == hipSuccess && per_cu > 0) {
      // (a barrier costs with the number of workgroups that meet at it: four per CU hide the gathers' latency, more only wait)
      static const int wgs_per_cu = getenv("SMK_SHADOW_WGS") ? std::max(1, atoi(getenv("SMK_SHADOW_WGS"))) : 4;  // (developer knob)
      const int grid = std::min(blocks, cus * std::min(per_cu, wgs_per_cu));
      RenderParams Pa = P;
      ShadowSlice Qa = Q;
      unsigned *bar = barrier;  // (the context's word, zeroed on the stream before every frame)
      if (bar && hipMemsetAsync(bar, 0, 16 * 9 * 4, s) != hipSuccess) bar = nullptr;
      void *args[] = {(void *)&Pa, (void *)&Qa, (void *)&L0, (void *)&L1, (void *)&bar};
      if (bar) {
      const hipError_t e = hipLaunchCooperativeKernel((const void *)smk_k_shadow_fused<DT, TF, SH>, dim3(grid), dim3(256), args, 0, s);
      if (e == hipSuccess) return hipGetLastError();
      (void)hipGetLastError();  // refused (resources): the per-slice launches below
      }
    }
  }
  for (int k = 1; k <= sc.nslices; ++k) {
    Q.k = k;
    Q.lnum = fmaf((float)k, sc.ldnum, sc.lnum0);
    Q.Lprev = (k & 1) ? L0 : L1;
    Q.Lnext = (k & 1) ? L1 : L0;
    hipLaunchKernelGGL((smk_k_shadow_slice<DT, TF, SH>), dim3(blocks), dim3(256), 0, s, P, Q);
  }
  return hipGetLastError();
}

// L0 must be cleared by the caller; after the call the light buffer is L1 for odd nslices, L0 for even ones
hipError_t smk_launch_shadow(const RenderParams &P, const smk_shadowcoef &sc, int dtype, int tf_mode, int shade_kind,
                             float4 *L0, float4 *L1, unsigned *barrier, hipStream_t s) {
  ShadowSlice Q;
  memset(&Q, 0, sizeof Q);
  Q.sc = sc;
#define CASE(D, T, S) \
  if (dtype == D && tf_mode == T && shade_kind == S) return run<D, T, S>(P, Q, L0, L1, barrier, s);
  CASE(0, 1, 0) CASE(0, 1, 1) CASE(0, 2, 0) CASE(0, 2, 1)
  CASE(1, 1, 0) CASE(1, 1, 1) CASE(1, 2, 0) CASE(1, 2, 1)
#undef CASE
  return hipErrorNotSupported;
}
