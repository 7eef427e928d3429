// smk_gather.hip -- kernel G: the generic ray-marcher.  One lane = one ray; voxel corners,
// transfer-function texels and noise texels are gathered straight from HBM/L2 (no staging).
// It covers every mode of the C ABI (u8/f32, 1-/2-/3-D classification, both Phong variants,
// perturbation, sharded regions) and is the fallback the slab-staged kernel (smk_slab.hip)
// defers to.  Replaces the per-slice polygon loop + fragment pipeline of
// VolumeRenderer::render3DVA (VolumeRenderer.cpp:507-741) / NV20VolRen3D::render3DVA (:852-1083).
//
// Wavefront packing: a 256-thread workgroup is a 16x16 pixel tile, each 64-lane wave an 8x8
// sub-tile, so the 64 rays of a wave stay spatially compact (their corner fetches fall in few
// cache lines) and walk the volume together.
#include "smk_device.h"

// SHD: the frame's planes are the half-angle slices of a frame with shadows (SmkShadowRays): the eye pass of smk_shadow.hip
template <int DT, int TF, int SH, bool SHD = false>
__global__ __launch_bounds__(256) void smk_k_gather(const RenderParams P) {
  int tx, ty;
  if (!smk_tile_of_block(P, blockIdx.x, tx, ty)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // wave sub-tile wave_w x (64/wave_w) pixels, blk_w x (4/blk_w) waves per workgroup
  const int wh = 64 / P.wave_w, bh = 4 / P.blk_w;
  const int i = tx * (P.wave_w * P.blk_w) + (wave % P.blk_w) * P.wave_w + (lane % P.wave_w);
  const int j = ty * (wh * bh) + (wave / P.blk_w) * wh + (lane / P.wave_w);
  const bool live = i < P.W && j < P.H;

  const smk_raycoef &rc = P.rc;
  const float px = __fmaf_rn((float)i + 0.5f, rc.pxs, rc.pxl);
  const float py = __fmaf_rn((float)j + 0.5f, rc.pys, rc.pyl);
  float A[3], B[3], tauA, dtau;  // (tauA, dtau: frames with shadows only, see smk_ray_AB)
  const bool ray_ok = smk_ray_AB_t<SHD>(P, px, py, A, B, tauA, dtau);

  // conservative plane range [m0,m1] from a slab test (+-2 planes of slack); the exact
  // per-sample inside test below is what decides membership
  float tenter = 0.0f, texit = (float)(rc.nplanes - 1);
  bool empty = rc.nplanes <= 0 || !ray_ok;
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
  int m0 = (int)floorf(fmaxf(tenter, 0.0f));
  int m1 = (int)ceilf(fminf(texit, (float)(rc.nplanes - 1)));
  if (empty || !(tenter <= texit) || !live) m1 = m0 - 1;
  int mlo = m0, mhi = m1;
  if (P.lockstep) {
    // all 64 rays of the wave visit the same plane in the same iteration: their corner
    // fetches then fall in one thin slab of the volume instead of 64 different depths
    int lo = m1 >= m0 ? m0 : 0x7fffffff, hi = m1 >= m0 ? m1 : -0x7fffffff;
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, __shfl_xor(lo, o));
      hi = max(hi, __shfl_xor(hi, o));
    }
    mlo = lo;
    mhi = hi;
  }

  float C0 = 0.f, C1 = 0.f, C2 = 0.f, C3 = 0.f;
  float first = __int_as_float(0x7f800000);
  SmkNoiseCell ncell;  // (perturbed fetch: the noise cell the FIRST octave's last lookup fell into, smk_noise_cached)
  ncell.key = -1;
  const int Dx = P.D[0], Dy = P.D[1];

  // blend order: front to back (and GL_MAX, which has no order) walk m upwards; back to front
  // (VolumeRenderer.cpp:590) starts at the far plane
  const bool btf = P.blend == SMK_BLEND_BACK_TO_FRONT;
  for (int t = 0, nt = mhi >= mlo ? mhi - mlo + 1 : 0; t < nt; ++t) {  // (an all-empty wave has mlo = INT_MAX, mhi = -INT_MAX)
    const int m = btf ? mhi - t : mlo + t;
    if (!btf && !__any(m <= m1)) break;  // every ray of the wave is past its last plane (or saturated)
    if (m < m0 || m > m1) continue;
    float p0 = __fmaf_rn((float)m, B[0], A[0]);
    float p1 = __fmaf_rn((float)m, B[1], A[1]);
    float p2 = __fmaf_rn((float)m, B[2], A[2]);
    bool in = (p0 >= P.lo[0] && (p0 < P.hi[0] || (P.top[0] && p0 <= P.hi[0]))) &&
              (p1 >= P.lo[1] && (p1 < P.hi[1] || (P.top[1] && p1 <= P.hi[1]))) &&
              (p2 >= P.lo[2] && (p2 < P.hi[2] || (P.top[2] && p2 <= P.hi[2])));
    if (!in) continue;
    // free clip plane (glClipPlane semantics, NV20VolRen3D.cpp:346-357): fragments on its negative side do not exist
    if (P.cplane_on && !(__fmaf_rn(p0, P.cplane[0], __fmaf_rn(p1, P.cplane[1], __fmaf_rn(p2, P.cplane[2], P.cplane[3]))) >= 0.0f)) continue;
    if (SHD && !smk_tau_ok(tauA, dtau, m)) continue;  // (half-angle slices: a sample behind the eye does not exist)
    const float q0 = p0, q1 = p1, q2 = p2;  // (the sample's own position: where its light-buffer lookup is made)

    if (P.pert_on) {
      if (TF != 0 && P.bricks_dil != nullptr) {
        // every brick the displaced fetch can reach from here is flagged empty (smk_api.hip build_params): the sample is
        // exactly transparent wherever the noise sends it -- neither the noise nor the voxels are looked at
        int u0, u1, v0, v1, w0, w1;
        float fu, fv, fw;
        smk_lin_clamp(p0, P.N[0], u0, u1, fu);
        smk_lin_clamp(p1, P.N[1], v0, v1, fv);
        smk_lin_clamp(p2, P.N[2], w0, w1, fw);
        u0 = min(max(u0 - P.O[0], 0), P.D[0] - 1);
        v0 = min(max(v0 - P.O[1], 0), P.D[1] - 1);
        w0 = min(max(w0 - P.O[2], 0), P.D[2] - 1);
        if (!P.bricks_dil[((size_t)(w0 >> SMK_BRICK_LOG2) * P.nbr[1] + (size_t)(v0 >> SMK_BRICK_LOG2)) * P.nbr[0] + (size_t)(u0 >> SMK_BRICK_LOG2)])
          continue;
      }
      // tc' = tc + sum w_m (noise(tc s_m) - .5)   (R8kVolRen3D_cpy.cpp:1590-1595, 3462-3490)
      float t0 = (p0 + 0.5f) * P.invN[0], t1 = (p1 + 0.5f) * P.invN[1], t2 = (p2 + 0.5f) * P.invN[2];
      float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (P.pw[q] == 0.0f) continue;
        float nz[3];
        // (the FIRST octave only: its cell spans 80 voxels, a wave crosses one rarely; the second octave's cells, 7.6 voxels,
        //  change for some lane of a wave nearly every turn -- cached as well, config 5 takes 8.8 ms instead of 8.5)
        if (q == 0 && P.nn_log2 >= 0 && P.nn_log2 <= 10) smk_noise_cached(P, t0 * P.ps[q], t1 * P.ps[q], t2 * P.ps[q], nz, ncell);
        else smk_noise(P, t0 * P.ps[q], t1 * P.ps[q], t2 * P.ps[q], nz);
        o0 = __fmaf_rn(P.pw[q], nz[0] - 0.5f, o0);
        o1 = __fmaf_rn(P.pw[q], nz[1] - 0.5f, o1);
        o2 = __fmaf_rn(P.pw[q], nz[2] - 0.5f, o2);
      }
      p0 = __fmaf_rn(t0 + o0, (float)P.N[0], -0.5f);
      p1 = __fmaf_rn(t1 + o1, (float)P.N[1], -0.5f);
      p2 = __fmaf_rn(t2 + o2, (float)P.N[2], -0.5f);
    }

    int x0, x1, y0, y1, z0, z1;
    float fx, fy, fz;
    smk_lin_clamp(p0, P.N[0], x0, x1, fx);
    smk_lin_clamp(p1, P.N[1], y0, y1, fy);
    smk_lin_clamp(p2, P.N[2], z0, z1, fz);
    // stored box is region + halo; a perturbed fetch may leave it: clamp for memory safety
    // (smk_set_perturb refuses configurations whose halo is too small, so this never alters
    // a result)
    x0 = min(max(x0 - P.O[0], 0), P.D[0] - 1);
    x1 = min(max(x1 - P.O[0], 0), P.D[0] - 1);
    y0 = min(max(y0 - P.O[1], 0), P.D[1] - 1);
    y1 = min(max(y1 - P.O[1], 0), P.D[1] - 1);
    z0 = min(max(z0 - P.O[2], 0), P.D[2] - 1);
    z1 = min(max(z1 - P.O[2], 0), P.D[2] - 1);
    // brick flags (smk_bricks.hip): no sample whose cell lies in a brick with a clear flag can be visible under the
    // current table -- it ends at a clear occupancy bit below, alpha exactly 0 -- so its eight corners are not fetched
    if (TF != 0 && P.bricks != nullptr &&
        !P.bricks[((size_t)(z0 >> SMK_BRICK_LOG2) * P.nbr[1] + (size_t)(y0 >> SMK_BRICK_LOG2)) * P.nbr[0] + (size_t)(x0 >> SMK_BRICK_LOG2)])
      continue;
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
    float ch0 = SMK_TRI(c0), ch1 = 0.f, ch2 = 0.f, ch3 = 0.f;
    if (DT == 0) ch0 *= sc;
    if (TF != 0 || SH != 0) {
      ch1 = SMK_TRI(c1);
      if (DT == 0) ch1 *= sc;
    }
    float4 col;
    if (TF == 2 && P.tf_occ) {
      // dense 3-D table: a clear occupancy bit of the (v, g) base texel (folded over the sheets, smk_set_tf3d) means the
      // eight-texel lookup returns alpha == 0 exactly -- checked before the third channel is even interpolated
      int s0, s1, t0, t1;
      float fs, ft;
      smk_lin_clamp(__fmaf_rn(ch0, (float)P.s3v, -0.5f), P.s3v, s0, s1, fs);
      smk_lin_clamp(__fmaf_rn(ch1, (float)P.s3g, -0.5f), P.s3g, t0, t1, ft);
      if (!((P.tf_occ[t0 * P.occ_roww + (s0 >> 5)] >> (s0 & 31)) & 1u)) continue;
    }
    if (TF == 2 || (TF == 1 && P.third_axis)) {
      ch2 = SMK_TRI(c2);
      if (DT == 0) ch2 *= sc;
      if (P.nelts == 4) {
        ch3 = SMK_TRI(c3);
        if (DT == 0) ch3 *= sc;
      }
    }

    if (!smk_classify<DT, TF>(P, ch0, ch1, ch2, ch3, col)) continue;

    float4 src;
    // frames with shadows: the light-buffer colour over the sample, as the slices nearer the light left it (smk_shadow.hip)
    float shadow[3];
    const float *shp = nullptr;
    if (TF != 0 && SHD) {
      smk_shadow_term(P, m, q0, q1, q2, shadow);
      shp = shadow;
    }
    if (TF == 0) {
      src = col;
    } else if (SH == 0) {
      src = smk_shade_sample<0>(P, col, 0.f, 0.f, 0.f, 0.f, shp);
    } else {
      float n0 = smk_nrm(k000.nb, k100.nb, k010.nb, k110.nb, k001.nb, k101.nb, k011.nb, k111.nb, 0, fx, fy, fz);
      float n1 = smk_nrm(k000.nb, k100.nb, k010.nb, k110.nb, k001.nb, k101.nb, k011.nb, k111.nb, 1, fx, fy, fz);
      float n2 = smk_nrm(k000.nb, k100.nb, k010.nb, k110.nb, k001.nb, k101.nb, k011.nb, k111.nb, 2, fx, fy, fz);
      src = smk_shade_sample<SH>(P, col, n0, n1, n2, ch1, shp);
    }
    if (P.blend == SMK_BLEND_FRONT_TO_BACK) {
      // C += (1-A) src   (GL_ONE_MINUS_DST_ALPHA, GL_ONE)
      float w = 1.0f - C3;
      if (first == __int_as_float(0x7f800000)) first = __fmaf_rn((float)m, rc.dtau, rc.tau0) * P.znear;
      C0 = __fmaf_rn(w, src.x, C0);
      C1 = __fmaf_rn(w, src.y, C1);
      C2 = __fmaf_rn(w, src.z, C2);
      C3 = __fmaf_rn(w, src.w, C3);
      // exact early termination: once A == 1.0f every later weight (1-A) is exactly 0, so no
      // later sample can change C or A (nor the first-hit depth)
      if (C3 == 1.0f) m1 = m;
    } else if (btf) {
      // D = S + (1-S.a) D   (GL_ONE, GL_ONE_MINUS_SRC_ALPHA); the nearest contributing sample is the last one
      float w = 1.0f - src.w;
      first = __fmaf_rn((float)m, rc.dtau, rc.tau0) * P.znear;
      C0 = __fmaf_rn(w, C0, src.x);
      C1 = __fmaf_rn(w, C1, src.y);
      C2 = __fmaf_rn(w, C2, src.z);
      C3 = __fmaf_rn(w, C3, src.w);
    } else {
      // D = max(S, D) per component (GL_MAX ignores the blend factors)
      if (first == __int_as_float(0x7f800000)) first = __fmaf_rn((float)m, rc.dtau, rc.tau0) * P.znear;
      C0 = fmaxf(C0, src.x);
      C1 = fmaxf(C1, src.y);
      C2 = fmaxf(C2, src.z);
      C3 = fmaxf(C3, src.w);
    }
  }
  if (!live) return;
  size_t o = (size_t)j * P.W + i;
  P.out[o] = make_float4(C0, C1, C2, C3);
  if (P.depth) P.depth[o] = first;
}

// How many samples of the frame lie inside the volume (region, clip planes): the membership predicate of the kernel
// above evaluated for every plane of every ray, nothing fetched.  SURVEY 8(d) asks for this count beside the nominal
// W * H * planes (most rays cross the volume, not every plane of them lands inside).
__global__ __launch_bounds__(256) void smk_k_count_inside(const RenderParams P, unsigned long long *count) {
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
  unsigned n = 0;
  if (i < P.W && j < P.H) {
    const smk_raycoef &rc = P.rc;
    const float px = __fmaf_rn((float)i + 0.5f, rc.pxs, rc.pxl);
    const float py = __fmaf_rn((float)j + 0.5f, rc.pys, rc.pyl);
    float A[3], B[3], tauA, dtau;
    const bool ray_ok = smk_ray_AB(P, px, py, A, B, tauA, dtau);
    for (int m = 0; m < (ray_ok ? rc.nplanes : 0); ++m) {
      const float p0 = __fmaf_rn((float)m, B[0], A[0]), p1 = __fmaf_rn((float)m, B[1], A[1]), p2 = __fmaf_rn((float)m, B[2], A[2]);
      bool in = (p0 >= P.lo[0] && (p0 < P.hi[0] || (P.top[0] && p0 <= P.hi[0]))) && (p1 >= P.lo[1] && (p1 < P.hi[1] || (P.top[1] && p1 <= P.hi[1]))) &&
                (p2 >= P.lo[2] && (p2 < P.hi[2] || (P.top[2] && p2 <= P.hi[2])));
      if (in && P.cplane_on) in = __fmaf_rn(p0, P.cplane[0], __fmaf_rn(p1, P.cplane[1], __fmaf_rn(p2, P.cplane[2], P.cplane[3]))) >= 0.0f;
      if (in && P.sh.on) in = smk_tau_ok(tauA, dtau, m);
      n += in ? 1u : 0u;
    }
  }
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(count, (unsigned long long)n);
}

hipError_t smk_launch_count_inside(const RenderParams &P, unsigned long long *d_count, hipStream_t s) {
  hipLaunchKernelGGL(smk_k_count_inside, dim3((P.W + 15) / 16, (P.H + 15) / 16), dim3(256), 0, s, P, d_count);
  return hipGetLastError();
}

template <int DT, int TF, int SH, bool SHD = false>
static hipError_t launch(const RenderParams &P, hipStream_t s) {
  dim3 grid(8 * P.tiles_per_xcd), block(256);
  hipLaunchKernelGGL((smk_k_gather<DT, TF, SH, SHD>), grid, block, 0, s, P);
  return hipGetLastError();
}

hipError_t smk_launch_gather(const RenderParams &P, int dtype, int tf_mode, int shade_kind, hipStream_t s) {
  if (P.sh.on) {  // the eye pass of a frame with shadows: 2-D / 3-D table, R8k shading or none
#define CASE(D, T, S) \
  if (dtype == D && tf_mode == T && shade_kind == S) return launch<D, T, S, true>(P, s);
    CASE(0, 1, 0) CASE(0, 1, 1) CASE(0, 2, 0) CASE(0, 2, 1)
    CASE(1, 1, 0) CASE(1, 1, 1) CASE(1, 2, 0) CASE(1, 2, 1)
#undef CASE
    return hipErrorInvalidValue;
  }
#define CASE(D, T, S) \
  if (dtype == D && tf_mode == T && shade_kind == S) return launch<D, T, S>(P, s);
  CASE(0, 0, 0) CASE(0, 1, 0) CASE(0, 1, 1) CASE(0, 1, 2) CASE(0, 2, 0) CASE(0, 2, 1) CASE(0, 2, 2)
  CASE(1, 0, 0) CASE(1, 1, 0) CASE(1, 1, 1) CASE(1, 1, 2) CASE(1, 2, 0) CASE(1, 2, 1) CASE(1, 2, 2)
#undef CASE
  return hipErrorInvalidValue;
}
