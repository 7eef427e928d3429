// smk_device.h -- per-sample device functions shared by the render kernels (gfx950 only).
//
// Arithmetic contract (DESIGN.md "sample placement"): everything that decides WHERE a sample
// is and WHICH texels it touches is an explicit fp32 fma chain (compile with
// -ffp-contract=off); the same chains, written independently, live in the CPU checker.
//
// Semantics per sample (reference file:line, relative to the reference tree):
//   trilinear, clamp-to-edge ............ NV20VolRen3D.cpp:1379-1383, VolumeRenderer.cpp:410-418
//   1-D colour table (post-filter) ...... VolumeRenderer.cpp:576-587, TLUT.cpp:65-80
//   2-D TF (V,G) x third axis (H,4th) ... NV20VolRen3D.cpp:544-596, 810-838
//   dense 3-D TF ........................ TFWidgetRen.cpp:779-845
//   Phong, R8k cube map + shader ........ R8kVolRen3D.cpp:2620-2679, 2831-2977
//   Phong, NV20 combiners ............... NV20VolRen3D.cpp:634-806
//   front-to-back blend ................. R8kVolRen3D.cpp:1441-1449
#pragma once
#include "smk_internal.h"

#define SMK_INV255 (1.0f / 255.0f)
#define SMK_RANGE_EPS 0.0078125f  // voxels; 30 x the fp32 rounding of coordinates up to 4096 (half an ulp = 2.4e-4)

__device__ __forceinline__ float smk_lerp(float a, float b, float f) { return __fmaf_rn(f, b - a, a); }
// clamp as ONE v_med3_f32 (fminf/fmaxf make hipcc add a canonicalising v_max first); for the
// finite values that occur here med3(x,lo,hi) == min(max(x,lo),hi) exactly
__device__ __forceinline__ float smk_clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ float smk_sat(float x) { return smk_clampf(x, 0.0f, 1.0f); }

// GL_LINEAR + clamp-to-edge along one axis, x in texel units
__device__ __forceinline__ void smk_lin_clamp(float x, int n, int &i0, int &i1, float &f) {
  float xc = smk_clampf(x, 0.0f, (float)(n - 1));
  int i = (int)xc;
  int imax = n >= 2 ? n - 2 : 0;
  i = min(i, imax);
  i0 = i;
  i1 = min(i + 1, n - 1);
  f = xc - (float)i;
}

// GL_LINEAR + GL_REPEAT
__device__ __forceinline__ void smk_lin_repeat(float x, int n, int &i0, int &i1, float &f) {
  float fl = floorf(x);
  f = x - fl;
  int i = (int)fl % n;
  if (i < 0) i += n;
  i0 = i;
  i1 = (i + 1) % n;
}

__device__ __forceinline__ float smk_ub(uint32_t v, int k) { return (float)((v >> (8 * k)) & 0xffu); }

// one voxel corner: 4 channels (raw: u8 as 0..255 floats, f32 as is) + packed normal bits
struct SmkCorner {
  float c0, c1, c2, c3;
  uint32_t nb;
};

template <int DT>
__device__ __forceinline__ SmkCorner smk_load_corner(const RenderParams &P, size_t idx) {
  SmkCorner k;
  if (DT == 0) {
    uint2 v = ((const uint2 *)P.vox)[idx];
    k.c0 = smk_ub(v.x, 0);
    k.c1 = smk_ub(v.x, 1);
    k.c2 = smk_ub(v.x, 2);
    k.c3 = smk_ub(v.x, 3);
    k.nb = v.y;
  } else {
    float4 v = ((const float4 *)P.vox)[idx];
    k.c0 = v.x;
    k.c1 = v.y;
    k.c2 = v.z;
    if (P.n_in_w) {
      k.c3 = 0.0f;
      k.nb = __float_as_uint(v.w);
    } else {
      k.c3 = v.w;
      k.nb = P.nrm ? P.nrm[idx] : 0x808080u;
    }
  }
  return k;
}

// u8 voxels are 8 bytes: the two x-neighbours of a sample sit side by side in memory and come with ONE
// 16-byte load (a lane whose pair is not adjacent -- a clamped perturbed fetch -- loads the second one
// separately).  Measured on BASELINE config 5: no change (16.8 ms either way): that frame is bound by
// the latency of three dependent gathers per sample, not by their count.
__device__ __forceinline__ void smk_load_pair_u8(const RenderParams &P, size_t i0, size_t i1, SmkCorner &a, SmkCorner &b) {
  uint4 v;
  __builtin_memcpy(&v, reinterpret_cast<const char *>(P.vox) + i0 * 8, 16);  // 8-byte aligned: one global_load_dwordx4
  a.c0 = smk_ub(v.x, 0); a.c1 = smk_ub(v.x, 1); a.c2 = smk_ub(v.x, 2); a.c3 = smk_ub(v.x, 3);
  a.nb = v.y;
  uint2 w = make_uint2(v.z, v.w);
  if (i1 != i0 + 1) w = ((const uint2 *)P.vox)[i1];
  b.c0 = smk_ub(w.x, 0); b.c1 = smk_ub(w.x, 1); b.c2 = smk_ub(w.x, 2); b.c3 = smk_ub(w.x, 3);
  b.nb = w.y;
}

#define SMK_TRI(field)                                                                  \
  smk_lerp(smk_lerp(smk_lerp(k000.field, k100.field, fx), smk_lerp(k010.field, k110.field, fx), fy), \
           smk_lerp(smk_lerp(k001.field, k101.field, fx), smk_lerp(k011.field, k111.field, fx), fy), fz)

// bilinear RGBA8 lookup, s,t in [0,1]
__device__ __forceinline__ float4 smk_tex2d(const uint32_t *tex, int ss, int st, float s, float t) {
  int s0, s1, t0, t1;
  float fs, ft;
  smk_lin_clamp(__fmaf_rn(s, (float)ss, -0.5f), ss, s0, s1, fs);
  smk_lin_clamp(__fmaf_rn(t, (float)st, -0.5f), st, t0, t1, ft);
  uint32_t a = tex[t0 * ss + s0], b = tex[t0 * ss + s1], c = tex[t1 * ss + s0], d = tex[t1 * ss + s1];
  float4 o;
  o.x = smk_lerp(smk_lerp(smk_ub(a, 0), smk_ub(b, 0), fs), smk_lerp(smk_ub(c, 0), smk_ub(d, 0), fs), ft) * SMK_INV255;
  o.y = smk_lerp(smk_lerp(smk_ub(a, 1), smk_ub(b, 1), fs), smk_lerp(smk_ub(c, 1), smk_ub(d, 1), fs), ft) * SMK_INV255;
  o.z = smk_lerp(smk_lerp(smk_ub(a, 2), smk_ub(b, 2), fs), smk_lerp(smk_ub(c, 2), smk_ub(d, 2), fs), ft) * SMK_INV255;
  o.w = smk_lerp(smk_lerp(smk_ub(a, 3), smk_ub(b, 3), fs), smk_lerp(smk_ub(c, 3), smk_ub(d, 3), fs), ft) * SMK_INV255;
  return o;
}

__device__ __forceinline__ float smk_tex2d_alpha(const uint32_t *tex, int ss, int st, float s, float t) {
  int s0, s1, t0, t1;
  float fs, ft;
  smk_lin_clamp(__fmaf_rn(s, (float)ss, -0.5f), ss, s0, s1, fs);
  smk_lin_clamp(__fmaf_rn(t, (float)st, -0.5f), st, t0, t1, ft);
  uint32_t a = tex[t0 * ss + s0], b = tex[t0 * ss + s1], c = tex[t1 * ss + s0], d = tex[t1 * ss + s1];
  return smk_lerp(smk_lerp(smk_ub(a, 3), smk_ub(b, 3), fs), smk_lerp(smk_ub(c, 3), smk_ub(d, 3), fs), ft) * SMK_INV255;
}

__device__ __forceinline__ float4 smk_tex3d(const uint32_t *tex, int ss, int st, int sr, float s, float t, float r) {
  int s0, s1, t0, t1, r0, r1;
  float fs, ft, fr;
  smk_lin_clamp(__fmaf_rn(s, (float)ss, -0.5f), ss, s0, s1, fs);
  smk_lin_clamp(__fmaf_rn(t, (float)st, -0.5f), st, t0, t1, ft);
  smk_lin_clamp(__fmaf_rn(r, (float)sr, -0.5f), sr, r0, r1, fr);
  uint32_t q[8];
  if (ss >= 2) {  // the clamped texel pair is (s0, s0 + 1): four 8-byte gathers instead of eight 4-byte ones
    uint2 p;
    __builtin_memcpy(&p, tex + (r0 * st + t0) * ss + s0, 8); q[0] = p.x; q[1] = p.y;
    __builtin_memcpy(&p, tex + (r0 * st + t1) * ss + s0, 8); q[2] = p.x; q[3] = p.y;
    __builtin_memcpy(&p, tex + (r1 * st + t0) * ss + s0, 8); q[4] = p.x; q[5] = p.y;
    __builtin_memcpy(&p, tex + (r1 * st + t1) * ss + s0, 8); q[6] = p.x; q[7] = p.y;
  } else {
    q[0] = tex[(r0 * st + t0) * ss + s0];
    q[1] = tex[(r0 * st + t0) * ss + s1];
    q[2] = tex[(r0 * st + t1) * ss + s0];
    q[3] = tex[(r0 * st + t1) * ss + s1];
    q[4] = tex[(r1 * st + t0) * ss + s0];
    q[5] = tex[(r1 * st + t0) * ss + s1];
    q[6] = tex[(r1 * st + t1) * ss + s0];
    q[7] = tex[(r1 * st + t1) * ss + s1];
  }
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    o[e] = smk_lerp(smk_lerp(smk_lerp(smk_ub(q[0], e), smk_ub(q[1], e), fs), smk_lerp(smk_ub(q[2], e), smk_ub(q[3], e), fs), ft),
                    smk_lerp(smk_lerp(smk_ub(q[4], e), smk_ub(q[5], e), fs), smk_lerp(smk_ub(q[6], e), smk_ub(q[7], e), fs), ft), fr) *
           SMK_INV255;
  return make_float4(o[0], o[1], o[2], o[3]);
}

// wrap-around trilinear fetch of the noise volume, rgb only.  A power-of-two edge (the reference's 32^3) wraps with a mask
// and indexes with shifts: the general path's twelve integer modulos and eight index products per lookup were most of the
// perturbed frame's vector instructions (config 5: 12.4 -> 7 ms)
__device__ __forceinline__ void smk_noise(const RenderParams &P, float s, float t, float r, float o[3]) {
  int n = P.nn, s0, s1, t0, t1, r0, r1;
  float fs, ft, fr;
  const uint32_t *tex = P.noise;
  uint32_t q[8];
  if (P.nn_log2 >= 0) {
    const int mask = n - 1, L = P.nn_log2;
    float xs = __fmaf_rn(s, (float)n, -0.5f), xt = __fmaf_rn(t, (float)n, -0.5f), xr = __fmaf_rn(r, (float)n, -0.5f);
    float ls = floorf(xs), lt = floorf(xt), lr = floorf(xr);
    fs = xs - ls; ft = xt - lt; fr = xr - lr;
    s0 = (int)ls & mask; t0 = (int)lt & mask; r0 = (int)lr & mask;   // == ((i % n) + n) % n for a power of two
    s1 = (s0 + 1) & mask; t1 = (t0 + 1) & mask; r1 = (r0 + 1) & mask;
    const int b00 = ((r0 << L) | t0) << L, b01 = ((r0 << L) | t1) << L, b10 = ((r1 << L) | t0) << L, b11 = ((r1 << L) | t1) << L;
    q[0] = tex[b00 | s0]; q[1] = tex[b00 | s1]; q[2] = tex[b01 | s0]; q[3] = tex[b01 | s1];
    q[4] = tex[b10 | s0]; q[5] = tex[b10 | s1]; q[6] = tex[b11 | s0]; q[7] = tex[b11 | s1];
  } else {
    smk_lin_repeat(__fmaf_rn(s, (float)n, -0.5f), n, s0, s1, fs);
    smk_lin_repeat(__fmaf_rn(t, (float)n, -0.5f), n, t0, t1, ft);
    smk_lin_repeat(__fmaf_rn(r, (float)n, -0.5f), n, r0, r1, fr);
    q[0] = tex[(r0 * n + t0) * n + s0];
    q[1] = tex[(r0 * n + t0) * n + s1];
    q[2] = tex[(r0 * n + t1) * n + s0];
    q[3] = tex[(r0 * n + t1) * n + s1];
    q[4] = tex[(r1 * n + t0) * n + s0];
    q[5] = tex[(r1 * n + t0) * n + s1];
    q[6] = tex[(r1 * n + t1) * n + s0];
    q[7] = tex[(r1 * n + t1) * n + s1];
  }
#pragma unroll
  for (int e = 0; e < 3; ++e)
    o[e] = smk_lerp(smk_lerp(smk_lerp(smk_ub(q[0], e), smk_ub(q[1], e), fs), smk_lerp(smk_ub(q[2], e), smk_ub(q[3], e), fs), ft),
                    smk_lerp(smk_lerp(smk_ub(q[4], e), smk_ub(q[5], e), fs), smk_lerp(smk_ub(q[6], e), smk_ub(q[7], e), fs), ft), fr) *
           SMK_INV255;
}

// The same lookup for a ray-marcher that walks a ray: consecutive samples mostly fall into the SAME cell of the noise volume
// (the first octave's cell spans N / (32 s) = 80 voxels of a 512^3 volume at the GUI's scale .2, the second's 7.6), so the
// cell's eight texels stay in registers and are fetched again only when the cell changes -- the same texels, the same
// interpolation: bit-identical.  Saves the index arithmetic and the eight gathers of most lookups (power-of-two edge only).
struct SmkNoiseCell {
  int key;  // s0 | t0 << L | r0 << 2L of the cached cell, -1 = none
  uint32_t q[8];
};
__device__ __forceinline__ void smk_noise_cached(const RenderParams &P, float s, float t, float r, float o[3], SmkNoiseCell &c) {
  const int n = P.nn, mask = n - 1, L = P.nn_log2;
  const float xs = __fmaf_rn(s, (float)n, -0.5f), xt = __fmaf_rn(t, (float)n, -0.5f), xr = __fmaf_rn(r, (float)n, -0.5f);
  const float ls = floorf(xs), lt = floorf(xt), lr = floorf(xr);
  const float fs = xs - ls, ft = xt - lt, fr = xr - lr;
  const int s0 = (int)ls & mask, t0 = (int)lt & mask, r0 = (int)lr & mask;
  const int key = s0 | (t0 << L) | (r0 << (2 * L));
  if (key != c.key) {
    const uint32_t *tex = P.noise;
    const int s1 = (s0 + 1) & mask, t1 = (t0 + 1) & mask, r1 = (r0 + 1) & mask;
    const int b00 = ((r0 << L) | t0) << L, b01 = ((r0 << L) | t1) << L, b10 = ((r1 << L) | t0) << L, b11 = ((r1 << L) | t1) << L;
    c.q[0] = tex[b00 | s0]; c.q[1] = tex[b00 | s1]; c.q[2] = tex[b01 | s0]; c.q[3] = tex[b01 | s1];
    c.q[4] = tex[b10 | s0]; c.q[5] = tex[b10 | s1]; c.q[6] = tex[b11 | s0]; c.q[7] = tex[b11 | s1];
    c.key = key;
  }
  const uint32_t *q = c.q;
#pragma unroll
  for (int e = 0; e < 3; ++e)
    o[e] = smk_lerp(smk_lerp(smk_lerp(smk_ub(q[0], e), smk_ub(q[1], e), fs), smk_lerp(smk_ub(q[2], e), smk_ub(q[3], e), fs), ft),
                    smk_lerp(smk_lerp(smk_ub(q[4], e), smk_ub(q[5], e), fs), smk_lerp(smk_ub(q[6], e), smk_ub(q[7], e), fs), ft), fr) *
           SMK_INV255;
}

__device__ __forceinline__ float smk_pow30(float x) {
  float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
  return ((x16 * x8) * x4) * x2;
}

// classification of interpolated channels -> straight colour + alpha; returns false if alpha==0
template <int DT, int TF>
__device__ __forceinline__ bool smk_classify(const RenderParams &P, float ch0, float ch1, float ch2, float ch3,
                                             float4 &col) {
  if (TF == 0) {
    int idx = (int)__fmaf_rn(ch0, (float)(P.tlut_size - 1), 0.5f);
    idx = max(0, min(idx, P.tlut_size - 1));
    col = P.tlut[idx];  // already premultiplied
    return col.w != 0.0f;
  } else if (TF == 1) {
    col = smk_tex2d(P.tf_vg, P.sv, P.sg, ch0, ch1);
    if (P.third_axis) col.w *= smk_tex2d_alpha(P.tf_h, P.sv, P.sg, ch2, ch3);
  } else {
    col = smk_tex3d(P.tf3d, P.s3v, P.s3g, P.s3h, ch0, ch1, ch2);
  }
  col.w = smk_sat(col.w);
  return col.w != 0.0f;
}

// SH: 0 none, 1 R8k, 2 NV20.  n = decoded interpolated normal, g = second data channel.
// in: straight colour col (col.w = alpha); out: premultiplied src
// `shadow`: the light-buffer colour at the sample (half-angle slicing), applied as the R8k eye shader does
// -- MUL r0, r0, 1 - r5 before the colour is weighted by its opacity (R8kVolRen3D.cpp:2928-2934); null = none
template <int SH>
__device__ __forceinline__ float4 smk_shade_sample(const RenderParams &P, float4 col, float n0, float n1, float n2, float g,
                                                   const float *shadow = nullptr) {
  float a = col.w;
  float c[3] = {col.x, col.y, col.z};
  if (SH == 1) {
    float w0 = __fmaf_rn(P.R[0], n0, __fmaf_rn(P.R[1], n1, P.R[2] * n2));
    float w1 = __fmaf_rn(P.R[3], n0, __fmaf_rn(P.R[4], n1, P.R[5] * n2));
    float w2 = __fmaf_rn(P.R[6], n0, __fmaf_rn(P.R[7], n1, P.R[8] * n2));
    float l2 = __fmaf_rn(w0, w0, __fmaf_rn(w1, w1, w2 * w2));
    float il = l2 > 0.0f ? rsqrtf(l2) : 0.0f;  // the cube map is addressed by direction
    w0 *= il;
    w1 *= il;
    w2 *= il;
    float dl = fabsf(__fmaf_rn(P.L[0], w0, __fmaf_rn(P.L[1], w1, P.L[2] * w2)));
    float dh = fabsf(__fmaf_rn(P.Hv[0], w0, __fmaf_rn(P.Hv[1], w1, P.Hv[2] * w2)));
    float kd = smk_clampf(dl, 0.2f, 1.0f) * P.intens;  // == sat(max(sat(dl), .2))
    float ks = P.use_spec ? smk_sat(smk_pow30(smk_sat(dh))) * P.intens : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float shaded = __fmaf_rn(c[k], kd, ks);
      c[k] = __fmaf_rn(g, shaded - c[k], c[k]);
    }
  } else if (SH == 2) {
    float dl = fabsf(__fmaf_rn(P.L[0], n0, __fmaf_rn(P.L[1], n1, P.L[2] * n2)));
    float dh = __fmaf_rn(P.Hv[0], n0, __fmaf_rn(P.Hv[1], n1, P.Hv[2] * n2));
    float s2 = smk_sat(dh * dh), s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8;
    float spec = P.use_spec ? s16 * P.intens * a : 0.0f;
    float ia = P.intens * a, aa = 0.3f * a;
    float4 o;
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float cc = smk_sat(__fmaf_rn(c[k] * smk_sat(dl), ia, c[k] * aa));
      r[k] = smk_sat(__fmaf_rn(spec, 1.0f - cc, cc));
    }
    o.x = r[0];
    o.y = r[1];
    o.z = r[2];
    o.w = a;
    return o;
  }
  if (shadow) {
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] *= 1.0f - shadow[k];
  }
  return make_float4(smk_sat(c[0] * a), smk_sat(c[1] * a), smk_sat(c[2] * a), a);
}

__device__ __forceinline__ float smk_nrm(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f,
                                         uint32_t g, uint32_t h, int k, float fx, float fy, float fz) {
  float v = smk_lerp(smk_lerp(smk_lerp(smk_ub(a, k), smk_ub(b, k), fx), smk_lerp(smk_ub(c, k), smk_ub(d, k), fx), fy),
                     smk_lerp(smk_lerp(smk_ub(e, k), smk_ub(f, k), fx), smk_lerp(smk_ub(g, k), smk_ub(h, k), fx), fy), fz);
  return __fmaf_rn(v, 2.0f * SMK_INV255, -1.0f);
}

// ---- a pixel's ray: the voxel coordinate of plane m on axis a is fma(m, B_a, A_a).  View-aligned planes: A and B are
// affine in the pixel coordinate (smk_raycoef).  Frames with shadows (SmkShadowRays, smk_internal.h): the planes are the
// half-angle slices, the coefficients carry a division by the ray's component along the slice normal; `tauA`, `dtau` give
// the ray parameter fma(m, dtau, tauA) of plane m, which must be positive and finite for the sample to exist (smk_tau_ok).
// Returns false for a ray that runs parallel to the slices (no sample at all).
// (SHD: a compile-time choice in the ray-marchers -- as a run-time test the shadow form cost the slice-ring kernel of the
//  plain cfg 3 frame 3 VGPRs it does not have: spills, 2-8 % on every small-workgroup frame)
template <bool SHD>
__device__ __forceinline__ bool smk_ray_AB_t(const RenderParams &P, float px, float py, float A[3], float B[3], float &tauA, float &dtau) {
  if (!SHD) {
    const smk_raycoef &rc = P.rc;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      A[a] = __fmaf_rn(px, rc.Ax[a], __fmaf_rn(py, rc.Ay[a], rc.Ac[a]));
      B[a] = __fmaf_rn(px, rc.Bx[a], __fmaf_rn(py, rc.By[a], rc.Bc[a]));
    }
    tauA = 1.0f;
    dtau = 0.0f;
    return true;
  }
  const SmkShadowRays &sh = P.sh;
  const float nD = __fmaf_rn(px, sh.nDx, __fmaf_rn(py, sh.nDy, sh.nDc));
  tauA = __fdiv_rn(sh.numA, nD);
  dtau = __fdiv_rn(sh.dB, nD);
  const bool ok = fabsf(nD) > 0.0f && fabsf(tauA) < __int_as_float(0x7f800000) && fabsf(dtau) < __int_as_float(0x7f800000);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float D = __fmaf_rn(px, sh.Dx[a], __fmaf_rn(py, sh.Dy[a], sh.Dc[a]));
    A[a] = ok ? __fmaf_rn(tauA, D, sh.Ec[a]) : 0.0f;
    B[a] = ok ? dtau * D : 0.0f;
  }
  if (!ok) { tauA = -1.0f; dtau = 0.0f; }
  return ok;
}
__device__ __forceinline__ bool smk_ray_AB(const RenderParams &P, float px, float py, float A[3], float B[3], float &tauA, float &dtau) {
  return P.sh.on ? smk_ray_AB_t<true>(P, px, py, A, B, tauA, dtau) : smk_ray_AB_t<false>(P, px, py, A, B, tauA, dtau);
}
__device__ __forceinline__ bool smk_tau_ok(float tauA, float dtau, int m) {
  const float t = __fmaf_rn((float)m, dtau, tauA);
  return t > 0.0f && t < __int_as_float(0x7f800000);
}

// bilinear lookup of a light buffer; texels outside it are 0 (the rest of the pbuffer stays cleared)
__device__ __forceinline__ void smk_light_lookup(const float4 *L, int LB, float lx, float ly, float out[3]) {
  const float fx0 = floorf(lx - 0.5f), fy0 = floorf(ly - 0.5f);
  const float fx = (lx - 0.5f) - fx0, fy = (ly - 0.5f) - fy0;
  out[0] = out[1] = out[2] = 0.0f;
  if (!(fx0 >= -1.0f && fx0 < (float)LB && fy0 >= -1.0f && fy0 < (float)LB)) return;  // (also NaN)
  const int x0 = (int)fx0, y0 = (int)fy0;
  float4 t[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = x0 + (q & 1), y = y0 + (q >> 1);
    t[q] = (x >= 0 && x < LB && y >= 0 && y < LB) ? L[(size_t)y * LB + x] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  out[0] = smk_lerp(smk_lerp(t[0].x, t[1].x, fx), smk_lerp(t[2].x, t[3].x, fx), fy);
  out[1] = smk_lerp(smk_lerp(t[0].y, t[1].y, fx), smk_lerp(t[2].y, t[3].y, fx), fy);
  out[2] = smk_lerp(smk_lerp(t[0].z, t[1].z, fx), smk_lerp(t[2].z, t[3].z, fx), fy);
}
// the light-buffer colour over the sample of plane m at voxel coordinate p (R8kVolRen3D.cpp:1664-1676: light-buffer
// coordinates; the buffer is the one slices < k left, k the plane's slice in the light's order)
__device__ __forceinline__ void smk_shadow_term(const RenderParams &P, int m, float p0, float p1, float p2, float out[3]) {
  const SmkShadowRays &sh = P.sh;
  const float lw = __fmaf_rn(p0, sh.Wm[0], __fmaf_rn(p1, sh.Wm[1], __fmaf_rn(p2, sh.Wm[2], sh.Wm[3])));
  const float lxx = __fmaf_rn(p0, sh.Xm[0], __fmaf_rn(p1, sh.Xm[1], __fmaf_rn(p2, sh.Xm[2], sh.Xm[3])));
  const float lyy = __fmaf_rn(p0, sh.Ym[0], __fmaf_rn(p1, sh.Ym[1], __fmaf_rn(p2, sh.Ym[2], sh.Ym[3])));
  const int k = sh.k0 + sh.dk * m;
  smk_light_lookup(sh.hist + (size_t)(k - 1) * (size_t)sh.hstride, sh.LB, __fmaf_rn(__fdiv_rn(lxx, lw), sh.lscale, sh.lbias),
                   __fmaf_rn(__fdiv_rn(lyy, lw), sh.lscale, sh.lbias), out);
}

// XCD-aware tile mapping: blocks are dealt round-robin over the 8 XCDs (bid % 8 shares an
// XCD), so give each XCD one contiguous run of tiles -- neighbouring image tiles, which walk
// neighbouring voxels, then share an L2.  Speed only; any placement is correct.
__device__ __forceinline__ bool smk_tile_of_block(const RenderParams &P, int bid, int &tx, int &ty) {
  int xcd = bid & 7, k = bid >> 3;
  int tile = xcd * P.tiles_per_xcd + k;
  if (k >= P.tiles_per_xcd || tile >= P.ntx * P.nty) return false;
  ty = tile / P.ntx;
  tx = tile - ty * P.ntx;
  return true;
}
