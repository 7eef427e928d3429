// smk_prep.hip -- data preparation on the GPU: the step before the hot path.
//
//   smk_make_vgh_device      genVGH makeVGH<T>                 genVGH/main.cpp:56-182
//   smk_normals_vgh_device   MetaVolume::normalsVGH            MetaVolume.cpp:1274-1324
//                            = derivative3DVGH + blurV3D + scalebiasN
//                                                              VectorMath.h:874-899, 1217-1281, 1133-1148
//   smk_synth_volume_device  bench/test input generator (no reference equivalent; genvol's
//                            Perlin tables depend on libc rand(), so the GPU generator is analytic)
//
// These are HBM-streaming stencils over bytes/floats; results are integers (u8), so the float
// expressions follow the reference's evaluation order exactly (compile with -ffp-contract=off)
// and the outputs are bit-identical to the CPU restatement.  Nothing is staged in temporaries
// for VGH: gradients are recomputed from the L2-resident neighbourhood (two passes: min/max
// statistics, then quantise), which keeps the 1024^3 case at zero extra HBM footprint.
#include <string.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "smk_internal.h"

#define PCHK(ctx, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      (ctx)->err = std::string(#call) + " failed: " + hipGetErrorString(e_);             \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

// ------------------------------------------------------------------------------- helpers

// order-preserving float <-> int map so min/max can use integer atomics
__device__ __forceinline__ int f2o(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float o2f(int i) {
  int j = i >= 0 ? i : i ^ 0x7fffffff;
#ifdef __HIP_DEVICE_COMPILE__
  return __int_as_float(j);
#else
  float f;
  memcpy(&f, &j, 4);
  return f;
#endif
}

struct VghStats {  // ordered-int encoded
  int dmin, dmax, gmin, gmax, hmin, hmax;
};

template <int DT>
__device__ __forceinline__ float ld(const void *d, size_t i) {
  return DT == 0 ? (float)((const unsigned char *)d)[i] : ((const float *)d)[i];
}

// un-normalised central differences (genVGH/main.cpp:86-88); u8 differences are exact ints
template <int DT>
__device__ __forceinline__ void grad_at(const void *d, int sx, int sy, int i, int j, int k, float g[3]) {
  size_t sxy = (size_t)sx * sy, o = (size_t)i * sxy + (size_t)j * sx + k;
  g[0] = ld<DT>(d, o + 1) - ld<DT>(d, o - 1);
  g[1] = ld<DT>(d, o + sx) - ld<DT>(d, o - sx);
  g[2] = ld<DT>(d, o + sxy) - ld<DT>(d, o - sxy);
}

__device__ __forceinline__ bool is_border(int sx, int sy, int sz, int i, int j, int k) {
  return (k < 1) || (k > sx - 2) || (j < 1) || (j > sy - 2) || (i < 1) || (i > sz - 2);
}

// gradient of a neighbour: zero on the 1-voxel border (:79-84)
template <int DT>
__device__ __forceinline__ void grad_nb(const void *d, int sx, int sy, int sz, int i, int j, int k, float g[3]) {
  if (is_border(sx, sy, sz, i, j, k)) {
    g[0] = g[1] = g[2] = 0.f;
  } else {
    grad_at<DT>(d, sx, sy, i, j, k, g);
  }
}

// gradient magnitude + second derivative along the gradient for an interior voxel (:110-146)
template <int DT>
__device__ __forceinline__ void gh_at(const void *d, int sx, int sy, int sz, int i, int j, int k, int compat,
                                      float &gm, float &hs) {
  float g[3];
  grad_at<DT>(d, sx, sy, i, j, k, g);
  gm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
  float xp[3], xm[3], yp[3], ym[3], zp[3], zm[3];
  grad_nb<DT>(d, sx, sy, sz, i, j, k + 1, xp);
  grad_nb<DT>(d, sx, sy, sz, i, j, k - 1, xm);
  grad_nb<DT>(d, sx, sy, sz, i, j + 1, k, yp);
  grad_nb<DT>(d, sx, sy, sz, i, j - 1, k, ym);
  grad_nb<DT>(d, sx, sy, sz, i + 1, j, k, zp);
  grad_nb<DT>(d, sx, sy, sz, i - 1, j, k, zm);
  float h[9];
  h[0] = xp[0] - xm[0];
  h[1] = yp[0] - ym[0];
  h[2] = zp[0] - zm[0];
  h[3] = xp[1] - xm[1];
  h[4] = yp[1] - ym[1];
  h[5] = zp[1] - zm[1];
  h[6] = xp[2] - xm[2];
  h[7] = yp[2] - ym[2];
  h[8] = zp[2] - zm[2];
  float tg[3] = {g[0] / gm, g[1] / gm, g[2] / gm};
  float tv0 = tg[0] * h[0] + tg[1] * h[1] + tg[2] * h[2];
  // the reference drops h[4] here (genVGH/main.cpp:135-137); compat keeps the typo
  float tv1 = compat ? tg[0] * h[3] + tg[1] + tg[2] * h[5] : tg[0] * h[3] + tg[1] * h[4] + tg[2] * h[5];
  float tv2 = tg[0] * h[6] + tg[1] * h[7] + tg[2] * h[8];
  hs = tg[0] * tv0 + tg[1] * tv1 + tg[2] * tv2;
}

__device__ __forceinline__ int wave_min(int v) {
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

#define ORD_POS_INF 0x7f800000
// f2o(-inf) = 0xff800000 ^ 0x7fffffff = 0x807fffff

template <int DT>
__global__ __launch_bounds__(256) void smk_k_vgh_stats(const void *d, int sx, int sy, int sz, int compat,
                                                       VghStats *st) {
  size_t n = (size_t)sx * sy * sz;
  int dmin = ORD_POS_INF, dmax = (int)0x807fffff, gmin = ORD_POS_INF, gmax = (int)0x807fffff;
  int hmin = ORD_POS_INF, hmax = (int)0x807fffff;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    int k = (int)(t % sx), j = (int)((t / sx) % sy), i = (int)(t / ((size_t)sx * sy));
    if (is_border(sx, sy, sz, i, j, k)) continue;
    float gm, hs;
    gh_at<DT>(d, sx, sy, sz, i, j, k, compat, gm, hs);
    float dv = ld<DT>(d, t);
    dmin = min(dmin, f2o(dv));
    dmax = max(dmax, f2o(dv));
    gmin = min(gmin, f2o(gm));
    gmax = max(gmax, f2o(gm));
    if (hs == hs) {  // NaN (zero gradient) never wins the reference's MAX/MIN macros
      hmin = min(hmin, f2o(hs));
      hmax = max(hmax, f2o(hs));
    }
  }
  dmin = wave_min(dmin);
  dmax = wave_max(dmax);
  gmin = wave_min(gmin);
  gmax = wave_max(gmax);
  hmin = wave_min(hmin);
  hmax = wave_max(hmax);
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&st->dmin, dmin);
    atomicMax(&st->dmax, dmax);
    atomicMin(&st->gmin, gmin);
    atomicMax(&st->gmax, gmax);
    atomicMin(&st->hmin, hmin);
    atomicMax(&st->hmax, hmax);
  }
}

__device__ __forceinline__ double affine_d(double i, double x, double I, double o, double O) {
  return ((O) - (o)) * ((x) - (i)) / ((I) - (i)) + (o);
}

// (unsigned char) cast as x86 does it: truncate through int32, low byte; NaN/out of range -> 0
__device__ __forceinline__ unsigned char uc_cast(double x) {
  if (!(x > -2147483649.0 && x < 2147483648.0)) return 0;
  return (unsigned char)((int)x & 0xff);
}

template <int DT>
__global__ __launch_bounds__(256) void smk_k_vgh_quant(const void *d, int sx, int sy, int sz, int compat,
                                                       float dmin, float dmax, float gmmin, float gmmax, float hmin,
                                                       float hmax, unsigned char *o8, float *of) {
  size_t n = (size_t)sx * sy * sz;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    int k = (int)(t % sx), j = (int)((t / sx) % sy), i = (int)(t / ((size_t)sx * sy));
    unsigned char q0 = 0, q1 = 0, q2 = 0;
    float f0 = 0.f, f1 = 0.f, f2 = 0.f;
    if (!is_border(sx, sy, sz, i, j, k)) {
      float gm, hs;
      gh_at<DT>(d, sx, sy, sz, i, j, k, compat, gm, hs);
      // genVGH/main.cpp:164-175
      double vq = affine_d(dmin, ld<DT>(d, t), dmax, 0, 255);
      double gq = affine_d(gmmin, gm, gmmax, 0, 255);
      double hq;
      if (hs < 0) {
        float th = (float)affine_d(hmin, hs, 0, 0, 1);
        hq = affine_d(0, th, 1, 0, 255 / 3);
      } else {
        float th = (float)affine_d(0, hs, hmax, 0, 1);
        hq = affine_d(0, th, 1, 255 / 3, 255 / 3 * 2);
      }
      q0 = uc_cast(vq);
      q1 = uc_cast(gq);
      q2 = uc_cast(hq);
      f0 = (float)(vq / 255.0);
      f1 = (float)(gq / 255.0);
      f2 = (hq == hq) ? (float)(hq / 255.0) : 0.0f;
    }
    if (o8) {
      o8[t * 3 + 0] = q0;
      o8[t * 3 + 1] = q1;
      o8[t * 3 + 2] = q2;
    }
    if (of) {
      of[t * 3 + 0] = f0;
      of[t * 3 + 1] = f1;
      of[t * 3 + 2] = f2;
    }
  }
}

extern "C" int smk_make_vgh_device(smk_ctx *c, const void *d, smk_dtype dt, int sx, int sy, int sz, int compat,
                                   void *o8, void *of) {
  if (!c) return 1;
  PCHK(c, hipSetDevice(c->device));
  if (!d || (!o8 && !of) || sx < 3 || sy < 3 || sz < 3) {
    c->err = "smk_make_vgh_device: bad arguments (need dims >= 3 and one output)";
    return 1;
  }
  VghStats *st = nullptr;
  PCHK(c, hipMalloc((void **)&st, sizeof(VghStats)));
  VghStats init = {ORD_POS_INF, (int)0x807fffff, ORD_POS_INF, (int)0x807fffff, ORD_POS_INF, (int)0x807fffff};
  PCHK(c, hipMemcpy(st, &init, sizeof init, hipMemcpyHostToDevice));
  size_t n = (size_t)sx * sy * sz;
  unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
  if (dt == SMK_U8) hipLaunchKernelGGL(smk_k_vgh_stats<0>, dim3(blocks), dim3(256), 0, c->stream, d, sx, sy, sz, compat, st);
  else hipLaunchKernelGGL(smk_k_vgh_stats<1>, dim3(blocks), dim3(256), 0, c->stream, d, sx, sy, sz, compat, st);
  PCHK(c, hipGetLastError());
  PCHK(c, hipStreamSynchronize(c->stream));
  VghStats h;
  PCHK(c, hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
  (void)hipFree(st);
  // the reference seeds its running min/max with +-1e8 (genVGH/main.cpp:69-72, 104-105)
  float dmin = std::min(o2f(h.dmin), 100000000.f), dmax = std::max(o2f(h.dmax), -100000000.f);
  float gmin = std::min(o2f(h.gmin), 100000000.f), gmax = std::max(o2f(h.gmax), -100000000.f);
  float hmin = std::min(o2f(h.hmin), 100000000.f), hmax = std::max(o2f(h.hmax), -100000000.f);
  if (dt == SMK_U8)
    hipLaunchKernelGGL(smk_k_vgh_quant<0>, dim3(blocks), dim3(256), 0, c->stream, d, sx, sy, sz, compat, dmin, dmax,
                       gmin, gmax, hmin, hmax, (unsigned char *)o8, (float *)of);
  else
    hipLaunchKernelGGL(smk_k_vgh_quant<1>, dim3(blocks), dim3(256), 0, c->stream, d, sx, sy, sz, compat, dmin, dmax,
                       gmin, gmax, hmin, hmax, (unsigned char *)o8, (float *)of);
  PCHK(c, hipGetLastError());
  PCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ------------------------------------------------------------------------------- normals

// derivative3DVGH: int central differences of channel 0, 0 on the border (VectorMath.h:874-899)
__device__ __forceinline__ void nrm_grad(const unsigned char *d, int ne, int sx, int sy, int sz, int i, int j, int k,
                                         float g[3]) {
  if (is_border(sx, sy, sz, i, j, k)) {
    g[0] = g[1] = g[2] = 0.f;
    return;
  }
  size_t sxy = (size_t)sx * sy, o = (size_t)i * sxy + (size_t)j * sx + k;
  g[0] = (float)((int)d[(o + 1) * ne] - (int)d[(o - 1) * ne]);
  g[1] = (float)((int)d[(o + sx) * ne] - (int)d[(o - sx) * ne]);
  g[2] = (float)((int)d[(o + sxy) * ne] - (int)d[(o - sxy) * ne]);
}

// scalebiasN (VectorMath.h:1133-1148) with the +1 -> 255 clamp
__device__ __forceinline__ void nrm_store(float g[3], unsigned char *out) {
  float len = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
  if (len > 0) {
    g[0] /= len;
    g[1] /= len;
    g[2] /= len;
  }
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    float s = g[e] * 128 + 128;
    out[e] = s >= 255.0f ? 255 : (unsigned char)((int)s & 0xff);
  }
}

__global__ __launch_bounds__(256) void smk_k_normals(const unsigned char *d, int ne, int sx, int sy, int sz, int blur,
                                                     unsigned char *out) {
  size_t n = (size_t)sx * sy * sz;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    int k = (int)(t % sx), j = (int)((t / sx) % sy), i = (int)(t / ((size_t)sx * sy));
    float g[3];
    if (!blur) {
      nrm_grad(d, ne, sx, sy, sz, i, j, k, g);
    } else {
      // blurV3D is a scatter from every interior voxel to its 27 neighbours in raster order
      // of the SOURCE (VectorMath.h:1232-1266).  Gathering the same terms in the same order
      // (source raster order == ascending di,dj,dk) reproduces the float sums bit for bit.
      const float w[4] = {1.0f, .3f, .2f, .1f};  // MetaVolume.cpp:1316
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      for (int di = -1; di <= 1; ++di)
        for (int dj = -1; dj <= 1; ++dj)
          for (int dk = -1; dk <= 1; ++dk) {
            int si = i + di, sj = j + dj, sk = k + dk;
            // only interior voxels scatter (loops run 1..n-2)
            if (si < 1 || si > sz - 2 || sj < 1 || sj > sy - 2 || sk < 1 || sk > sx - 2) continue;
            float s[3];
            nrm_grad(d, ne, sx, sy, sz, si, sj, sk, s);
            float ww = w[(di != 0) + (dj != 0) + (dk != 0)];
            a0 = a0 + ww * s[0];
            a1 = a1 + ww * s[1];
            a2 = a2 + ww * s[2];
          }
      const float div = 1.0f + 6 * .3f + 12 * .2f + 8 * .1f;
      g[0] = a0 / div;
      g[1] = a1 / div;
      g[2] = a2 / div;
    }
    nrm_store(g, out + t * 3);
  }
}

extern "C" int smk_normals_vgh_device(smk_ctx *c, const void *d, int ne, int sx, int sy, int sz, int blur,
                                      void *out) {
  if (!c) return 1;
  PCHK(c, hipSetDevice(c->device));
  if (!d || !out || ne < 1 || sx < 1 || sy < 1 || sz < 1) {
    c->err = "smk_normals_vgh_device: bad arguments";
    return 1;
  }
  size_t n = (size_t)sx * sy * sz;
  unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(smk_k_normals, dim3(blocks), dim3(256), 0, c->stream, (const unsigned char *)d, ne, sx, sy, sz,
                     blur, (unsigned char *)out);
  PCHK(c, hipGetLastError());
  PCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ------------------------------------------------------------------------------- multi-field merge
// MetaVolume::mergeMV with addG (MetaVolume.cpp:1109-1268): nf scalar fields interleaved
// [z][y][x][nf] -> [z][y][x][nf+1] with G = |sum over fields of un-normalised central differences|
// (AGradArb, VectorMath.h:945-1004; 0 on the 1-voxel border) scaled by the volume's maximum to
// 0..255 (GMag, :1010-1030), and the normal bytes of that summed gradient (scalebiasN).
__device__ __forceinline__ void merge_grad(const unsigned char *in, int nf, int sx, int sy, int sz, int i, int j, int k, float g[3]) {
  g[0] = g[1] = g[2] = 0.f;
  if (is_border(sx, sy, sz, i, j, k)) return;
  const size_t sxy = (size_t)sx * sy, o = (size_t)i * sxy + (size_t)j * sx + k;
  for (int e = 0; e < nf; ++e) {  // byte differences and their sums are exact in float
    g[0] += (float)in[(o + 1) * nf + e] - (float)in[(o - 1) * nf + e];
    g[1] += (float)in[(o + sx) * nf + e] - (float)in[(o - sx) * nf + e];
    g[2] += (float)in[(o + sxy) * nf + e] - (float)in[(o - sxy) * nf + e];
  }
}

__global__ __launch_bounds__(256) void smk_k_merge_max(const unsigned char *in, int nf, int sx, int sy, int sz, int *omax) {
  const size_t n = (size_t)sx * sy * sz;
  int m = 0;  // f2o(0.0f): magnitudes are >= 0
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    int k = (int)(t % sx), j = (int)((t / sx) % sy), i = (int)(t / ((size_t)sx * sy));
    float g[3];
    merge_grad(in, nf, sx, sy, sz, i, j, k, g);
    m = max(m, f2o(sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2])));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(omax, m);
}

__global__ __launch_bounds__(256) void smk_k_merge_write(const unsigned char *in, int nf, int sx, int sy, int sz, const int *omax,
                                                         unsigned char *out, unsigned char *nrm) {
  const size_t n = (size_t)sx * sy * sz;
  const float maxm = o2f(*omax);
  const int ne = nf + 1;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    int k = (int)(t % sx), j = (int)((t / sx) % sy), i = (int)(t / ((size_t)sx * sy));
    float g[3];
    merge_grad(in, nf, sx, sy, sz, i, j, k, g);
    for (int e = 0; e < nf; ++e) out[t * ne + e] = in[t * nf + e];
    const float mag = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    const double q = (double)(mag / maxm) * 255.0;  // (uchar)(mag/max*255.0): float quotient, double product
    out[t * ne + nf] = (q > -2147483649.0 && q < 2147483648.0) ? (unsigned char)((int)q & 0xff) : 0;
    if (nrm) nrm_store(g, nrm + t * 3);
  }
}

extern "C" int smk_merge_fields_device(smk_ctx *c, const void *d_fields, int nf, int sx, int sy, int sz, void *d_out,
                                       void *d_normals) {
  if (!c) return 1;
  PCHK(c, hipSetDevice(c->device));
  if (!d_fields || !d_out || nf < 1 || nf > 3 || sx < 3 || sy < 3 || sz < 3) {
    c->err = "smk_merge_fields_device: bad arguments (1..3 fields, dims >= 3)";
    return 1;
  }
  int *d_max = nullptr;
  PCHK(c, hipMalloc((void **)&d_max, sizeof(int)));
  const size_t n = (size_t)sx * sy * sz;
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
  int rc = hipMemsetAsync(d_max, 0, sizeof(int), c->stream) != hipSuccess;
  if (!rc) {
    hipLaunchKernelGGL(smk_k_merge_max, dim3(blocks), dim3(256), 0, c->stream, (const unsigned char *)d_fields, nf, sx, sy, sz, d_max);
    hipLaunchKernelGGL(smk_k_merge_write, dim3(blocks), dim3(256), 0, c->stream, (const unsigned char *)d_fields, nf, sx, sy, sz,
                       d_max, (unsigned char *)d_out, (unsigned char *)d_normals);
    rc = hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess;
  }
  (void)hipFree(d_max);
  if (rc) {
    c->err = "smk_merge_fields_device: HIP error";
    return 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------- 2-D histogram
// MetaVolume::hist2D (MetaVolume.cpp:1650-1688): counts of (value, gradient) byte pairs.  HBM-bound
// byte work: every workgroup keeps ALL 65536 bins in LDS as 16-bit counters (128 KB) and flushes
// them to the global 32-bit bins after at most 61440 voxels, so no counter can overflow; a wave
// whose 64 voxels fall into one bin (air, the common case) adds once instead of 64 times.
#define SMK_HIST_CHUNK 61440  // voxels per flush (< 65536), a multiple of the block size

__global__ __launch_bounds__(1024) void smk_k_hist2d(const unsigned char *vol, int nelts, size_t nvox, unsigned *bins) {
  extern __shared__ unsigned h16[];  // 32768 words = 65536 halves
  const size_t nchunks = (nvox + SMK_HIST_CHUNK - 1) / SMK_HIST_CHUNK;
  for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    for (int w = threadIdx.x; w < 32768; w += 1024) h16[w] = 0;
    __syncthreads();
    const size_t base = chunk * SMK_HIST_CHUNK;
    for (int k = 0; k < SMK_HIST_CHUNK / 1024; ++k) {
      const size_t i = base + (size_t)k * 1024 + threadIdx.x;
      const bool ok = i < nvox;
      unsigned key = 0;
      if (ok) {
        const unsigned char *d = vol + i * nelts;
        key = (unsigned)d[0] | ((unsigned)d[1] << 8);
      }
      const unsigned long long live = __ballot(ok);
      if (!live) continue;
      const unsigned first = __builtin_amdgcn_readfirstlane(key);  // (of the first live lane)
      if (__ballot(ok && key == first) == live) {
        if (ok && (int)(threadIdx.x & 63) == __builtin_ctzll(live))
          atomicAdd(&h16[first >> 1], (unsigned)__builtin_popcountll(live) << ((first & 1) * 16));
      } else if (ok) {
        atomicAdd(&h16[key >> 1], 1u << ((key & 1) * 16));
      }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < 32768; w += 1024) {
      const unsigned x = h16[w];
      if (x & 0xffffu) atomicAdd(&bins[2 * w], x & 0xffffu);
      if (x >> 16) atomicAdd(&bins[2 * w + 1], x >> 16);
    }
    __syncthreads();
  }
}

static int hist2d_count(smk_ctx *c, const unsigned char *d_vol, int nelts, size_t nvox, unsigned *d_bins) {
  static bool attr[64] = {};  // per device
  const int dev = c->device;
  if (dev < 0 || dev >= 64 || !attr[dev]) {
    PCHK(c, hipFuncSetAttribute((const void *)smk_k_hist2d, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    if (dev >= 0 && dev < 64) attr[dev] = true;
  }
  const size_t nchunks = (nvox + SMK_HIST_CHUNK - 1) / SMK_HIST_CHUNK;
  const unsigned blocks = (unsigned)std::min<size_t>(nchunks, 256 * 4);
  hipLaunchKernelGGL(smk_k_hist2d, dim3(blocks), dim3(1024), 128 * 1024, c->stream, d_vol, nelts, nvox, d_bins);
  PCHK(c, hipGetLastError());
  return 0;
}

// counts -> the reference's log-scaled bytes.  Its bins are floats incremented by 1: exact up to
// 2^24, where they stop growing.
static void hist2d_finish(const unsigned *counts, unsigned char *hist) {
  std::vector<float> lg(65536);
  float mx = 0;
  for (int i = 0; i < 65536; ++i) {
    const float ih = (float)std::min(counts[i], 16777216u);
    lg[i] = (float)log((double)ih);
    mx = lg[i] > mx ? lg[i] : mx;
  }
  for (int i = 0; i < 65536; ++i) hist[i] = (!(mx > 0) || !std::isfinite(lg[i])) ? 0 : (unsigned char)(lg[i] / mx * 255);
}

extern "C" int smk_hist2d_device(smk_ctx *c, const void *d_vol, int nelts, int sx, int sy, int sz, unsigned char *hist) {
  if (!c) return 1;
  PCHK(c, hipSetDevice(c->device));
  if (!d_vol || !hist || sx < 1 || sy < 1 || sz < 1) {
    c->err = "smk_hist2d_device: bad arguments";
    return 1;
  }
  if (nelts < 2) {
    c->err = "smk_hist2d_device: sorry this type of histogram is not implemented (needs value and gradient bytes)";
    return 1;
  }
  unsigned *d_bins = nullptr;
  PCHK(c, hipMalloc((void **)&d_bins, 65536 * sizeof(unsigned)));
  int rc = 0;
  std::vector<unsigned> counts(65536);
  if (hipMemsetAsync(d_bins, 0, 65536 * sizeof(unsigned), c->stream) != hipSuccess ||
      hist2d_count(c, (const unsigned char *)d_vol, nelts, (size_t)sx * sy * sz, d_bins) ||
      hipMemcpyAsync(counts.data(), d_bins, 65536 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) {
    if (c->err.empty()) c->err = "smk_hist2d_device: HIP error";
    rc = 1;
  }
  (void)hipFree(d_bins);
  if (rc) return rc;
  hist2d_finish(counts.data(), hist);
  return 0;
}

// host bricks (the MetaVolume the application holds): staged through the device brick by brick
extern "C" int smk_hist2d(smk_ctx *c, const smk_volume_desc *b, int nb, int nelts, unsigned char *hist) {
  if (!c) return 1;
  PCHK(c, hipSetDevice(c->device));
  if (!b || nb < 1 || !hist) {
    c->err = "smk_hist2d: bad arguments";
    return 1;
  }
  if (nelts < 2) {
    c->err = "smk_hist2d: sorry this type of histogram is not implemented (needs value and gradient bytes)";
    return 1;
  }
  unsigned *d_bins = nullptr;
  unsigned char *d_stage = nullptr;
  size_t cap = 0;
  PCHK(c, hipMalloc((void **)&d_bins, 65536 * sizeof(unsigned)));
  int rc = hipMemsetAsync(d_bins, 0, 65536 * sizeof(unsigned), c->stream) != hipSuccess;
  for (int i = 0; i < nb && !rc; ++i) {
    const size_t nvox = (size_t)b[i].xiSize * b[i].yiSize * b[i].ziSize, bytes = nvox * nelts;
    if (!b[i].data || !nvox) { c->err = "smk_hist2d: brick without data"; rc = 1; break; }
    if (bytes > cap) {
      if (d_stage) (void)hipFree(d_stage);
      d_stage = nullptr;
      if (hipMalloc((void **)&d_stage, bytes) != hipSuccess) { c->err = "smk_hist2d: out of device memory"; rc = 1; break; }
      cap = bytes;
    }
    rc = hipMemcpyAsync(d_stage, b[i].data, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
         hist2d_count(c, d_stage, nelts, nvox, d_bins) || hipStreamSynchronize(c->stream) != hipSuccess;
  }
  std::vector<unsigned> counts(65536);
  if (!rc) rc = hipMemcpy(counts.data(), d_bins, 65536 * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess;
  if (d_stage) (void)hipFree(d_stage);
  (void)hipFree(d_bins);
  if (rc) {
    if (c->err.empty()) c->err = "smk_hist2d: HIP error";
    return 1;
  }
  hist2d_finish(counts.data(), hist);
  return 0;
}

// ------------------------------------------------------------------------------- synthetic volume

__device__ __forceinline__ uint32_t hash3(uint32_t x, uint32_t y, uint32_t z, uint32_t s) {
  uint32_t h = x * 0x8da6b343u ^ y * 0xd8163841u ^ z * 0xcb1ab31fu ^ s * 0x9e3779b9u;
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}

// lattice value noise in [-1,1], smoothstep-interpolated
__device__ __forceinline__ float vnoise(float x, float y, float z, uint32_t seed) {
  float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  float tx = x - fx, ty = y - fy, tz = z - fz;
  tx = tx * tx * (3.f - 2.f * tx);
  ty = ty * ty * (3.f - 2.f * ty);
  tz = tz * tz * (3.f - 2.f * tz);
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    v[q] = (float)(hash3(ix + (q & 1), iy + ((q >> 1) & 1), iz + (q >> 2), seed) >> 8) * (2.0f / 16777216.0f) - 1.0f;
  float a = v[0] + tx * (v[1] - v[0]), b = v[2] + tx * (v[3] - v[2]);
  float cc = v[4] + tx * (v[5] - v[4]), dd = v[6] + tx * (v[7] - v[6]);
  float e = a + ty * (b - a), f = cc + ty * (dd - cc);
  return e + tz * (f - e);
}

// kind 0: smooth concentric shells whose radius is perturbed by 3 octaves of value noise --
// the shape of genvol's "-spheres 4 -p .. -pabs" volumes (genvol/main.cpp:212-256) without
// the staircase, so gradients and second derivatives are non-degenerate everywhere
__global__ __launch_bounds__(256) void smk_k_synth(int kind, uint32_t seed, int sx, int sy, int sz,
                                                   unsigned char *out) {
  size_t n = (size_t)sx * sy * sz;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    int k = (int)(t % sx), j = (int)((t / sx) % sy), i = (int)(t / ((size_t)sx * sy));
    float x = (k + 0.5f) / sx, y = (j + 0.5f) / sy, z = (i + 0.5f) / sz;
    float dx = x - .5f, dy = y - .5f, dz = z - .5f;
    float r = sqrtf(dx * dx + dy * dy + dz * dz);
    float nz = vnoise(x * 4.f, y * 4.f, z * 4.f, seed) + 0.5f * vnoise(x * 8.f, y * 8.f, z * 8.f, seed + 1) +
               0.25f * vnoise(x * 16.f, y * 16.f, z * 16.f, seed + 2);
    float rr = fminf(fmaxf(r + 0.06f * nz, 0.f), .5f);
    float s = 1.f - 2.f * rr;
    float v = s * (0.62f + 0.38f * __cosf(15.f * s));
    out[t] = (unsigned char)(fminf(fmaxf(v, 0.f), 1.f) * 255.f + 0.5f);
  }
}

// ------------------------------------------------------------------------------ genvol on the GPU
// kind 1: the reference's test-volume generator, `genvol -spheres 4 -p 10 -pscale .7 -pwrap 3 3 3
// -pabs -blur -bw 1 1 1 .7` (genvol/scripts/make64.bat:1; SURVEY 8d's input for configs 2-5):
//   makeSpheres (genvol/main.cpp:212-256): r = min(|p - 1/2|, .48) + perl(p), clamped to [0, .5];
//       val = (int)(2 r n); voxel = (n - val) * (255 / n)
//   perl (main.cpp:153-165) with -pabs: PerlinNoise3Dabs(p * wrap, alpha 2, beta 2, 10 harmonics) * pscale
//       (genvol/perlin.c:76-124 noise3, :246-263 the absolute-value harmonic sum); double precision
//   blur (main.cpp:334-430): 27-tap, weights bw0, bw1/6, bw2/12, bw3/8, divisor bw0+bw1+bw2+bw3
// Bytes identical to the CPU checker's restatement (itself pinned against the reference's perlin.c):
// every operation keeps the C source's type (float where it is float, double where it is double),
// -ffp-contract=off.  The Perlin tables come from libc's rand() in the reference (srand(seed), perlin.c
// :145-176, drawn TWICE: genvol's main calls init(), and noise3's `start` flag runs it again on the
// first call, perlin.c:84-87); glibc's TYPE_3 generator is restated below so the volume does not
// depend on the host's libc.
namespace {
struct GenvolTables {
  int p[514];
  double g3[514][3];
};

struct GlibcRand {  // glibc random_r TYPE_3: r[k] = r[k-31] + r[k-3] (mod 2^32), output >> 1, after 310 discarded draws
  uint32_t ring[34];
  int pos = 0;
  explicit GlibcRand(unsigned seed) {
    int32_t r[344];
    if (seed == 0) seed = 1;
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
      long long w = (16807LL * r[i - 1]) % 2147483647LL;
      if (w < 0) w += 2147483647LL;
      r[i] = (int32_t)w;
    }
    for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
    for (int i = 34; i < 344; ++i) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
    for (int i = 0; i < 34; ++i) ring[i] = (uint32_t)r[310 + i];
  }
  int next() {
    uint32_t v = ring[(pos + 34 - 31) % 34] + ring[(pos + 34 - 3) % 34];
    ring[pos] = v;
    pos = (pos + 1) % 34;
    return (int)(v >> 1);
  }
};

void genvol_perlin_init(GlibcRand &rnd, GenvolTables &T) {  // perlin.c:145-176 (the 1-D and 2-D tables only consume draws)
  const int B = 0x100;
  int i, j, k;
  for (i = 0; i < B; i++) {
    T.p[i] = i;
    (void)rnd.next();                          // g1[i]
    for (j = 0; j < 2; j++) (void)rnd.next();  // g2[i]
    for (j = 0; j < 3; j++) T.g3[i][j] = (double)((rnd.next() % (B + B)) - B) / B;
    const double s = sqrt(T.g3[i][0] * T.g3[i][0] + T.g3[i][1] * T.g3[i][1] + T.g3[i][2] * T.g3[i][2]);
    for (j = 0; j < 3; j++) T.g3[i][j] /= s;
  }
  while (--i) {
    k = T.p[i];
    T.p[i] = T.p[j = rnd.next() % B];
    T.p[j] = k;
  }
  for (i = 0; i < B + 2; i++) {
    T.p[B + i] = T.p[i];
    for (j = 0; j < 3; j++) T.g3[B + i][j] = T.g3[i][j];
  }
}
}  // namespace

__device__ __forceinline__ double gv_noise3(const int *pp, const double (*pg3)[3], const double vec[3]) {
  int b0[3], b1[3];
  double r0[3], r1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double t = vec[a] + 4096.0;  // N = 0x1000
    b0[a] = ((int)t) & 0xff;
    b1[a] = (b0[a] + 1) & 0xff;
    r0[a] = t - (int)t;
    r1[a] = r0[a] - 1.;
  }
  const int i = pp[b0[0]], j = pp[b1[0]];
  const int b00 = pp[i + b0[1]], b10 = pp[j + b0[1]], b01 = pp[i + b1[1]], b11 = pp[j + b1[1]];
  const double t = r0[0] * r0[0] * (3. - 2. * r0[0]);
  const double sy = r0[1] * r0[1] * (3. - 2. * r0[1]);
  const double sz = r0[2] * r0[2] * (3. - 2. * r0[2]);
  const double *q;
  double u, v, a, b, c, d;
#define AT3(rx, ry, rz) (rx * q[0] + ry * q[1] + rz * q[2])
#define LERP(t, a, b) (a + t * (b - a))
  q = pg3[b00 + b0[2]]; u = AT3(r0[0], r0[1], r0[2]);
  q = pg3[b10 + b0[2]]; v = AT3(r1[0], r0[1], r0[2]);
  a = LERP(t, u, v);
  q = pg3[b01 + b0[2]]; u = AT3(r0[0], r1[1], r0[2]);
  q = pg3[b11 + b0[2]]; v = AT3(r1[0], r1[1], r0[2]);
  b = LERP(t, u, v);
  c = LERP(sy, a, b);
  q = pg3[b00 + b1[2]]; u = AT3(r0[0], r0[1], r1[2]);
  q = pg3[b10 + b1[2]]; v = AT3(r1[0], r0[1], r1[2]);
  a = LERP(t, u, v);
  q = pg3[b01 + b1[2]]; u = AT3(r0[0], r1[1], r1[2]);
  q = pg3[b11 + b1[2]]; v = AT3(r1[0], r1[1], r1[2]);
  b = LERP(t, u, v);
  d = LERP(sy, a, b);
  return LERP(sz, c, d);
#undef AT3
#undef LERP
}

// makeSpheres with the -pabs perturbation; tables staged in LDS (14.4 KB)
__global__ __launch_bounds__(256) void smk_k_genvol_spheres(const GenvolTables *tab, int sx, int sy, int sz, int nspheres, int pharm,
                                                            double pscale, float w0, float w1, float w2, float palpha, float pbeta,
                                                            unsigned char *out) {
  __shared__ int pp[514];
  __shared__ double pg3[514][3];
  for (int e = threadIdx.x; e < 514; e += 256) {
    pp[e] = tab->p[e];
    pg3[e][0] = tab->g3[e][0];
    pg3[e][1] = tab->g3[e][1];
    pg3[e][2] = tab->g3[e][2];
  }
  __syncthreads();
  const float dd = 255 / (float)nspheres;
  const float dx = 1 / (float)sx, dy = 1 / (float)sy, dz = 1 / (float)sz;
  const size_t n = (size_t)sx * sy * sz;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(t % sx), j = (int)((t / sx) % sy), i = (int)(t / ((size_t)sx * sy));
    const float p[3] = {k * dx, j * dy, i * dz};
    const float v[3] = {p[0] - .5f, p[1] - .5f, p[2] - .5f};
    const float nv = (float)sqrt((double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));  // normV3
    float r = (float)(((double)nv < .48) ? (double)nv : .48);
    {
      // PerlinNoise3Dabs(x * wrap, alpha, beta, n) * pscale (perlin.c:246-263)
      double q[3] = {(double)p[0] * w0, (double)p[1] * w1, (double)p[2] * w2};
      double sum = 0, scale = 1;
      for (int h = 0; h < pharm; h++) {
        double val = gv_noise3(pp, pg3, q);
        val = val < 0 ? -val : val;
        sum += val / scale;
        scale *= palpha;
        q[0] *= pbeta;
        q[1] *= pbeta;
        q[2] *= pbeta;
      }
      r += (float)(sum * pscale);
      const double rd = r;
      r = (float)(rd > 0 ? (rd < .5 ? rd : .5) : 0);  // CLAMP_ARB(0, r, .5)
    }
    const int val = (int)(r * 2 * nspheres);
    out[t] = (unsigned char)(int)((double)((nspheres - val) * dd));
  }
}

// blur as a gather: the reference scatters every interior voxel's weighted value into its 27
// neighbours in voxel order (i, j, k ascending); a target therefore receives its contributions in
// ascending order of the SOURCE index, i.e. descending offset -- kept here, so the float sums round alike
__global__ __launch_bounds__(256) void smk_k_genvol_blur(const unsigned char *in, int sx, int sy, int sz, float bw0, float bw1, float bw2,
                                                         float bw3, unsigned char *out) {
  const size_t n = (size_t)sx * sy * sz;
  const float div = bw0 + bw1 + bw2 + bw3;
  // a tap's contribution (float)(d * w / count), d = byte / 255.0 in double, depends on the byte and on the tap's class
  // (centre, face, edge, corner) alone: 4 x 256 floats per workgroup, computed once with exactly that arithmetic
  // (54 double divisions per voxel otherwise: 31 -> 20 ms on 1024^3; the rest is 27 byte gathers per voxel)
  __shared__ float lut[4][256];
  {
    const double d = (double)threadIdx.x / 255.0;
    lut[0][threadIdx.x] = (float)(d * bw0);
    lut[1][threadIdx.x] = (float)(d * bw1 / 6.0);
    lut[2][threadIdx.x] = (float)(d * bw2 / 12.0);
    lut[3][threadIdx.x] = (float)(d * bw3 / 8.0);
  }
  __syncthreads();
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(t % sx), j = (int)((t / sx) % sy), i = (int)(t / ((size_t)sx * sy));
    float acc = 0.f;
    for (int di = 1; di >= -1; --di)
      for (int dj = 1; dj >= -1; --dj)
        for (int dk = 1; dk >= -1; --dk) {
          const int si = i - di, sj = j - dj, sk = k - dk;  // the source whose offset (di, dj, dk) lands here
          if (si < 1 || si > sz - 2 || sj < 1 || sj > sy - 2 || sk < 1 || sk > sx - 2) continue;
          const int nz = (di != 0) + (dj != 0) + (dk != 0);
          acc += lut[nz][in[((size_t)si * sy + sj) * sx + sk]];
        }
    const double c = acc / div;
    out[t] = (unsigned char)(int)((c > 0 ? (c < 1 ? c : 1) : 0) * 255);
  }
}

static int genvol_spheres_device(smk_ctx *c, unsigned seed, int sx, int sy, int sz, unsigned char *out) {
  GenvolTables T;
  GlibcRand rnd(seed);
  genvol_perlin_init(rnd, T);  // genvol main's init() (genvol/main.cpp:118-120)
  genvol_perlin_init(rnd, T);  // noise3's own first-call init (perlin.c:84-87): these are the tables in use
  GenvolTables *d_tab = nullptr;
  unsigned char *d_tmp = nullptr;
  const size_t n = (size_t)sx * sy * sz;
  PCHK(c, hipMalloc((void **)&d_tab, sizeof T));
  PCHK(c, hipMalloc((void **)&d_tmp, n));
  PCHK(c, hipMemcpy(d_tab, &T, sizeof T, hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(smk_k_genvol_spheres, dim3(blocks), dim3(256), 0, c->stream, (const GenvolTables *)d_tab, sx, sy, sz, 4, 10, .7, 3.f, 3.f, 3.f,
                     2.f, 2.f, d_tmp);
  hipLaunchKernelGGL(smk_k_genvol_blur, dim3(blocks), dim3(256), 0, c->stream, (const unsigned char *)d_tmp, sx, sy, sz, 1.f, 1.f, 1.f, .7f, out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_tab);
  (void)hipFree(d_tmp);
  PCHK(c, e);
  return 0;
}

extern "C" int smk_synth_volume_device(smk_ctx *c, int kind, unsigned seed, int sx, int sy, int sz, void *out) {
  if (!c) return 1;
  PCHK(c, hipSetDevice(c->device));
  if (!out || (kind != 0 && kind != 1) || sx < 1 || sy < 1 || sz < 1) {
    c->err = "smk_synth_volume_device: bad arguments (kind 0: smooth noisy shells, 1: genvol spheres)";
    return 1;
  }
  if (kind == 1) return genvol_spheres_device(c, seed, sx, sy, sz, (unsigned char *)out);
  size_t n = (size_t)sx * sy * sz;
  unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(smk_k_synth, dim3(blocks), dim3(256), 0, c->stream, kind, seed, sx, sy, sz, (unsigned char *)out);
  PCHK(c, hipGetLastError());
  PCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
