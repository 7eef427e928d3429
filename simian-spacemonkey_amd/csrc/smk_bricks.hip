// smk_bricks.hip -- which 8x8x8-cell bricks of the stored volume can hold a visible sample under the
// current (value, gradient) table: the slice-ring kernel neither streams nor samples the rest.
//
// The reference draws every slice of every brick (VolumeRenderer.cpp:507-741, NV20VolRen3D.cpp:852-1083)
// and lets the blend unit discard what the table made transparent.  A sample whose table lookup touches
// only texels of alpha 0 contributes exactly nothing to the frame (kernels S and G already stop such a
// sample at the table's occupancy bit, after interpolating it); this file moves that decision in front
// of the fetch:
//
//   per volume upload   smk_k_brick_minmax: the range of the first two channels over the (8+1)^3 voxels
//                       the cells of a brick touch  ->  float4 {vmin, vmax, gmin, gmax} per brick
//   per table refresh   smk_k_occ_sat: summed-area table of the occupancy bitmap (bit (t, s): the
//                       bilinear lookup based at texel (s, t) can be non-transparent);
//                       smk_k_brick_flags: a brick is flagged 1 when ANY base texel its value range can
//                       reach has its bit set.  The range is widened by one texel on every side: an
//                       interpolated channel is a chain of fma lerps between the corner values and may
//                       leave their range by a rounding error, never by a texel.
//
// A clear flag therefore means: every sample whose cell lies in the brick ends at a clear occupancy bit,
// i.e. its alpha is exactly 0 -- skipping it changes no bit of the frame (tests: with and without the
// flags, bit for bit, tests/test_gpu_bricks.py).
#include "smk_device.h"

namespace {

constexpr int BL = SMK_BRICK_LOG2, BR = 1 << SMK_BRICK_LOG2;

// one wave per brick: the lanes share the brick's (BR+1)^3 voxels, then reduce
template <int DT>
__global__ __launch_bounds__(256) void smk_k_brick_minmax(const void *vox, int Dx, int Dy, int Dz, int nbx, int nby, int nbz,
                                                           float4 *mm) {
  const long long brick = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (brick >= (long long)nbx * nby * nbz) return;
  const int bx = (int)(brick % nbx), by = (int)((brick / nbx) % nby), bz = (int)(brick / ((long long)nbx * nby));
  const int x0 = bx << BL, y0 = by << BL, z0 = bz << BL;
  const int ex = min(BR + 1, Dx - x0), ey = min(BR + 1, Dy - y0), ez = min(BR + 1, Dz - z0);
  const int n = ex * ey * ez;
  float vmin = 3.0e38f, vmax = -3.0e38f, gmin = 3.0e38f, gmax = -3.0e38f;
  bool bad = false;
  for (int i = lane; i < n; i += 64) {
    const int x = i % ex, y = (i / ex) % ey, z = i / (ex * ey);
    const size_t o = ((size_t)(z0 + z) * Dy + (y0 + y)) * Dx + (x0 + x);
    float v, g;
    if (DT == 0) {
      const uint32_t d = ((const uint2 *)vox)[o].x;
      v = (float)(d & 0xffu) * SMK_INV255;
      g = (float)((d >> 8) & 0xffu) * SMK_INV255;
    } else {
      const float4 f = ((const float4 *)vox)[o];
      v = f.x;
      g = f.y;
    }
    // (fminf / fmaxf drop a NaN operand: a brick that mixes NaN and finite voxels would end with a finite range, while a
    //  sample interpolated from a NaN corner classifies at base texel 0 -- possibly visible, possibly outside that range)
    if (!(fabsf(v) <= 3.0e38f) || !(fabsf(g) <= 3.0e38f)) bad = true;
    vmin = fminf(vmin, v);
    vmax = fmaxf(vmax, v);
    gmin = fminf(gmin, g);
    gmax = fmaxf(gmax, g);
  }
  // (a brick that saw one is stored with an inverted range: smk_k_brick_flags flags it whatever the table)
  for (int o = 32; o > 0; o >>= 1) {
    vmin = fminf(vmin, __shfl_xor(vmin, o));
    vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    gmin = fminf(gmin, __shfl_xor(gmin, o));
    gmax = fmaxf(gmax, __shfl_xor(gmax, o));
  }
  const bool any_bad = __any(bad);  // (all 64 lanes vote: not inside the one-lane store)
  if (lane == 0) mm[brick] = any_bad ? make_float4(1.0f, 0.0f, 1.0f, 0.0f) : make_float4(vmin, vmax, gmin, gmax);
}

// sat[(t + 1) * (sv + 1) + (s + 1)] = number of set bits (t', s') with t' <= t, s' <= s.  A wave per column s: the bits of
// row t up to column s are popcounts of the row's words, each lane sums them over its run of consecutive rows, a wave
// scan makes the runs' offsets, and a second pass over the same words writes the running sums -- no load depends on a
// store, and sv + 1 waves share the work (the extra one clears row 0 and column 0).  (Rounds 2-3: ONE workgroup, a thread
// per column walking all rows: 150 us for a 256 x 256 table -- longer than the frame of a 1/8 shard, whose camera moves.)
__global__ __launch_bounds__(64) void smk_k_occ_sat(const uint32_t *occ, int roww, int sv, int sg, uint32_t *sat) {
  const int s = blockIdx.x, lane = threadIdx.x, pitch = sv + 1;
  if (s >= sv) {
    for (int i = lane; i < pitch; i += 64) sat[i] = 0;
    for (int t = lane; t < sg; t += 64) sat[(size_t)(t + 1) * pitch] = 0;
    return;
  }
  const int w = s >> 5;
  const uint32_t last = 0xffffffffu >> (31 - (s & 31));  // bits 0..s of word w
  const int R = (sg + 63) >> 6;                           // rows per lane: t = lane * R + r
  auto row_bits = [&](int t) -> uint32_t {
    const uint32_t *row = occ + (size_t)t * roww;
    uint32_t n = __popc(row[w] & last);
    for (int k = 0; k < w; ++k) n += __popc(row[k]);
    return n;
  };
  uint32_t mine = 0;
  for (int r = 0; r < R; ++r) {
    const int t = lane * R + r;
    if (t < sg) mine += row_bits(t);
  }
  uint32_t incl = mine;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  uint32_t run = incl - mine;
  for (int r = 0; r < R; ++r) {
    const int t = lane * R + r;
    if (t < sg) {
      run += row_bits(t);
      sat[(size_t)(t + 1) * pitch + s + 1] = run;
    }
  }
}

__global__ __launch_bounds__(256) void smk_k_brick_flags(const float4 *mm, long long nbricks, const uint32_t *sat, int sv, int sg,
                                                          unsigned char *flags, unsigned *count) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  bool set = false;
  if (b < nbricks) {
  const float4 r = mm[b];
  // base texel of a channel value c: floor(clamp(c * size - 0.5, 0, size - 1)), at most size - 2 (smk_lin_clamp)
  auto base = [](float c, int size) -> int {
    const float x = fminf(fmaxf(__fmaf_rn(c, (float)size, -0.5f), 0.0f), (float)(size - 1));
    return min((int)x, max(size - 2, 0));
  };
  const int s_lo = max(base(r.x, sv) - 1, 0), s_hi = min(base(r.y, sv) + 1, sv - 1);
  const int t_lo = max(base(r.z, sg) - 1, 0), t_hi = min(base(r.w, sg) + 1, sg - 1);
  const int pitch = sv + 1;
  const uint32_t n = sat[(size_t)(t_hi + 1) * pitch + s_hi + 1] - sat[(size_t)t_lo * pitch + s_hi + 1] -
                     sat[(size_t)(t_hi + 1) * pitch + s_lo] + sat[(size_t)t_lo * pitch + s_lo];
  // (a brick that holds a non-finite voxel comes with an inverted range, smk_k_brick_minmax: flagged)
  set = n != 0 || !(r.x <= r.y) || !(r.z <= r.w);
  flags[b] = set ? 1 : 0;
  }
  // how many bricks are flagged: when nearly all are, the flags only cost their set-up and the caller drops them
  const unsigned long long m = __ballot(set);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned)__popcll(m));
}

// out[b] = any flag within (rx, ry, rz) bricks of b: where a perturbed fetch (displaced by at most that far) can land
__global__ __launch_bounds__(256) void smk_k_brick_dilate(const unsigned char *flags, int nbx, int nby, int nbz, int rx, int ry, int rz,
                                                           unsigned char *out) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= (long long)nbx * nby * nbz) return;
  const int bx = (int)(b % nbx), by = (int)((b / nbx) % nby), bz = (int)(b / ((long long)nbx * nby));
  unsigned any = 0;
  for (int z = max(bz - rz, 0); z <= min(bz + rz, nbz - 1); ++z)
    for (int y = max(by - ry, 0); y <= min(by + ry, nby - 1); ++y)
      for (int x = max(bx - rx, 0); x <= min(bx + rx, nbx - 1); ++x) any |= flags[((size_t)z * nby + y) * nbx + x];
  out[b] = any ? 1 : 0;
}

}  // namespace

hipError_t smk_bricks_dilate(const unsigned char *flags, const int nb[3], const int r[3], unsigned char *out, hipStream_t s) {
  const long long nbricks = (long long)nb[0] * nb[1] * nb[2];
  hipLaunchKernelGGL(smk_k_brick_dilate, dim3((unsigned)((nbricks + 255) / 256)), dim3(256), 0, s, flags, nb[0], nb[1], nb[2], r[0], r[1], r[2], out);
  return hipGetLastError();
}

hipError_t smk_bricks_minmax(const void *vox, int dtype, const int D[3], const int nb[3], float4 *mm, hipStream_t s) {
  const long long nbricks = (long long)nb[0] * nb[1] * nb[2];
  const unsigned blocks = (unsigned)((nbricks + 3) / 4);
  if (dtype == 0) hipLaunchKernelGGL(smk_k_brick_minmax<0>, dim3(blocks), dim3(256), 0, s, vox, D[0], D[1], D[2], nb[0], nb[1], nb[2], mm);
  else hipLaunchKernelGGL(smk_k_brick_minmax<1>, dim3(blocks), dim3(256), 0, s, vox, D[0], D[1], D[2], nb[0], nb[1], nb[2], mm);
  return hipGetLastError();
}

hipError_t smk_bricks_flags(const float4 *mm, const int nb[3], const uint32_t *occ, int roww, int sv, int sg, uint32_t *sat,
                            unsigned char *flags, unsigned *count, hipStream_t s) {
  const long long nbricks = (long long)nb[0] * nb[1] * nb[2];
  hipLaunchKernelGGL(smk_k_occ_sat, dim3((unsigned)sv + 1), dim3(64), 0, s, occ, roww, sv, sg, sat);
  hipError_t e = hipMemsetAsync(count, 0, 4, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(smk_k_brick_flags, dim3((unsigned)((nbricks + 255) / 256)), dim3(256), 0, s, mm, nbricks, sat, sv, sg, flags, count);
  return hipGetLastError();
}
