// stage_probe.hip -- developer probe (GPU box only): global -> LDS staging rate per CU, LDS-DMA
// against plain loads + ds_write, from HBM (streaming a 16 GiB buffer) and from cache (every wave
// re-reads a small region).  Question it answers: is the ~5-6 TB/s the slice-ring loaders reach an
// HBM limit or a limit of the LDS-DMA path?
//
//   hipcc -O3 --offload-arch=gfx950 tools/stage_probe.hip -o gpurun_out/stage_probe && gpurun_out/stage_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <vector>

#define CK(x)                                        \
  do {                                               \
    hipError_t e_ = (x);                             \
    if (e_ != hipSuccess) {                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                      \
    }                                                \
  } while (0)

// region = bytes each wave walks before wrapping (power of two); 1 KiB chunks
template <int WIDTH>  // 16: global_load_lds_dwordx4, 4: global_load_lds_dword (256 B per instruction)
__global__ void stage_dma(const char *buf, size_t wave_stride, unsigned region_mask, int nchunks, int nwaves) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * nwaves + wave;
  const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)smem + wave * 32 * 1024;
  const char *base = buf + (size_t)gw * wave_stride;
  const unsigned voff = lane * WIDTH;
  unsigned keep, off = 0;
  for (int c = 0; c < nchunks; c += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned dst = ring + (unsigned)((c + k) & 31) * 1024;
      const char *src = base + off;
      if (WIDTH == 16) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(src) : "memory");
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(dst + p * 256u), "s"(src + p * 256) : "memory");
      }
      off = (off + 1024u) & region_mask;
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// plain loads into registers, K KiB per batch, two batches alternate (one in flight while the other is written)
template <int K>
__global__ void stage_regs(const char *buf, size_t wave_stride, unsigned region_mask, int nchunks, int nwaves) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * nwaves + wave;
  float4 *ring = reinterpret_cast<float4 *>(smem + wave * 32 * 1024) + lane;
  const char *base = buf + (size_t)gw * wave_stride + lane * 16;
  float4 r0[K], r1[K];
  unsigned off = 0;
  auto load = [&](float4 *r) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      r[k] = *reinterpret_cast<const float4 *>(base + off);
      off = (off + 1024u) & region_mask;
    }
  };
  auto store = [&](const float4 *r, int c) {
#pragma unroll
    for (int k = 0; k < K; ++k) ring[((c + k) & 31) * 64] = r[k];
  };
  load(r0);
  for (int c = 0; c < nchunks; c += 2 * K) {
    load(r1);
    store(r0, c);
    load(r0);
    store(r1, c + K);
  }
  store(r0, 0);
  __syncthreads();
  if (smem[threadIdx.x] == 123 && buf == nullptr) printf("x");  // keep the LDS writes
}

template <typename F>
static int time_it(const char *what, F launch, double bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  launch();
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-58s %8.3f ms  %8.1f GB/s chip  %6.2f B/clk/CU @2.4GHz\n", what, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.4);
  fflush(stdout);
  return 0;
}

int main() {
  size_t bytes = (size_t)16 << 30;
  char *buf;
  CK(hipMalloc((void **)&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  const int wgs = 256;
  for (int hot = 0; hot < 2; ++hot)
    for (int nw : {1, 2, 4, 8}) {
      const int nchunks = 8192 / nw * (hot ? 4 : 1);
      // streaming: every wave its own contiguous 16 MiB/nw region; hot: every wave re-reads 32 KiB
      const size_t stride = hot ? (size_t)32768 : ((size_t)16 << 20) / nw * 1;
      const unsigned mask = hot ? 32767u : 0xffffffffu;
      const double total = (double)wgs * nw * nchunks * 1024.0;
      const size_t lds = (size_t)nw * 32 * 1024;
      char name[128];
      auto kd16 = stage_dma<16>;
      auto kd4 = stage_dma<4>;
      auto kr8 = stage_regs<8>;
      auto kr4 = stage_regs<4>;
      CK(hipFuncSetAttribute((const void *)kd16, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      CK(hipFuncSetAttribute((const void *)kd4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      CK(hipFuncSetAttribute((const void *)kr8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      CK(hipFuncSetAttribute((const void *)kr4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      if (lds > 160 * 1024) continue;
      snprintf(name, sizeof name, "%s  %d waves/CU  LDS-DMA dwordx4", hot ? "cache" : "HBM  ", nw);
      if (time_it(name, [&] { hipLaunchKernelGGL(kd16, dim3(wgs), dim3(nw * 64), lds, 0, buf, stride, mask, nchunks, nw); }, total)) return 1;
      snprintf(name, sizeof name, "%s  %d waves/CU  LDS-DMA dword (4 per KiB)", hot ? "cache" : "HBM  ", nw);
      if (time_it(name, [&] { hipLaunchKernelGGL(kd4, dim3(wgs), dim3(nw * 64), lds, 0, buf, stride, mask, nchunks, nw); }, total)) return 1;
      snprintf(name, sizeof name, "%s  %d waves/CU  load dwordx4 + ds_write_b128, 2x8 KiB", hot ? "cache" : "HBM  ", nw);
      if (time_it(name, [&] { hipLaunchKernelGGL(kr8, dim3(wgs), dim3(nw * 64), lds, 0, buf, stride, mask, nchunks, nw); }, total)) return 1;
      snprintf(name, sizeof name, "%s  %d waves/CU  load dwordx4 + ds_write_b128, 2x4 KiB", hot ? "cache" : "HBM  ", nw);
      if (time_it(name, [&] { hipLaunchKernelGGL(kr4, dim3(wgs), dim3(nw * 64), lds, 0, buf, stride, mask, nchunks, nw); }, total)) return 1;
    }
  return 0;
}
