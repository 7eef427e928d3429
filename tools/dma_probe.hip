// dma_probe.hip -- developer probe (GPU box only): what rate do LDS-DMA loader waves sustain on
// MI355X for the access patterns the slice-ring kernel uses?  No consumers: every loader wave
// streams `nchunks` 1-KiB wave-instructions into a private LDS ring behind a counted vmcnt.
//
//   hipcc -O3 --offload-arch=gfx950 tools/dma_probe.hip -o gpurun_out/dma_probe && gpurun_out/dma_probe
//
// pattern 0: contiguous 1 KiB per chunk, each wave walks its own region of a 4 GiB buffer
// pattern 1: window rows: a chunk = 2 rows x 512 B, rows 16 KiB apart, 19 chunks per "slice",
//            slices 16 MiB apart (1024^3 x 16 B volume, 32-voxel-wide window)
// pattern 2: like 1 with only 30 of every 32 lanes active
// pattern 3: like 1 but every slice re-reads slice 0 (L2-hot)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

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

// each loader wave: nchunks instructions, `fly` kept in flight, LDS ring of `fly`+1 KiB-slots per wave
template <int PATTERN>
__global__ void probe(const char *buf, size_t bufbytes, int nchunks, int fly, int nwaves, long long *cycles) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * nwaves + wave;  // global loader index
  const int ringslots = fly + 1;
  unsigned char *ring = smem + (size_t)wave * ringslots * 1024;
  long long t0 = __builtin_amdgcn_s_memtime();
  if (PATTERN == 0) {
    // contiguous: wave gw streams region [gw*nchunks KiB ...)
    size_t base = ((size_t)gw * nchunks * 1024) % (bufbytes - (size_t)nchunks * 1024 - 4096);
    const char *src = buf + base;
    const unsigned voff = lane * 16;
    int slot = 0, inflight = 0;
    for (int c = 0; c < nchunks; ++c) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + voff), (lds_ptr_t)(ring + slot * 1024), 16, 0, 0);
      src += 1024;
      if (++slot == ringslots) slot = 0;
      if (++inflight > fly) {
        wait_vmcnt(fly);
        --inflight;
      }
    }
  } else {
    // window rows: tile index = blockIdx.x -> window origin; loader `wave` of nwaves takes chunks wave, wave+nwaves..
    const unsigned strideVb = 16384;
    const size_t strideSb = (size_t)16384 * 1024;
    const int lrow = lane >> 5, lcol = lane & 31;
    const bool ok = PATTERN == 2 ? lcol < 30 : true;
    const unsigned voff = lrow * strideVb + lcol * 16;
    const int tx = blockIdx.x % 40, ty = (blockIdx.x / 40) % 30;
    const size_t origin = (size_t)ty * 32 * strideVb + (size_t)tx * 24 * 16;
    const int chunks = 19;  // per slice
    int slot = 0, inflight = 0, issued = 0;
    for (int sl = 0; issued < nchunks; ++sl) {
      const size_t s_src = PATTERN == 3 ? 0 : (size_t)(sl % 1000) * strideSb;
      const char *src = buf + s_src + origin + (size_t)(wave * 2) * strideVb;
      for (int c = wave; c < chunks; c += nwaves) {
        if (ok) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + voff), (lds_ptr_t)(ring + slot * 1024), 16, 0, 0);
        src += (size_t)(nwaves * 2) * strideVb;
        if (++slot == ringslots) slot = 0;
        ++issued;
        if (++inflight > fly) {
          wait_vmcnt(fly);
          --inflight;
        }
      }
    }
  }
  wait_vmcnt(0);
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[gw] = t1 - t0;
}

// lean issue loop: saddr-form LDS-DMA in one asm statement per chunk, 8x unrolled, immediate vmcnt
template <int FLY>
__global__ void probe_lean(const char *buf, size_t bufbytes, int nchunks, int nwaves, long long *cycles) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * nwaves + wave;
  const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)smem + wave * (FLY + 8) * 1024;
  long long t0 = __builtin_amdgcn_s_memtime();
  size_t base = ((size_t)gw * nchunks * 1024) % (bufbytes - (size_t)nchunks * 1024 - 4096);
  const char *src = buf + base;
  const unsigned voff = lane * 16;
  unsigned keep;
  for (int c = 0; c < nchunks; c += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned dst = ring + (unsigned)((c + k) % (FLY + 8)) * 1024;  // any slot pattern will do for a rate probe
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(voff), "s"(dst), "s"(src)
                   : "memory");
      src += 1024;
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(FLY) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[gw] = t1 - t0;
}

// lean loop over scattered row pieces: every wave-instruction fetches ROWS rows of 1024/ROWS bytes
// (ROWS = 1, 2, 4), rows 16 KiB apart, slices 16 MiB apart; every workgroup walks the slices from
// its own phase (as image tiles at different depths do) from an origin that is only 16-B aligned
template <int ROWS, int DEPHASE, int BYTES = 16>
__global__ void probe_rows(const char *buf, int nchunks, int nwaves, int active_lanes, long long *cycles, int hotmod = 1000) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * nwaves + wave;
  const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)smem + wave * 32 * 1024;
  constexpr int LPR = 64 / ROWS;  // lanes per row
  const unsigned voff = (unsigned)(lane / LPR) * 16384u + (unsigned)(lane % LPR) * 16u;
  const bool ok = (lane % LPR) < active_lanes;
  const int tx = blockIdx.x % 16, ty = (blockIdx.x / 16) % 16;
  const size_t origin = (size_t)ty * 60 * 16384 + (size_t)tx * (LPR * 16 - 16) + (DEPHASE ? 48 : 0);
  const int phase = DEPHASE ? (int)((blockIdx.x * 37u) % 997u) : 0;
  const int rows_per_slice = 36;  // window height
  long long t0 = __builtin_amdgcn_s_memtime();
  unsigned keep;
  int issued = 0;
  if (ok) {
    for (int sl = 0; issued < nchunks; ++sl) {
      const char *src = buf + (size_t)((sl + phase) % hotmod) * ((size_t)16384 * 1024) + origin + (size_t)(wave * ROWS) * 16384;
      for (int r = wave * ROWS; r < rows_per_slice; r += nwaves * ROWS) {
        const unsigned dst = ring + (unsigned)(issued & 31) * 1024;
        if (BYTES == 16)
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep)
                       : "v"(voff), "s"(dst), "s"(src)
                       : "memory");
        else  // 12 of every 16 bytes: LDS image with 12-byte lanes
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep)
                       : "v"(voff), "s"(dst), "s"(src)
                       : "memory");
        src += (size_t)(nwaves * ROWS) * 16384;
        ++issued;
        if ((issued & 7) == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[gw] = t1 - t0;
}


// strip-major layout [s][ustrip][v][SWU units of 16 B]: a window = NSTR strips x ROWS rows; every
// wave-instruction reads 1 KiB CONTIGUOUS (64/SWU rows of one strip), a strip's rows are one
// contiguous run of ROWS * SWU * 16 bytes; strips are Dv rows apart, slices 16 MiB apart.
// NT: the DMA carries the nt cache policy.
template <int SWU, int NT>
__global__ void probe_strips(const char *buf, int nslices, int nwaves, int nstr, int rows, long long *cycles) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * nwaves + wave;
  const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)smem + wave * 32 * 1024;
  constexpr int RPI = 64 / SWU;            // rows per wave-instruction
  constexpr unsigned ROWB = SWU * 16;      // bytes per strip row
  const size_t strip_stride = (size_t)1024 * ROWB, slice_stride = (size_t)16384 * 1024;
  const int ipr = (rows + RPI - 1) / RPI;  // instructions per strip
  const int tx = blockIdx.x % 16, ty = (blockIdx.x / 16) % 16;
  const size_t origin = (size_t)(tx * 32 * 16 / ROWB) * strip_stride + (size_t)(ty * 24 + 5) * ROWB;
  const int phase = (int)((blockIdx.x * 37u) % 997u);
  const unsigned voff = lane * 16u;
  long long t0 = __builtin_amdgcn_s_memtime();
  unsigned keep;
  int issued = 0;
  for (int sl = 0; sl < nslices; ++sl) {
    const char *sbase = buf + (size_t)((sl + phase) % 1000) * slice_stride + origin;
    for (int i = wave; i < nstr * ipr; i += nwaves) {
      const int j = i / ipr, r = i - j * ipr;
      const char *src = sbase + (size_t)j * strip_stride + (size_t)(r * RPI) * ROWB;
      const int rows_here = min(RPI, rows - r * RPI);
      const unsigned dst = ring + (unsigned)(issued & 31) * 1024;
      if (lane < rows_here * SWU) {
        if (NT)
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(dst), "s"(src) : "memory");
        else
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(dst), "s"(src) : "memory");
      }
      ++issued;
      if ((issued & 7) == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[gw] = t1 - t0;
}

// the same strip-major HBM layout, but the LDS image stays ROW-major over the window (pitch = nstr * SWU
// units): consecutive lanes walk along a window row, i.e. across strips -- every wave-instruction reads
// 64/SWU pieces of SWU*16 B from up to nstr different strips.  Same bytes as probe_strips, different grouping.
template <int SWU>
__global__ void probe_strips_rm(const char *buf, int nslices, int nwaves, int nstr, int rows, long long *cycles) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * nwaves + wave;
  const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)smem + wave * 32 * 1024;
  constexpr unsigned ROWB = SWU * 16;
  const size_t strip_stride = (size_t)1024 * ROWB, slice_stride = (size_t)16384 * 1024;
  const int pitch = nstr * SWU, units = rows * pitch, ninstr = (units + 63) / 64;
  const int tx = blockIdx.x % 16, ty = (blockIdx.x / 16) % 16;
  const size_t origin = (size_t)(tx * 32 * 16 / ROWB) * strip_stride + (size_t)(ty * 24 + 5) * ROWB;
  const int phase = (int)((blockIdx.x * 37u) % 997u);
  long long t0 = __builtin_amdgcn_s_memtime();
  unsigned keep;
  int issued = 0;
  // per-lane source offsets of this wave's instructions (static: the window shape never changes)
  unsigned voff[8];
  bool okk[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = wave + q * nwaves;
    const int g = 64 * i + lane;
    const int row = g / pitch, col = g - row * pitch;
    voff[q] = (unsigned)((col / SWU) * strip_stride + (size_t)row * ROWB + (col % SWU) * 16);
    okk[q] = i < ninstr && g < units;
  }
  for (int sl = 0; sl < nslices; ++sl) {
    const char *src = buf + (size_t)((sl + phase) % 1000) * slice_stride + origin;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (wave + q * nwaves >= ninstr) break;
      const unsigned dst = ring + (unsigned)(issued & 31) * 1024;
      if (okk[q])
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff[q]), "s"(dst), "s"(src) : "memory");
      ++issued;
      if ((issued & 7) == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[gw] = t1 - t0;
}

template <int SWU, int NT>
static int run_strips(const char *buf, int nwaves, int wgs, int nstr, int rows, int nslices, long long *d_cyc);

#define CK(x)                                                      \
  do {                                                             \
    hipError_t e_ = (x);                                           \
    if (e_ != hipSuccess) {                                        \
      printf("%s: %s\n", #x, hipGetErrorString(e_));               \
      return 1;                                                    \
    }                                                              \
  } while (0)

template <int PATTERN>
static int run(const char *buf, size_t bytes, int nwaves, int wgs, int fly, int nchunks, long long *d_cyc) {
  auto k = probe<PATTERN>;
  size_t lds = (size_t)nwaves * (fly + 1) * 1024;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, bytes, nchunks, fly, nwaves, d_cyc);  // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, bytes, nchunks, fly, nwaves, d_cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> cyc((size_t)wgs * nwaves);
  CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
  double mean = 0;
  for (long long c : cyc) mean += (double)c;
  mean /= cyc.size();
  double total = (double)wgs * nwaves * nchunks * 1024.0;
  printf("pattern %d  waves/WG %d  WGs %4d  in-flight %2d KiB/wave (LDS %3zu KB/WG): %7.3f ms  %7.1f GB/s chip  %6.1f cycles/chunk/wave\n",
         PATTERN, nwaves, wgs, fly, lds / 1024, ms, total / ms / 1e6, mean / nchunks);
  fflush(stdout);
  return 0;
}

template <int FLY>
static int run_lean(const char *buf, size_t bytes, int nwaves, int wgs, int nchunks, long long *d_cyc) {
  auto k = probe_lean<FLY>;
  size_t lds = (size_t)nwaves * (FLY + 8) * 1024;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, bytes, nchunks, nwaves, d_cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, bytes, nchunks, nwaves, d_cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> cyc((size_t)wgs * nwaves);
  CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
  double mean = 0;
  for (long long c : cyc) mean += (double)c;
  mean /= cyc.size();
  double total = (double)wgs * nwaves * nchunks * 1024.0;
  printf("lean     waves/WG %d  WGs %4d  in-flight %2d..%2d KiB/wave (LDS %3zu KB/WG): %7.3f ms  %7.1f GB/s chip  %6.1f cycles/chunk/wave\n",
         nwaves, wgs, FLY, FLY + 8, lds / 1024, ms, total / ms / 1e6, mean / nchunks);
  fflush(stdout);
  return 0;
}

template <int ROWS, int DEPHASE, int BYTES = 16>
static int run_rows(const char *buf, int nwaves, int wgs, int active, int nchunks, long long *d_cyc, int hotmod = 1000) {
  auto k = probe_rows<ROWS, DEPHASE, BYTES>;
  size_t lds = (size_t)nwaves * 32 * 1024;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, nchunks, nwaves, active, d_cyc, hotmod);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, nchunks, nwaves, active, d_cyc, hotmod);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = (double)wgs * nwaves * nchunks * (ROWS * active * 16.0);
  printf("rows     %d x %4d B per instr (lanes %2d/%2d, %2d of 16 B to LDS)  dephased %d  slices re-read mod %4d  waves/WG %d: %7.3f ms  %7.1f GB/s chip (source bytes)\n", ROWS, active * 16, active,
         64 / ROWS, BYTES, DEPHASE, hotmod, nwaves, ms, bytes / ms / 1e6);
  fflush(stdout);
  return 0;
}

template <int SWU, int NT>
static int run_strips(const char *buf, int nwaves, int wgs, int nstr, int rows, int nslices, long long *d_cyc) {
  auto k = probe_strips<SWU, NT>;
  size_t lds = (size_t)nwaves * 32 * 1024;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, nslices, nwaves, nstr, rows, d_cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, nslices, nwaves, nstr, rows, d_cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = (double)wgs * nslices * nstr * rows * SWU * 16.0;
  printf("strips   %d strips x %2d rows x %3d B (runs of %5d B)  nt %d  waves/WG %d: %7.3f ms  %7.1f GB/s chip (source bytes)\n", nstr, rows,
         SWU * 16, rows * SWU * 16, NT, nwaves, ms, bytes / ms / 1e6);
  fflush(stdout);
  return 0;
}

template <int SWU>
static int run_strips_rm(const char *buf, int nwaves, int wgs, int nstr, int rows, int nslices, long long *d_cyc) {
  auto k = probe_strips_rm<SWU>;
  size_t lds = (size_t)nwaves * 32 * 1024;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, nslices, nwaves, nstr, rows, d_cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(nwaves * 64), lds, 0, buf, nslices, nwaves, nstr, rows, d_cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = (double)wgs * nslices * nstr * rows * SWU * 16.0;
  printf("strips-rm %d strips x %2d rows x %3d B, row-major LDS image  waves/WG %d: %7.3f ms  %7.1f GB/s chip (source bytes)\n", nstr, rows,
         SWU * 16, nwaves, ms, bytes / ms / 1e6);
  fflush(stdout);
  return 0;
}

int main(int argc, char **argv) {
  size_t bytes = (size_t)17 << 30;
  char *buf;
  CK(hipMalloc((void **)&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  long long *d_cyc;
  CK(hipMalloc((void **)&d_cyc, 8 * 65536));
  const int nchunks = 8192;  // 8 MiB per wave

  if (argc > 1 && !strcmp(argv[1], "strips2")) {
    for (int rep = 0; rep < 2; ++rep) {
      if (run_rows<2, 1>(buf, 4, 256, 30, 8192 / 4, d_cyc)) return 1;
      if (run_strips<4, 0>(buf, 4, 256, 12, 35, 400, d_cyc)) return 1;
      if (run_strips_rm<4>(buf, 4, 256, 12, 35, 400, d_cyc)) return 1;
      if (run_strips<8, 0>(buf, 4, 256, 7, 35, 400, d_cyc)) return 1;
      if (run_strips_rm<8>(buf, 4, 256, 7, 35, 400, d_cyc)) return 1;
      if (run_strips_rm<8>(buf, 4, 256, 6, 35, 400, d_cyc)) return 1;
      if (run_strips_rm<2>(buf, 4, 256, 24, 35, 400, d_cyc)) return 1;
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "strips")) {
    // the slice-ring kernel's north-star window (44 x 35 voxels of 16 B) in three HBM layouts
    for (int nw : {2, 4}) {
      if (run_rows<2, 1>(buf, nw, 256, 30, 8192 / nw, d_cyc)) return 1;          // row-major: 480-B pieces 16 KiB apart
      if (run_strips<4, 0>(buf, nw, 256, 12, 35, 400, d_cyc)) return 1;          // 4-voxel strips
      if (run_strips<4, 1>(buf, nw, 256, 12, 35, 400, d_cyc)) return 1;
      if (run_strips<8, 0>(buf, nw, 256, 7, 35, 400, d_cyc)) return 1;           // 8-voxel strips
      if (run_strips<8, 1>(buf, nw, 256, 7, 35, 400, d_cyc)) return 1;
      if (run_strips<16, 0>(buf, nw, 256, 4, 35, 400, d_cyc)) return 1;          // 16-voxel strips
      if (run_strips<4, 0>(buf, nw, 256, 12, 48, 400, d_cyc)) return 1;          // whole 16-row instructions
      if (run_strips<8, 0>(buf, nw, 256, 6, 40, 400, d_cyc)) return 1;
    }
    return 0;
  }
  for (int nw : {1, 2, 4}) {   // scattered row pieces from cache (2 slices re-read) against HBM
    if (run_rows<2, 1>(buf, nw, 256, 32, nchunks / nw, d_cyc, 2)) return 1;
    if (run_rows<2, 1>(buf, nw, 256, 30, nchunks / nw, d_cyc, 2)) return 1;
    if (run_rows<1, 1>(buf, nw, 256, 64, nchunks / nw, d_cyc, 2)) return 1;
    if (run_rows<2, 1>(buf, nw, 256, 30, nchunks / nw, d_cyc)) return 1;
    if (run_rows<1, 1>(buf, nw, 256, 64, nchunks / nw, d_cyc)) return 1;
  }
  return 0;
  for (int nw : {1, 2, 4}) {
    if (run_rows<2, 1, 12>(buf, nw, 256, 32, nchunks / nw, d_cyc)) return 1;
    if (run_rows<2, 1, 12>(buf, nw, 256, 30, nchunks / nw, d_cyc)) return 1;
    if (run_rows<2, 0>(buf, nw, 256, 32, nchunks / nw, d_cyc)) return 1;
    if (run_rows<2, 1>(buf, nw, 256, 32, nchunks / nw, d_cyc)) return 1;
    if (run_rows<2, 1>(buf, nw, 256, 30, nchunks / nw, d_cyc)) return 1;
    if (run_rows<1, 1>(buf, nw, 256, 64, nchunks / nw, d_cyc)) return 1;
    if (run_rows<1, 1>(buf, nw, 256, 60, nchunks / nw, d_cyc)) return 1;
    if (run_rows<4, 1>(buf, nw, 256, 16, nchunks / nw, d_cyc)) return 1;
  }
  return 0;
  for (int nw : {1}) {
    if (run_lean<8>(buf, bytes, nw, 256, nchunks, d_cyc)) return 1;
    if (run_lean<24>(buf, bytes, nw, 256, nchunks, d_cyc)) return 1;
    if (nw < 4 && run_lean<48>(buf, bytes, nw, 256, nchunks, d_cyc)) return 1;
  }
  for (int wgs : {256}) {
    for (int nw : {1, 2, 4, 8}) {
      for (int fly : {16}) {
        if ((size_t)nw * (fly + 1) * 1024 > (size_t)(wgs == 512 ? 78 : 158) * 1024) continue;
        if (run<0>(buf, bytes, nw, wgs, fly, nchunks, d_cyc)) return 1;
      }
    }
  }
  for (int nw : {1, 2, 4, 8})
    for (int fly : {16}) {
      if ((size_t)nw * (fly + 1) * 1024 > (size_t)158 * 1024) continue;
      if (run<1>(buf, bytes, nw, 256, fly, nchunks, d_cyc)) return 1;
      if (run<2>(buf, bytes, nw, 256, fly, nchunks, d_cyc)) return 1;
      if (run<3>(buf, bytes, nw, 256, fly, nchunks, d_cyc)) return 1;
    }
  return 0;
}
