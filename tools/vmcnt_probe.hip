// vmcnt_probe.hip -- developer probe (GPU box only): can a wave READ its outstanding vector-memory count
// without waiting?  s_getreg_b32 HW_REG_IB_STS (id 7) carries VM_CNT in bits [3:0] and, on gfx9, its two
// high bits in [23:22].  One wave issues N LDS-DMA loads of cold memory and samples the register right
// after the issue, then in a loop until it reads 0.
//   hipcc -O3 --offload-arch=gfx950 tools/vmcnt_probe.hip -o build/vmcnt_probe && build/vmcnt_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(const char *buf, int n, unsigned *out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)smem;
  const unsigned voff = threadIdx.x * 16;
  unsigned keep;
  for (int c = 0; c < n; ++c) {
    const char *src = buf + (size_t)c * (64u << 20);  // far apart: every load misses
    const unsigned dst = ring + (unsigned)c * 1024;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(src) : "memory");
  }
  int k = 0;
  for (; k < 60; ++k) {
    unsigned r = __builtin_amdgcn_s_getreg((31 << 11) | 7);  // whole register: size 32, offset 0, id 7
    if (threadIdx.x == 0) out[k] = r;
    if (((r & 0xf) | (((r >> 22) & 3) << 4)) == 0) { ++k; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    out[62] = k;
    out[63] = __builtin_amdgcn_s_getreg((31 << 11) | 7);
  }
}

int main() {
  char *buf;
  unsigned *d, h[64];
  if (hipMalloc((void **)&buf, (size_t)40 * (64u << 20) + 4096) != hipSuccess) return 1;
  hipMalloc((void **)&d, sizeof h);
  for (int n : {1, 5, 20, 40}) {
    hipMemset(d, 0xff, sizeof h);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 48 * 1024, 0, buf, n, d);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("issued %2d:", n);
    for (unsigned k = 0; k < h[62] && k < 60; ++k) printf(" %u", (h[k] & 0xf) | (((h[k] >> 22) & 3) << 4));
    printf("   (raw first 0x%08x, after vmcnt(0) 0x%08x)\n", h[0], h[63]);
  }
  return 0;
}
