// dma_layout_probe.hip -- developer probe: where does global_load_lds_dwordx3 put a lane's 12 bytes?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const unsigned *src, unsigned *out) {
  __shared__ __align__(16) unsigned lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned voff = threadIdx.x * 16;  // lane l reads source dwords 4l, 4l+1, 4l+2
  unsigned keep;
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)lds;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %3\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(voff), "s"(dst), "s"(src) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  unsigned h[256], *d, *o, r[1024];
  for (int i = 0; i < 256; ++i) h[i] = i;  // source dword i holds i
  hipMalloc((void **)&d, sizeof h); hipMalloc((void **)&o, sizeof r);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  for (int i = 0; i < 256; ++i) { if (r[i] == 0xdeadbeefu) printf("  .  "); else printf("%4u ", r[i]); if (i % 16 == 15) printf("\n"); }
  return 0;
}
