// valu_probe.hip -- developer probe: issue cost of packed fp32 VALU ops vs scalar ones on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, long long *cyc) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float f = 0.999f;
  const v2f vf = {f, f};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent scalar fmas
      asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                   "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(f));
    } else if (MODE == 1) {  // 4 packed fmas = the same 8 flops-pairs
      asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(vf));
    } else if (MODE == 2) {  // 4 packed adds
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(vf));
    } else {  // 8 scalar adds
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(f));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float *out;
  long long *cyc, h;
  hipMalloc((void **)&out, 256 * 1024 * 4);
  hipMalloc((void **)&cyc, 8);
  const int iters = 20000;
  const char *names[4] = {"8 x v_fma_f32   ", "4 x v_pk_fma_f32", "4 x v_pk_add_f32", "8 x v_add_f32   "};
  for (int wpb : {64, 128, 256, 512}) {  // 1, 2, 4, 8 waves per CU (one block per CU)
    for (int mode = 0; mode < 4; ++mode) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(wpb), 0, 0, out, iters, cyc);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(wpb), 0, 0, out, iters, cyc);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(wpb), 0, 0, out, iters, cyc);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(wpb), 0, 0, out, iters, cyc);
      hipDeviceSynchronize();
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%d waves/CU  %s per iteration: %.1f cycles (wave 0's clock)\n", wpb / 64, names[mode], (double)h / iters);
    }
  }
  return 0;
}
