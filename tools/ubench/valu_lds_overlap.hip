// Micro-benchmark: do VALU work and LDS traffic of the same wavefronts overlap on a gfx950 CU, or do their times add?
// Shape of rx_symbols_kernel: 256-thread workgroups, 5 resident per CU (31 KB LDS each), per iteration
// NV independent v_fma_f32 and NL (ds_write_b64 + ds_read_b64) pairs per lane, conflict-free.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int NV, int NL>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float2 lds[];
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-6f + i;
  float2 r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = make_float2(threadIdx.x, i);
  float2* p = lds + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NL; ++j) p[256 * (j & 7)] = r[j & 7];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int j = 0; j < NV / 2; ++j) a[j & 15] = __builtin_fmaf(a[j & 15], 0.999f, 0.001f);
#pragma unroll
    for (int j = 0; j < NL; ++j) r[j & 7] = p[256 * (j & 7) + ((j >> 3) & 1)];
#pragma unroll
    for (int j = 0; j < NV - NV / 2; ++j) a[j & 15] = __builtin_fmaf(a[j & 15], 0.999f, 0.001f);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] += r[i].x * 1e-9f;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NL>
int run(float* d) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 2000, blocks = 256 * 5;
  const size_t dyn = 31 * 1024;
  CK(hipFuncSetAttribute((const void*)k<NV, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  hipLaunchKernelGGL((k<NV, NL>), dim3(blocks), dim3(256), dyn, 0, d, iters);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<NV, NL>), dim3(blocks), dim3(256), dyn, 0, d, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  // per CU: 5 workgroups x 4 wavefronts, per SIMD 5 wavefronts
  const double cyc_per_iter_per_simd = ms * 1e-3 * 2.4e9 / iters;          // at a nominal 2.4 GHz
  printf("NV=%3d NL=%2d: %8.1f us  -> %7.1f cycles/iteration/SIMD(5 waves)  = %5.2f cycles per wave-iteration-instruction\n", NV, NL,
         ms * 1e3, cyc_per_iter_per_simd, cyc_per_iter_per_simd / 5.0 / (NV + 2 * NL + 8 + 0.001));
  return 0;
}

int main() {
  float* d; CK(hipMalloc(&d, 256 * 5 * 256 * 4));
  if (run<192, 0>(d)) return 1;      // VALU only
  if (run<0, 24>(d)) return 1;       // LDS only
  if (run<192, 24>(d)) return 1;     // both (roughly the 8:1 mix of rx_symbols_kernel: 380 VALU, 51 LDS)
  if (run<384, 0>(d)) return 1;
  if (run<0, 48>(d)) return 1;
  if (run<384, 48>(d)) return 1;
  if (run<384, 24>(d)) return 1;
  return 0;
}
