// Micro-benchmark: VALU op rate of plain v_fma_f32 / v_add_f32 vs packed v_pk_fma_f32 / v_pk_add_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  const float s = seed + threadIdx.x * 1e-7f;
  if (MODE == 0) {          // plain fma, 16 independent chains
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = s + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
  } else if (MODE == 1) {   // packed fma, 8 independent chains of 2 lanes = same flops per iteration
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{s + i, s - i};
    const f2 m{0.999f, 0.998f}, c{0.001f, 0.002f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    f2 r{0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
  } else if (MODE == 2) {   // plain add
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = s + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = a[i] + 1.0001f;
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
  } else {                  // packed add
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{s + i, s - i};
    const f2 c{1.0001f, 1.0002f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = a[i] + c;
    f2 r{0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r.x + r.y;
  }
}
template <int MODE> void run(const char* name, float* d) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4096, blocks = 256 * 8;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double lane_ops = (double)blocks * 256 * iters * 16;      // scalar float ops (fma counted as 1 op)
  printf("%-12s %8.3f ms  %7.2f Tera lane-ops/s\n", name, ms, lane_ops / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_add_f32", d); run<3>("v_pk_add_f32", d);
  return 0;
}
