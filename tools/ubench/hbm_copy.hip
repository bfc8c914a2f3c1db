// Micro-benchmark: achievable HBM bandwidth of WRITE-only and COPY (read + write) streams on MI355X, the ceilings of the kernels
// that write what they read (OFDM_modulator / OFDM_demodulator stand-alone, the Task-4 demodulator: 3.77 GB in + 1.31 GB out).
// Wave-per-stream shape of the transform kernels: every wavefront walks its own contiguous run, 32 x 8-byte (or 16 x 16-byte)
// nontemporal loads / stores per 16 KB step.  ratio = bytes written per byte read (1 = copy, 1/3 ~ the Task-4 demodulator).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int WRITE_EVERY>   // 0: write only; k >= 1: read every step, write every k-th step's data
__global__ __launch_bounds__(256) void stream_kernel(const v4f* __restrict__ x, v4f* __restrict__ y, long n_steps) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
  v4f acc = {0, 0, 0, 0};
  for (long s = wave; s < n_steps; s += n_waves) {
    const v4f* p = x + s * 1024 + lane;        // 16 KB per step
    v4f v[16];
    if (WRITE_EVERY > 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load(p + 64 * j);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = v4f{(float)s, (float)j, 1.f, 2.f};
    }
    if (WRITE_EVERY == 0 || (s / n_waves) % WRITE_EVERY == 0) {
      v4f* q = y + (WRITE_EVERY <= 1 ? s : s / WRITE_EVERY) * 1024 + lane;
#pragma unroll
      for (int j = 0; j < 16; ++j) __builtin_nontemporal_store(v[j], q + 64 * j);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += v[j];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) y[0] = acc;
}

template <typename K>
static void run(const char* name, K kern, const v4f* x, v4f* y, long n_steps, double bytes, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, x, y, n_steps);
  hipEventRecord(a);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, x, y, n_steps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("%-34s grid %5d  %8.3f ms  %7.2f TB/s (read + write)\n", name, grid, ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
  const long n_steps = 196608;                     // 3 GiB of source
  const size_t bytes = (size_t)n_steps * 16384;
  v4f *x, *y;
  if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
  for (int per_cu : {3, 4, 8}) {
    const int grid = 256 * per_cu;
    run("write only (nt stores)", stream_kernel<0>, x, y, n_steps, (double)bytes, grid);
    run("copy 1:1 (nt loads + nt stores)", stream_kernel<1>, x, y, n_steps, 2.0 * bytes, grid);
    run("read 3 : write 1", stream_kernel<3>, x, y, n_steps, (1.0 + 1.0 / 3.0) * bytes, grid);
  }
  hipFree(x); hipFree(y);
  return 0;
}
