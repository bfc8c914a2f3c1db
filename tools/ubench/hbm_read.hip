// Micro-benchmark: achievable HBM read bandwidth on MI355X for the access shape of rx_symbols_kernel
// (persistent workgroups of 256 threads, 16-byte loads, each workgroup streams its own 258 KB "frame",
// optionally skipping the 2 KB guard interval in front of every 16 KB symbol) versus a flat grid-stride copy-free read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void read_frames(const float4* __restrict__ x, float* out, long n_frames, int sym_f4,
                                                   int guard_f4, int n_symb, int depth) {
  float4 acc = {0, 0, 0, 0};
  const long frame_f4 = (long)(sym_f4 + guard_f4) * n_symb;
  for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
    const float4* p = x + f * frame_f4;
    for (int s = 0; s < n_symb; ++s) {
      const float4* q = p + (long)s * (sym_f4 + guard_f4) + guard_f4;
      for (int i = threadIdx.x; i < sym_f4; i += 256 * 4) {   // 4 loads in flight per thread
        float4 a = q[i], b = q[i + 256], c = q[i + 512], d = q[i + 768];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
        acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
      }
    }
  }
  (void)depth;
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// the same frame walk with nontemporal loads (global_load_dwordx4 ... nt)
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_frames_nt(const v4f* __restrict__ x, float* out, long n_frames, int sym_f4,
                                                      int guard_f4, int n_symb) {
  v4f acc = {0, 0, 0, 0};
  const long frame_f4 = (long)(sym_f4 + guard_f4) * n_symb;
  for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
    const v4f* p = x + f * frame_f4;
    for (int s = 0; s < n_symb; ++s) {
      const v4f* q = p + (long)s * (sym_f4 + guard_f4) + guard_f4;
      for (int i = threadIdx.x; i < sym_f4; i += 256 * 4) {
        const v4f a = __builtin_nontemporal_load(q + i), b = __builtin_nontemporal_load(q + i + 256);
        const v4f c = __builtin_nontemporal_load(q + i + 512), d = __builtin_nontemporal_load(q + i + 768);
        acc += a + b + c + d;
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

__global__ __launch_bounds__(256) void read_flat(const float4* __restrict__ x, float* out, long n_f4) {
  float4 acc = {0, 0, 0, 0};
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n_f4; i += stride) {
    float4 a = x[i], b = x[i + 256], c = x[i + 512], d = x[i + 768];
    acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
    acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// one WAVEFRONT per frame (the shape of rx_symbols_wave_kernel): lane l reads x[l + 64 j], j < 32, of every 2048-sample
// symbol with 8-byte nontemporal loads (W8 = true) or 16-byte ones (lane l reads samples 2l, 2l+1 of every 128)
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool W8>
__global__ __launch_bounds__(512) void read_frames_wave(const float* __restrict__ x, float* out, long n_frames, int n_symb) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (long)gridDim.x * (blockDim.x >> 6);
  float acc = 0;
  for (long f = wave; f < n_frames; f += n_waves) {
    const float* p = x + f * (long)(2304 * 2) * n_symb;
    for (int s = 0; s < n_symb; ++s) {
      const float* q = p + (long)s * 4608 + 512;
      if constexpr (W8) {
        v2f v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = __builtin_nontemporal_load((const v2f*)q + lane + 64 * j);
#pragma unroll
        for (int j = 0; j < 32; ++j) acc += v[j].x + v[j].y;
      } else {
        v4f v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_nontemporal_load((const v4f*)q + lane + 64 * j);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const long n_frames = 8192;
  const int n_symb = 14, sym_f4 = 2048 * 8 / 16, guard_f4 = 256 * 8 / 16;
  const long frame_f4 = (long)(sym_f4 + guard_f4) * n_symb;
  const size_t bytes = (size_t)n_frames * frame_f4 * 16;
  float4* x; float* out;
  hipMalloc(&x, bytes + 65536); hipMalloc(&out, 4);
  hipMemset(x, 0, bytes + 65536);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int wg_per_cu : {2, 4, 5, 8}) {
    for (int guard : {1, 0}) {
      const int g4 = guard ? guard_f4 : 0;
      const int sf4 = guard ? sym_f4 : sym_f4 + guard_f4;      // guard=0: read everything
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        for (int it = 0; it < 10; ++it)
          hipLaunchKernelGGL(read_frames, dim3(256 * wg_per_cu), dim3(256), 0, 0, x, out, n_frames, sf4, g4, n_symb, 0);
        hipEventRecord(b); hipEventSynchronize(b);
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      const double rd = (double)n_frames * n_symb * sf4 * 16;
      printf("frames wg/cu=%d skip_guard=%d: %.1f us/launch, %.0f GB/s\n", wg_per_cu, guard, ms * 100, rd / (ms / 10 * 1e-3) / 1e9);
    }
  }
  for (int wg_per_cu : {4, 5, 8}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      for (int it = 0; it < 10; ++it)
        hipLaunchKernelGGL(read_frames_nt, dim3(256 * wg_per_cu), dim3(256), 0, 0, (const v4f*)x, out, n_frames, sym_f4, guard_f4, n_symb);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double rd = (double)n_frames * n_symb * sym_f4 * 16;
    printf("frames nt wg/cu=%d skip_guard=1: %.1f us/launch, %.0f GB/s\n", wg_per_cu, ms * 100, rd / (ms / 10 * 1e-3) / 1e9);
  }
  for (int w8 : {1, 0}) {
    for (int waves_per_cu : {8, 12, 16}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        for (int it = 0; it < 10; ++it) {
          if (w8) hipLaunchKernelGGL(read_frames_wave<true>, dim3(256 * waves_per_cu / 4), dim3(256), 0, 0, (const float*)x, out, n_frames, n_symb);
          else hipLaunchKernelGGL(read_frames_wave<false>, dim3(256 * waves_per_cu / 4), dim3(256), 0, 0, (const float*)x, out, n_frames, n_symb);
        }
        hipEventRecord(b); hipEventSynchronize(b);
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      const double rd = (double)n_frames * n_symb * sym_f4 * 16;
      printf("wave-per-frame nt %d-byte loads, %d waves/cu: %.1f us/launch, %.0f GB/s\n", w8 ? 8 : 16, waves_per_cu, ms * 100, rd / (ms / 10 * 1e-3) / 1e9);
    }
  }
  for (int wg_per_cu : {4, 8, 16}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      for (int it = 0; it < 10; ++it)
        hipLaunchKernelGGL(read_flat, dim3(256 * wg_per_cu), dim3(256), 0, 0, x, out, (long)(bytes / 16));
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("flat wg/cu=%d: %.1f us/launch, %.0f GB/s\n", wg_per_cu, ms * 100, (double)bytes / (ms / 10 * 1e-3) / 1e9);
  }
  return 0;
}
