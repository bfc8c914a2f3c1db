// OFDM_demodulator / OFDM_modulator on the wave-local transform of the fused chain (chain_fast_core.hpp) for
// Nfft = 512 * NW, NW = 1, 2, 4, 8: persistent workgroups of 64*NW threads, the next symbol's samples in flight while
// this one is transformed, the output column collected in LDS and written out in contiguous rows.  The generic
// workgroup FFT of ofdm_modem.hip stays for the other sizes (64..256, 8192).
//   demodulator: drop rows 1..Tg, fft per column, unscaled                            (T5/OFDM_demodulator.m:2-10)
//   modulator:   x = ifft(X) = conj(fft(conj(X))) / Nfft, y = [x(end-Tg+1:end); x]    (T5/OFDM_modulator.m:2-11)
#include <algorithm>

#include "chain_fast_core.hpp"

namespace ofdm {

template <typename T, int NW, bool MOD>
__global__ __launch_bounds__(64 * NW) void modem_wave_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                                             const cx<T>* __restrict__ tw, int64_t n_symb, int t_guard) {
  constexpr int N = 512 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* lwv = (cx<T>*)smem;                                   // [NW][WAVE_LDS_ELEMS] exchange / private / output staging
  cx<T>* const ex = lwv;
  cx<T>* twl = lwv + NW * WAVE_LDS_ELEMS;                      // [WAVE_TW_ELEMS]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gid = threadIdx.x;
  DifTw<T, NW> dt;
  wave_tw_fill<T, NW>(twl, tw);
  dif_tw_init<T, NW>(dt, gid, tw);
  cx<T> twb[7];
#pragma unroll
  for (int t = 1; t < 8; ++t) twb[t - 1] = tw[(t * (lane & 7) * 8) * NW];
  __syncthreads();
  const int64_t in_stride = MOD ? N : N + t_guard, in_off = MOD ? 0 : t_guard;
  const int64_t out_stride = MOD ? N + t_guard : N;
  const T scale = MOD ? T(1) / T(N) : T(1);
  cx<T> v[8], nx[8];
  if ((int64_t)blockIdx.x < n_symb) frame_load<T, NW>(nx, in + (int64_t)blockIdx.x * in_stride + in_off, gid, lane);
  for (int64_t s = blockIdx.x; s < n_symb; s += gridDim.x) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = MOD ? conj(nx[e]) : nx[e];
    if (s + gridDim.x < n_symb) frame_load<T, NW>(nx, in + (s + gridDim.x) * in_stride + in_off, gid, lane);
    if constexpr (NW > 1) {
      dif_stage<T, NW>(v, dt);
      __syncthreads();                                         // the previous column has been written out
      dif_scatter<T, NW>(v, gid, ex);
      __syncthreads();
      dif_gather<T>(v, wave, lane, ex);
    } else {
      __syncthreads();
    }
    wave_fft512<T, false>(v, lane, twb, twl, lwv + wave * WAVE_LDS_ELEMS);
    __syncthreads();                                           // every wavefront is done with its private region
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const cx<T> r = MOD ? conj(v[t]) * scale : v[t];
      lwv[NW * (lane + 64 * t) + wave] = r;                    // bin k = NW (lane + 64 t) + wave
    }
    __syncthreads();
    cx<T>* dst = out + s * out_stride;                         // written once, read by a later launch: nontemporal stores
    if constexpr (MOD) {
      for (int i = gid; i < N + t_guard; i += 64 * NW) nt_store(dst + i, lwv[i < t_guard ? N - t_guard + i : i - t_guard]);   // :8-9
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) nt_store(dst + gid + 64 * NW * e, lwv[gid + 64 * NW * e]);
    }
  }
}

template <typename T, int NW, bool MOD>
static int modem_wave_launch(const void* in, void* out, const void* tw, int64_t n_symb, int t_guard) {
  const size_t dyn = sizeof(cx<T>) * ((size_t)NW * WAVE_LDS_ELEMS + WAVE_TW_ELEMS);
  auto kern = modem_wave_kernel<T, NW, MOD>;
  const int per_cu = resident_blocks_per_cu((const void*)kern, 64 * NW, dyn);
  const unsigned grid = (unsigned)std::min<int64_t>(n_symb, (int64_t)ctx().num_cu * per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), dyn, ctx().stream, (const cx<T>*)in, (cx<T>*)out, (const cx<T>*)tw,
                     n_symb, t_guard);
  return check_launch(MOD ? "modem_wave_kernel(mod)" : "modem_wave_kernel(demod)");
}

bool modem_wave_supported(int nfft) {
  return (nfft == 512 || nfft == 1024 || nfft == 2048 || nfft == 4096) && !getenv("OFDM_MODEM_GENERIC");
}

// device pointers; modulate = false: OFDM_demodulator, true: OFDM_modulator
bool modem_run_supported(int nfft, int t_guard, bool f64, bool mod);                             // ofdm_modem_run.hip
int modem_run_launch(const void* in, void* out, const void* tw, int nfft, int64_t n_symb, int t_guard, bool mod);

int modem_wave_run(const void* in, void* out, int nfft, int64_t n_symb, int t_guard, bool f64, bool modulate) {
  const void* tw = nullptr;
  OFDM_TRY(get_twiddles(nfft, f64, &tw));
  // fp32, Nfft 1024 / 2048: one wavefront per run of symbols (no workgroup barrier, whole-line stores)
  if (modem_run_supported(nfft, t_guard, f64, modulate)) return modem_run_launch(in, out, tw, nfft, n_symb, t_guard, modulate);
#define MW_CALL(NWV)                                                                                             \
  if (f64) return modulate ? modem_wave_launch<double, NWV, true>(in, out, tw, n_symb, t_guard)                  \
                           : modem_wave_launch<double, NWV, false>(in, out, tw, n_symb, t_guard);                \
  return modulate ? modem_wave_launch<float, NWV, true>(in, out, tw, n_symb, t_guard)                            \
                  : modem_wave_launch<float, NWV, false>(in, out, tw, n_symb, t_guard)
  switch (nfft / 512) {
    case 1: MW_CALL(1);
    case 2: MW_CALL(2);
    case 4: MW_CALL(4);
    case 8: MW_CALL(8);
  }
#undef MW_CALL
  set_error("modem_wave_run: unsupported Nfft %d", nfft);
  return OFDM_ERR_UNSUPPORTED;
}

}  // namespace ofdm
