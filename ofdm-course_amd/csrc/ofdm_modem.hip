// OFDM_modulator / OFDM_demodulator (batched (I)FFT + cyclic prefix), OFDM_map_carriers,
// get_payload, equalize_signal.
#include "fft_core.hpp"

namespace ofdm {

bool modem_wave_supported(int nfft);                                                              // ofdm_modem_wave.hip
int modem_wave_run(const void* in, void* out, int nfft, int64_t n_symb, int t_guard, bool f64, bool modulate);

// ---------------------------------------------------------------------------------------------
// OFDM_demodulator -- T5/OFDM_demodulator.m:2-10: drop rows 1..Tg, fft per column (unscaled).
// Generic form (Nfft 64..256 and 8192; 512..4096 run on the wave-local transform, ofdm_modem_wave.hip):
// one transform per N/8 threads; the CP rows are never read.
// ---------------------------------------------------------------------------------------------
template <typename T, int N>
__global__ __launch_bounds__(fft_wg_threads(N)) void demod_kernel(const cx<T>* __restrict__ y,
                                                                  cx<T>* __restrict__ x,
                                                                  const cx<T>* __restrict__ tw,
                                                                  int64_t n_symb, int t_guard) {
  constexpr int TPX = N / 8;
  constexpr int XPW = fft_xforms_per_wg(N);
  __shared__ cx<T> lds[XPW * fft_lds_elems(N)];
  const int g = threadIdx.x / TPX;
  const int j = threadIdx.x % TPX;
  const int64_t s = (int64_t)blockIdx.x * XPW + g;
  const bool live = s < n_symb;
  cx<T> v[8];
  const cx<T>* src = y + (live ? s : 0) * (int64_t)(N + t_guard) + t_guard;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = live ? src[j + e * TPX] : mk<T>(0, 0);
  wg_fft<T, N, false>(v, j, tw, lds + g * fft_lds_elems(N));
  if (live) {
    cx<T>* dst = x + s * (int64_t)N;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[j + e * TPX] = v[e];
  }
}

// Same transform, keeping only rows 1..n_keep of every column (the carriers the RX chain reads afterwards:
// equalize_signal.m:6 touches 1..N_carrier only) -- x is [n_keep x n_symb].
template <typename T, int N>
__global__ __launch_bounds__(fft_wg_threads(N)) void demod_keep_kernel(const cx<T>* __restrict__ y,
                                                                       cx<T>* __restrict__ x,
                                                                       const cx<T>* __restrict__ tw,
                                                                       int64_t n_symb, int t_guard, int n_keep) {
  constexpr int TPX = N / 8;
  constexpr int XPW = fft_xforms_per_wg(N);
  __shared__ cx<T> lds[XPW * fft_lds_elems(N)];
  const int g = threadIdx.x / TPX;
  const int j = threadIdx.x % TPX;
  const int64_t s = (int64_t)blockIdx.x * XPW + g;
  const bool live = s < n_symb;
  cx<T> v[8];
  const cx<T>* src = y + (live ? s : 0) * (int64_t)(N + t_guard) + t_guard;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = live ? src[j + e * TPX] : mk<T>(0, 0);
  wg_fft<T, N, false>(v, j, tw, lds + g * fft_lds_elems(N));
  if (live) {
    cx<T>* dst = x + s * (int64_t)n_keep;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (j + e * TPX < n_keep) dst[j + e * TPX] = v[e];
  }
}

// ---------------------------------------------------------------------------------------------
// OFDM_modulator -- T5/OFDM_modulator.m:2-11: ifft per column (1/N), CP = last Tg rows prepended.
// ---------------------------------------------------------------------------------------------
template <typename T, int N>
__global__ __launch_bounds__(fft_wg_threads(N)) void mod_kernel(const cx<T>* __restrict__ x,
                                                                cx<T>* __restrict__ y,
                                                                const cx<T>* __restrict__ tw,
                                                                int64_t n_symb, int t_guard) {
  constexpr int TPX = N / 8;
  constexpr int XPW = fft_xforms_per_wg(N);
  __shared__ cx<T> lds[XPW * fft_lds_elems(N)];
  const int g = threadIdx.x / TPX;
  const int j = threadIdx.x % TPX;
  const int64_t s = (int64_t)blockIdx.x * XPW + g;
  const bool live = s < n_symb;
  cx<T> v[8];
  const cx<T>* src = x + (live ? s : 0) * (int64_t)N;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = live ? src[j + e * TPX] : mk<T>(0, 0);
  wg_fft<T, N, true>(v, j, tw, lds + g * fft_lds_elems(N));
  if (live) {
    const T scale = T(1) / T(N);
    cx<T>* dst = y + s * (int64_t)(N + t_guard);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = j + e * TPX;
      const cx<T> val = v[e] * scale;
      dst[t_guard + idx] = val;
      if (idx >= N - t_guard) dst[idx - (N - t_guard)] = val;   // OFDM_modulator.m:8-9
    }
  }
}

// ---------------------------------------------------------------------------------------------
// OFDM_map_carriers -- T5/OFDM_map_carriers.m:2-9.  One thread per output element through a
// per-row role table: role[r] = 0 (zero), d+1 (data row d), -(p+1) (pilot row p).  Building the
// table on the host in "data first, pilots second" order reproduces "pilot wins on overlap".
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void map_carriers_kernel(const cx<T>* __restrict__ payload, const cx<T>* __restrict__ pilots,
                                    const int32_t* __restrict__ role, cx<T>* __restrict__ out,
                                    int nfft, int64_t n_symb, int n_data, int n_pilots, int pilot_scalar) {
  const int64_t total = (int64_t)nfft * n_symb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i % nfft);
    const int64_t s = i / nfft;
    const int ro = role[r];
    cx<T> val = mk<T>(0, 0);
    if (ro > 0) val = payload[s * n_data + (ro - 1)];
    else if (ro < 0) val = pilot_scalar ? pilots[0] : pilots[s * n_pilots + (-ro - 1)];
    out[i] = val;
  }
}

// get_payload -- T5/get_payload.m:2-4
template <typename T>
__global__ void get_payload_kernel(const cx<T>* __restrict__ x, const int32_t* __restrict__ dc0,
                                   cx<T>* __restrict__ out, int nfft, int64_t n_symb, int n_data) {
  const int64_t total = (int64_t)n_data * n_symb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % n_data);
    const int64_t s = i / n_data;
    out[i] = x[s * nfft + dc0[d]];
  }
}

// equalize_signal -- T5/equalize_signal.m:1-8
template <typename T>
__global__ void equalize_kernel(const cx<T>* __restrict__ x, const cx<T>* __restrict__ h,
                                cx<T>* __restrict__ out, int nfft, int64_t n_symb, int n_carrier) {
  const int64_t total = (int64_t)nfft * n_symb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i % nfft);
    out[i] = r < n_carrier ? cdiv(x[i], h[r]) : mk<T>(0, 0);
  }
}

template <typename T, int N>
static int launch_demod(const void* y, void* x, const void* tw, int64_t n_symb, int t_guard) {
  constexpr int XPW = fft_xforms_per_wg(N);
  dim3 grid(cdiv_u(n_symb, XPW)), block(fft_wg_threads(N));
  hipLaunchKernelGGL((demod_kernel<T, N>), grid, block, 0, ctx().stream, (const cx<T>*)y, (cx<T>*)x,
                     (const cx<T>*)tw, n_symb, t_guard);
  return check_launch("demod_kernel");
}

template <typename T, int N>
static int launch_mod(const void* x, void* y, const void* tw, int64_t n_symb, int t_guard) {
  constexpr int XPW = fft_xforms_per_wg(N);
  dim3 grid(cdiv_u(n_symb, XPW)), block(fft_wg_threads(N));
  hipLaunchKernelGGL((mod_kernel<T, N>), grid, block, 0, ctx().stream, (const cx<T>*)x, (cx<T>*)y,
                     (const cx<T>*)tw, n_symb, t_guard);
  return check_launch("mod_kernel");
}

// internal device-pointer entry points reused by other translation units
int demod_device(const void* y, void* x, int nfft, int64_t n_symb, int t_guard, bool f64) {
  if (n_symb == 0) return OFDM_OK;
  if (modem_wave_supported(nfft)) return modem_wave_run(y, x, nfft, n_symb, t_guard, f64, false);
  const void* tw = nullptr;
  OFDM_TRY(get_twiddles(nfft, f64, &tw));
#define CALL(NN)                                                                  \
  if (f64) OFDM_TRY((launch_demod<double, NN>(y, x, tw, n_symb, t_guard)));       \
  else OFDM_TRY((launch_demod<float, NN>(y, x, tw, n_symb, t_guard)));
  OFDM_FFT_DISPATCH(nfft, CALL)
#undef CALL
  return OFDM_OK;
}

template <typename T, int N>
static int launch_demod_keep(const void* y, void* x, const void* tw, int64_t n_symb, int t_guard, int n_keep) {
  constexpr int XPW = fft_xforms_per_wg(N);
  dim3 grid(cdiv_u(n_symb, XPW)), block(fft_wg_threads(N));
  hipLaunchKernelGGL((demod_keep_kernel<T, N>), grid, block, 0, ctx().stream, (const cx<T>*)y, (cx<T>*)x,
                     (const cx<T>*)tw, n_symb, t_guard, n_keep);
  return check_launch("demod_keep_kernel");
}

int demod_keep_device(const void* y, void* x, int nfft, int64_t n_symb, int t_guard, int n_keep, bool f64) {
  if (n_symb == 0) return OFDM_OK;
  const void* tw = nullptr;
  OFDM_TRY(get_twiddles(nfft, f64, &tw));
#define CALL(NN)                                                                             \
  if (f64) OFDM_TRY((launch_demod_keep<double, NN>(y, x, tw, n_symb, t_guard, n_keep)));     \
  else OFDM_TRY((launch_demod_keep<float, NN>(y, x, tw, n_symb, t_guard, n_keep)));
  OFDM_FFT_DISPATCH(nfft, CALL)
#undef CALL
  return OFDM_OK;
}

int mod_device(const void* x, void* y, int nfft, int64_t n_symb, int t_guard, bool f64) {
  if (n_symb == 0) return OFDM_OK;
  if (modem_wave_supported(nfft)) return modem_wave_run(x, y, nfft, n_symb, t_guard, f64, true);
  const void* tw = nullptr;
  OFDM_TRY(get_twiddles(nfft, f64, &tw));
#define CALL(NN)                                                                \
  if (f64) OFDM_TRY((launch_mod<double, NN>(x, y, tw, n_symb, t_guard)));       \
  else OFDM_TRY((launch_mod<float, NN>(x, y, tw, n_symb, t_guard)));
  OFDM_FFT_DISPATCH(nfft, CALL)
#undef CALL
  return OFDM_OK;
}

static unsigned ew_grid(int64_t total) {
  int64_t b = (total + 255) / 256;
  int64_t cap = (int64_t)ctx().num_cu * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// 1-based int32 host index vector -> validated 0-based copy
int to_zero_based(const int32_t* idx1, int n, int limit, std::vector<int32_t>& out, const char* what) {
  out.resize(n);
  for (int i = 0; i < n; ++i) {
    OFDM_ARG(idx1[i] >= 1 && idx1[i] <= limit, "%s: index %d at position %d outside 1..%d", what,
             (int)idx1[i], i + 1, limit);
    out[i] = idx1[i] - 1;
  }
  return OFDM_OK;
}

}  // namespace ofdm

using namespace ofdm;

extern "C" {

int ofdm_OFDM_demodulator(const void* y, void* x, int nfft, int64_t n_symb, int t_guard, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(fft_size_ok(nfft), "OFDM_demodulator: unsupported Nfft %d", nfft);
  OFDM_ARG(n_symb >= 0 && t_guard >= 0, "OFDM_demodulator: negative size");
  OFDM_ARG(y && x || n_symb == 0, "OFDM_demodulator: null pointer");
  Stage st(flags);
  const void* dy; void* dx;
  OFDM_TRY(st.in(y, csize(flags) * (size_t)(nfft + t_guard) * n_symb, &dy));
  OFDM_TRY(st.out(x, csize(flags) * (size_t)nfft * n_symb, &dx));
  OFDM_TRY(demod_device(dy, dx, nfft, n_symb, t_guard, is_f64(flags)));
  return st.finish();
}

int ofdm_OFDM_modulator(const void* x, void* y, int nfft, int64_t n_symb, int t_guard, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(fft_size_ok(nfft), "OFDM_modulator: unsupported Nfft %d", nfft);
  OFDM_ARG(n_symb >= 0 && t_guard >= 0 && t_guard <= nfft, "OFDM_modulator: T_guard must be in 0..Nfft");
  OFDM_ARG(y && x || n_symb == 0, "OFDM_modulator: null pointer");
  Stage st(flags);
  const void* dx; void* dy;
  OFDM_TRY(st.in(x, csize(flags) * (size_t)nfft * n_symb, &dx));
  OFDM_TRY(st.out(y, csize(flags) * (size_t)(nfft + t_guard) * n_symb, &dy));
  OFDM_TRY(mod_device(dx, dy, nfft, n_symb, t_guard, is_f64(flags)));
  return st.finish();
}

int ofdm_OFDM_map_carriers(const void* payload, int64_t n_symb, int nfft, const int32_t* data_carriers,
                           int n_data, const int32_t* pilot_carriers, int n_pilots,
                           const void* pilot_values, int pilot_scalar, void* out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(nfft > 0 && n_symb >= 0 && n_data >= 0 && n_pilots >= 0, "OFDM_map_carriers: bad sizes");
  std::vector<int32_t> dc, pc;
  OFDM_TRY(to_zero_based(data_carriers, n_data, nfft, dc, "OFDM_map_carriers(dataCarriers)"));
  OFDM_TRY(to_zero_based(pilot_carriers, n_pilots, nfft, pc, "OFDM_map_carriers(pilotCarriers)"));
  std::vector<int32_t> role(nfft, 0);
  for (int d = 0; d < n_data; ++d) role[dc[d]] = d + 1;        // :6 (a repeated row keeps the last)
  for (int p = 0; p < n_pilots; ++p) role[pc[p]] = -(p + 1);   // :8 pilots written after data
  Stage st(flags);
  const void *dpay, *dpil, *drole; void* dout;
  OFDM_TRY(st.in(payload, csize(flags) * (size_t)n_data * n_symb, &dpay));
  if (pilot_scalar) OFDM_TRY(st.in(pilot_values, csize(flags), &dpil));
  else OFDM_TRY(st.in(pilot_values, csize(flags) * (size_t)n_pilots * n_symb, &dpil));
  OFDM_TRY(st.upload(role.data(), sizeof(int32_t) * nfft, &drole));
  OFDM_TRY(st.out(out, csize(flags) * (size_t)nfft * n_symb, &dout));
  const int64_t total = (int64_t)nfft * n_symb;
  if (total > 0) {
    if (is_f64(flags))
      hipLaunchKernelGGL(map_carriers_kernel<double>, dim3(ew_grid(total)), dim3(256), 0, ctx().stream,
                         (const c64*)dpay, (const c64*)dpil, (const int32_t*)drole, (c64*)dout, nfft, n_symb,
                         n_data, n_pilots, pilot_scalar);
    else
      hipLaunchKernelGGL(map_carriers_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, ctx().stream,
                         (const c32*)dpay, (const c32*)dpil, (const int32_t*)drole, (c32*)dout, nfft, n_symb,
                         n_data, n_pilots, pilot_scalar);
    OFDM_TRY(check_launch("map_carriers_kernel"));
  }
  return st.finish();
}

int ofdm_get_payload(const void* x, int nfft, int64_t n_symb, const int32_t* data_carriers, int n_data,
                     void* out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(nfft > 0 && n_symb >= 0 && n_data >= 0, "get_payload: bad sizes");
  std::vector<int32_t> dc;
  OFDM_TRY(to_zero_based(data_carriers, n_data, nfft, dc, "get_payload(dataCarriers)"));
  Stage st(flags);
  const void *dx, *ddc; void* dout;
  OFDM_TRY(st.in(x, csize(flags) * (size_t)nfft * n_symb, &dx));
  OFDM_TRY(st.upload(dc.data(), sizeof(int32_t) * n_data, &ddc));
  OFDM_TRY(st.out(out, csize(flags) * (size_t)n_data * n_symb, &dout));
  const int64_t total = (int64_t)n_data * n_symb;
  if (total > 0) {
    if (is_f64(flags))
      hipLaunchKernelGGL(get_payload_kernel<double>, dim3(ew_grid(total)), dim3(256), 0, ctx().stream,
                         (const c64*)dx, (const int32_t*)ddc, (c64*)dout, nfft, n_symb, n_data);
    else
      hipLaunchKernelGGL(get_payload_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, ctx().stream,
                         (const c32*)dx, (const int32_t*)ddc, (c32*)dout, nfft, n_symb, n_data);
    OFDM_TRY(check_launch("get_payload_kernel"));
  }
  return st.finish();
}

int ofdm_equalize_signal(const void* x, int nfft, int64_t n_symb, const void* h_est, int n_carrier,
                         void* out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(nfft > 0 && n_symb >= 0 && n_carrier >= 0 && n_carrier <= nfft,
           "equalize_signal: N_carrier must be in 0..Nfft");
  Stage st(flags);
  const void *dx, *dh; void* dout;
  OFDM_TRY(st.in(x, csize(flags) * (size_t)nfft * n_symb, &dx));
  OFDM_TRY(st.in(h_est, csize(flags) * (size_t)n_carrier, &dh));
  OFDM_TRY(st.out(out, csize(flags) * (size_t)nfft * n_symb, &dout));
  const int64_t total = (int64_t)nfft * n_symb;
  if (total > 0) {
    if (is_f64(flags))
      hipLaunchKernelGGL(equalize_kernel<double>, dim3(ew_grid(total)), dim3(256), 0, ctx().stream,
                         (const c64*)dx, (const c64*)dh, (c64*)dout, nfft, n_symb, n_carrier);
    else
      hipLaunchKernelGGL(equalize_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, ctx().stream,
                         (const c32*)dx, (const c32*)dh, (c32*)dout, nfft, n_symb, n_carrier);
    OFDM_TRY(check_launch("equalize_kernel"));
  }
  return st.finish();
}

}  // extern "C"
