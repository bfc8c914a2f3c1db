// Receiver synchronisation: AutoCorrFunction (sliding CP autocorrelation + plateau search),
// remove_IFO, fine_sync.
#include "ofdm_common.hpp"

namespace ofdm {

int demod_device(const void* y, void* x, int nfft, int64_t n_symb, int t_guard, bool f64);   // ofdm_modem.hip
int cfo_device(const void* y, void* out, int64_t len, double cfo, int nfft, bool f64);       // ofdm_channel.hip

static unsigned ew_grid(int64_t total, int per_block = 256) {
  int64_t b = (total + per_block - 1) / per_block;
  int64_t cap = (int64_t)ctx().num_cu * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------
// AutoCorrFunction.m:3-7.  rho(n) = sum_W x[m] conj(x[m+N]) / sqrt(sum_W |x[m]|^2 * sum_W |x[m+N]|^2)
// for every n.  The reference is O(L*W); here each workgroup owns a tile of ACF_TILE outputs, forms
// the three running sums as tile-local exclusive prefix sums (rows of 256 elements, wave-shuffle
// scan + carry, double accumulation) in LDS and takes S[n+W]-S[n]: O(L) work, each sample read
// twice (once as x[m], once as x[m+N]) and rho written once.
// ---------------------------------------------------------------------------------------------
constexpr int ACF_TILE = 1024;
constexpr int ACF_THREADS = 256;
constexpr int ACF_MAXW = 1024;                       // WidthWindow limit (T_guard <= 1024 <-> Nfft <= 8192)
constexpr int ACF_ELEMS = ACF_TILE + ACF_MAXW;       // prefix entries 0..ACF_ELEMS

struct acf4 { double pr, pi, e1, e2; };

__device__ __forceinline__ acf4 acf_add(acf4 a, acf4 b) { return acf4{a.pr + b.pr, a.pi + b.pi, a.e1 + b.e1, a.e2 + b.e2}; }
__device__ __forceinline__ acf4 acf_shfl_up(acf4 v, int d) {
  return acf4{__shfl_up(v.pr, d, 64), __shfl_up(v.pi, d, 64), __shfl_up(v.e1, d, 64), __shfl_up(v.e2, d, 64)};
}

template <typename T>
__global__ __launch_bounds__(ACF_THREADS) void acf_kernel(const cx<T>* __restrict__ x, int64_t len, int W, int nfft,
                                                          cx<T>* __restrict__ rho, int64_t n_out) {
  __shared__ acf4 S[ACF_ELEMS + 1];                  // exclusive prefix: S[i] = sum_{m<i}
  __shared__ acf4 wtot[ACF_THREADS / 64];
  __shared__ acf4 carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * ACF_TILE;
  const int n_here = (int)((n_out - n0 < ACF_TILE) ? (n_out - n0) : ACF_TILE);
  const int m_cnt = n_here + W - 1;                  // elements needed: m = n0 .. n0+n_here+W-2
  if (tid == 0) { carry_s = acf4{0, 0, 0, 0}; S[0] = acf4{0, 0, 0, 0}; }
  __syncthreads();
  for (int base = 0; base < m_cnt; base += ACF_THREADS) {
    const int i = base + tid;
    acf4 v{0, 0, 0, 0};
    if (i < m_cnt) {
      const int64_t m = n0 + i;                      // m + nfft < len is guaranteed by n_out
      const cx<T> a = x[m], b = x[m + nfft];
      const double ar = a.x, ai = a.y, br = b.x, bi = b.y;
      v.pr = ar * br + ai * bi;                      // a * conj(b)
      v.pi = ai * br - ar * bi;
      v.e1 = ar * ar + ai * ai;
      v.e2 = br * br + bi * bi;
    }
    // inclusive scan inside the wave
    for (int d = 1; d < 64; d <<= 1) {
      acf4 u = acf_shfl_up(v, d);
      if (lane >= d) v = acf_add(v, u);
    }
    if (lane == 63) wtot[wid] = v;
    __syncthreads();
    acf4 off = carry_s;
    for (int w = 0; w < wid; ++w) off = acf_add(off, wtot[w]);
    v = acf_add(v, off);
    if (i < m_cnt) S[i + 1] = v;
    __syncthreads();
    if (tid == ACF_THREADS - 1) carry_s = v;         // running total after this row
    __syncthreads();
  }
  for (int i = tid; i < n_here; i += ACF_THREADS) {
    const acf4 hi = S[i + W], lo = S[i];
    const double pr = hi.pr - lo.pr, pi = hi.pi - lo.pi, e1 = hi.e1 - lo.e1, e2 = hi.e2 - lo.e2;
    const double den = sqrt(e1 * e2);                // AutoCorrFunction.m:6
    rho[n0 + i] = mk<T>((T)(pr / den), (T)(pi / den));
  }
}

// ---------------------------------------------------------------------------------------------
// AutoCorrFunction.m:10-24 -- threshold 0.77, indices > WidthWindow, first run of consecutive
// indices; needs a second run to exist (result(2)) or the catch branch gives 65.
// One workgroup scans in chunks with early exit.  out[0]=TgPosition (1-based), out[1]=ok,
// outv[0..1] = rho(TgPosition).
// ---------------------------------------------------------------------------------------------
constexpr int PLAT_THREADS = 1024;

template <typename T, typename Pred>
__device__ int64_t first_index_where(const cx<T>* __restrict__ rho, int64_t from, int64_t n, Pred pred, int64_t* sh) {
  // returns the smallest i in [from, n) with pred(i), or -1; all threads must call
  for (int64_t base = from; base < n; base += PLAT_THREADS) {
    const int64_t i = base + threadIdx.x;
    const bool hit = (i < n) && pred(rho[i]);
    if (threadIdx.x == 0) *sh = INT64_MAX;
    __syncthreads();
    if (hit) atomicMin((unsigned long long*)sh, (unsigned long long)i);
    __syncthreads();
    const int64_t r = *sh;
    __syncthreads();
    if (r != INT64_MAX) return r;
  }
  return -1;
}

template <typename T>
__global__ __launch_bounds__(PLAT_THREADS) void acf_plateau_kernel(const cx<T>* __restrict__ rho, int64_t n, int W,
                                                                   double thr, int64_t* __restrict__ out,
                                                                   double* __restrict__ outv) {
  __shared__ int64_t sh;
  auto above = [thr](cx<T> v) { return sqrt((double)v.x * v.x + (double)v.y * v.y) > thr; };
  auto below = [thr](cx<T> v) { return !(sqrt((double)v.x * v.x + (double)v.y * v.y) > thr); };
  int64_t pos = 65;                                   // AutoCorrFunction.m:23
  int ok = 0;
  // 1-based index idx = i+1 must satisfy idx > W  <=>  i >= W
  const int64_t f = first_index_where<T>(rho, W, n, above, &sh);
  if (f >= 0) {
    const int64_t g = first_index_where<T>(rho, f + 1, n, below, &sh);      // run 1 = [f, g-1]
    if (g >= 0) {
      const int64_t h = first_index_where<T>(rho, g + 1, n, above, &sh);    // a second run exists
      if (h >= 0) {
        pos = ((f + 1) + (g - 1 + 1)) / 2;                                   // :20 floor of the mean of 1-based ends
        ok = 1;
      }
    }
  }
  if (threadIdx.x == 0) {
    out[0] = pos;
    out[1] = ok;
    if (pos >= 1 && pos <= n) { outv[0] = rho[pos - 1].x; outv[1] = rho[pos - 1].y; }
    else { outv[0] = NAN; outv[1] = NAN; }
  }
}

// first spectral bin with |X| > thr (remove_IFO.m:6-8); out[0] = 0-based bin or -1
template <typename T>
__global__ __launch_bounds__(PLAT_THREADS) void first_above_kernel(const cx<T>* __restrict__ spec, int64_t n, double thr,
                                                                   int64_t* __restrict__ out) {
  __shared__ int64_t sh;
  auto above = [thr](cx<T> v) { return sqrt((double)v.x * v.x + (double)v.y * v.y) > thr; };
  const int64_t f = first_index_where<T>(spec, 0, n, above, &sh);
  if (threadIdx.x == 0) out[0] = f;
}

// ---------------------------------------------------------------------------------------------
// fine_sync.m:10-20 -- residual timing estimate.  Pilots of all symbols flattened column-major
// (M = Np*S); taus(i) = angle(q(i+1) conj(q(i)))/(2 pi dk), taus(M)=0; keep where
// |taus(i)-taus(i-1)| < 1e-3; tau = mean of the kept values after dropping the first Np KEPT ones.
// Single workgroup, chunked, with a running rank (ballot prefix count).  angle(0) := 0.
// ---------------------------------------------------------------------------------------------
constexpr int FS_THREADS = 256;

__device__ __forceinline__ double angle0(double re, double im) { return (re == 0.0 && im == 0.0) ? 0.0 : atan2(im, re); }

template <typename T>
struct PilotView {
  const cx<T>* rx;          // [nfft x n_symb]
  const cx<T>* tx;          // [np x n_symb]
  const int32_t* pc0;       // 0-based pilot rows
  int nfft, np;
  int64_t M;
  __device__ void q(int64_t i, double& qr, double& qi) const {     // q = tx * conj(rx)
    const int p = (int)(i % np);
    const int64_t s = i / np;
    const cx<T> r = rx[s * nfft + pc0[p]];
    const cx<T> t = tx[i];
    qr = (double)t.x * r.x + (double)t.y * r.y;
    qi = (double)t.y * r.x - (double)t.x * r.y;
  }
  __device__ double tau_at(int64_t i, double inv2pidk) const {     // taus(i), 0-based, i < M
    if (i >= M - 1) return 0.0;
    double ar, ai, br, bi;
    q(i, ar, ai);
    q(i + 1, br, bi);
    // q(i+1) * conj(q(i))
    return angle0(br * ar + bi * ai, bi * ar - br * ai) * inv2pidk;
  }
};

template <typename T>
__global__ __launch_bounds__(FS_THREADS) void fine_tau_kernel(PilotView<T> pv, double deltak, int variant,
                                                              double* __restrict__ out /* [0]=tau */) {
  __shared__ int wcnt[FS_THREADS / 64];
  __shared__ double wsum[FS_THREADS / 64];
  __shared__ int64_t wn[FS_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double inv = 1.0 / (2.0 * M_PI * deltak);
  int64_t rank_base = 0;       // kept elements before this chunk
  double sum = 0.0;
  int64_t cnt = 0;
  for (int64_t base = 0; base < pv.M; base += FS_THREADS) {
    const int64_t i = base + tid;
    bool keep = false;
    double ti = 0.0;
    if (i >= 1 && i < pv.M) {
      ti = pv.tau_at(i, inv);
      const double d = ti - pv.tau_at(i - 1, inv);
      keep = fabs(d) < 1e-3;                                     // fine_sync.m:18
      if (variant == 1) keep = keep && (d != 0.0);               // T4/fine_sync.m:33
    }
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wid] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < FS_THREADS / 64; ++w) { if (w < wid) woff += wcnt[w]; tot += wcnt[w]; }
    const int64_t rank = rank_base + woff + before;              // 0-based rank among kept
    if (keep && rank >= pv.np) { sum += ti; cnt += 1; }          // :20 taus_result(Np+1:end)
    rank_base += tot;
    __syncthreads();
  }
  for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off, 64); cnt += __shfl_down(cnt, off, 64); }
  if (lane == 0) { wsum[wid] = sum; wn[wid] = cnt; }
  __syncthreads();
  if (tid == 0) {
    double s = 0; int64_t n = 0;
    for (int w = 0; w < FS_THREADS / 64; ++w) { s += wsum[w]; n += wn[w]; }
    out[0] = n > 0 ? s / (double)n : NAN;                        // mean([]) = NaN
  }
}

// fine_sync.m:32-37 -- common phase after the (optional) timing derotation; out[1] = phase_shift
template <typename T>
__global__ __launch_bounds__(FS_THREADS) void fine_phase_kernel(PilotView<T> pv, int time_desync,
                                                                double* __restrict__ out) {
  __shared__ double wsum[FS_THREADS / 64];
  __shared__ int64_t wn[FS_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double tau = out[0];
  double sum = 0.0;
  int64_t cnt = 0;
  for (int64_t i = tid; i < pv.M; i += FS_THREADS) {
    double qr, qi;
    pv.q(i, qr, qi);
    if (time_desync) {
      // rx' = rx * exp(+2 pi j tau k)  =>  q' = q * exp(-2 pi j tau k)   (fine_sync.m:25-27, nn_exp')
      const int k = pv.pc0[(int)(i % pv.np)];
      const double t = tau * (double)k;
      double sn, cs;
      sincospi(2.0 * (t - floor(t)), &sn, &cs);
      const double r2 = qr * cs + qi * sn, i2 = qi * cs - qr * sn;
      qr = r2; qi = i2;
    }
    const double a = angle0(qr, qi);                             // :35
    if (fabs(a) > 1e-3) { sum += a; cnt += 1; }                  // :37
  }
  for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off, 64); cnt += __shfl_down(cnt, off, 64); }
  if (lane == 0) { wsum[wid] = sum; wn[wid] = cnt; }
  __syncthreads();
  if (tid == 0) {
    double s = 0; int64_t n = 0;
    for (int w = 0; w < FS_THREADS / 64; ++w) { s += wsum[w]; n += wn[w]; }
    out[1] = n > 0 ? s / (double)n : NAN;
  }
}

// fine_sync.m:23-29,:39-43 -- apply both corrections in one pass
template <typename T>
__global__ void fine_apply_kernel(const cx<T>* __restrict__ x, cx<T>* __restrict__ y, int nfft, int64_t n_symb,
                                  int time_desync, int freq_desync, const double* __restrict__ est) {
  const double tau = est[0], ph = est[1];
  double psn = 0.0, pcs = 1.0;
  if (freq_desync) sincos(ph, &psn, &pcs);
  const int64_t total = (int64_t)nfft * n_symb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % nfft);
    double cs = 1.0, sn = 0.0;
    if (time_desync) {
      const double t = tau * (double)k;
      sincospi(2.0 * (t - floor(t)), &sn, &cs);
    }
    // total rotation = exp(j(2 pi tau k + ph))
    const double rc = cs * pcs - sn * psn, rs = sn * pcs + cs * psn;
    const cx<T> v = x[i];
    y[i] = mk<T>((T)((double)v.x * rc - (double)v.y * rs), (T)((double)v.x * rs + (double)v.y * rc));
  }
}

}  // namespace ofdm

using namespace ofdm;

extern "C" {

int ofdm_AutoCorrFunction(const void* rx, int64_t len, int width_window, int nfft, void* rho_out,
                          int64_t* tg_position_out, double* freq_offset_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(len >= 0 && width_window >= 1 && nfft >= 1, "AutoCorrFunction: bad sizes");
  OFDM_ARG(width_window <= ACF_MAXW, "AutoCorrFunction: WidthWindow %d exceeds the supported %d", width_window, ACF_MAXW);
  const bool f64 = is_f64(flags);
  const int64_t n_out = len - width_window - nfft;
  Stage st(flags);
  int64_t res[2] = {65, 0};
  double resv[2] = {NAN, NAN};
  if (n_out <= 0) {
    // zeros(1, <=0) -> empty AutoCorr; find() empty -> catch -> 65; AutoCorr(65) errors in MATLAB
    set_error("AutoCorrFunction: signal shorter than WidthWindow+Nfft (index exceeds array bounds at :27)");
    return OFDM_ERR_ARG;
  }
  const void* dx; void *drho, *dres, *dresv;
  OFDM_TRY(st.in(rx, csize(flags) * (size_t)len, &dx));
  if (rho_out) OFDM_TRY(st.out(rho_out, csize(flags) * (size_t)n_out, &drho));
  else OFDM_TRY(st.scratch(csize(flags) * (size_t)n_out, &drho));
  OFDM_TRY(st.fetch(res, sizeof(res), &dres));
  OFDM_TRY(st.fetch(resv, sizeof(resv), &dresv));
  const unsigned grid = cdiv_u(n_out, ACF_TILE);
  if (f64) {
    hipLaunchKernelGGL(acf_kernel<double>, dim3(grid), dim3(ACF_THREADS), 0, ctx().stream, (const c64*)dx, len,
                       width_window, nfft, (c64*)drho, n_out);
    OFDM_TRY(check_launch("acf_kernel"));
    hipLaunchKernelGGL(acf_plateau_kernel<double>, dim3(1), dim3(PLAT_THREADS), 0, ctx().stream, (const c64*)drho,
                       n_out, width_window, 0.77, (int64_t*)dres, (double*)dresv);
  } else {
    hipLaunchKernelGGL(acf_kernel<float>, dim3(grid), dim3(ACF_THREADS), 0, ctx().stream, (const c32*)dx, len,
                       width_window, nfft, (c32*)drho, n_out);
    OFDM_TRY(check_launch("acf_kernel"));
    hipLaunchKernelGGL(acf_plateau_kernel<float>, dim3(1), dim3(PLAT_THREADS), 0, ctx().stream, (const c32*)drho,
                       n_out, width_window, 0.77, (int64_t*)dres, (double*)dresv);
  }
  OFDM_TRY(check_launch("acf_plateau_kernel"));
  OFDM_TRY(st.finish());
  if (tg_position_out) *tg_position_out = res[0];
  if (res[0] > n_out) {
    set_error("AutoCorrFunction: TgPosition %lld exceeds numel(AutoCorr)=%lld (index error at :27)",
              (long long)res[0], (long long)n_out);
    return OFDM_ERR_ARG;
  }
  if (freq_offset_out) *freq_offset_out = -std::atan2(resv[1], resv[0]) / (2.0 * M_PI);   // :27
  return res[1] ? OFDM_OK : OFDM_SOFT_ACF_FALLBACK;
}

int ofdm_remove_IFO(const void* rx, int64_t len, int nfft, void* fixed_out, int* ifo_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(nfft >= 64 && (nfft & (nfft - 1)) == 0 && nfft <= 8192, "remove_IFO: unsupported Nfft %d", nfft);
  OFDM_ARG(len >= 2 * (int64_t)nfft, "remove_IFO: rx_signal(Nfft+1:2*Nfft) exceeds the signal length");
  const bool f64 = is_f64(flags);
  Stage st(flags);
  const void* dx; void *dout, *dspec, *dres;
  OFDM_TRY(st.in(rx, csize(flags) * (size_t)len, &dx));
  OFDM_TRY(st.out(fixed_out, csize(flags) * (size_t)len, &dout));
  OFDM_TRY(st.scratch(csize(flags) * (size_t)nfft, &dspec));
  int64_t first = -1;
  OFDM_TRY(st.scratch(sizeof(int64_t), &dres));
  // fft(rx(Nfft+1:2*Nfft)) : "symbol 0 with a guard of Nfft samples"   (remove_IFO.m:5)
  OFDM_TRY(demod_device(dx, dspec, nfft, 1, nfft, f64));
  if (f64) hipLaunchKernelGGL(first_above_kernel<double>, dim3(1), dim3(PLAT_THREADS), 0, ctx().stream, (const c64*)dspec, (int64_t)nfft, 0.77, (int64_t*)dres);
  else hipLaunchKernelGGL(first_above_kernel<float>, dim3(1), dim3(PLAT_THREADS), 0, ctx().stream, (const c32*)dspec, (int64_t)nfft, 0.77, (int64_t*)dres);
  OFDM_TRY(check_launch("first_above_kernel"));
  OFDM_HIP(hipMemcpyAsync(&first, dres, sizeof(first), hipMemcpyDeviceToHost, ctx().stream));
  OFDM_HIP(hipStreamSynchronize(ctx().stream));
  OFDM_ARG(first >= 0, "remove_IFO: no spectral line above 0.77 (inds(1) index error at :8)");
  if (ifo_out) *ifo_out = (int)first;                                        // inds(1)-1
  OFDM_TRY(cfo_device(dx, dout, len, -(double)first, nfft, f64));            // :9
  return st.finish();
}

int ofdm_fine_sync(const void* rx, int nfft, int64_t n_symb, const int32_t* pilot_carriers, int n_pilots,
                   const void* pilot_values, int time_desync, int freq_desync, int variant, void* out,
                   double* tau_out, double* phase_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(nfft > 0 && n_symb > 0 && n_pilots >= 2, "fine_sync: needs at least two pilot carriers");
  std::vector<int32_t> pc0(n_pilots);
  for (int i = 0; i < n_pilots; ++i) {
    OFDM_ARG(pilot_carriers[i] >= 1 && pilot_carriers[i] <= nfft, "fine_sync: pilot index outside 1..Nfft");
    pc0[i] = pilot_carriers[i] - 1;
  }
  const double deltak = (double)pilot_carriers[1] - (double)pilot_carriers[0];   // fine_sync.m:6
  const bool f64 = is_f64(flags);
  Stage st(flags);
  const void *dx, *dtx, *dpc; void *dout, *dest;
  OFDM_TRY(st.in(rx, csize(flags) * (size_t)nfft * n_symb, &dx));
  OFDM_TRY(st.in(pilot_values, csize(flags) * (size_t)n_pilots * n_symb, &dtx));
  OFDM_TRY(st.upload(pc0.data(), sizeof(int32_t) * n_pilots, &dpc));
  OFDM_TRY(st.out(out, csize(flags) * (size_t)nfft * n_symb, &dout));
  double est[2] = {0, 0};
  OFDM_TRY(st.fetch((tau_out || phase_out) ? est : nullptr, sizeof(est), &dest));
  const int64_t total = (int64_t)nfft * n_symb;
  if (f64) {
    PilotView<double> pv{(const c64*)dx, (const c64*)dtx, (const int32_t*)dpc, nfft, n_pilots, (int64_t)n_pilots * n_symb};
    hipLaunchKernelGGL(fine_tau_kernel<double>, dim3(1), dim3(FS_THREADS), 0, ctx().stream, pv, deltak, variant, (double*)dest);
    hipLaunchKernelGGL(fine_phase_kernel<double>, dim3(1), dim3(FS_THREADS), 0, ctx().stream, pv, time_desync, (double*)dest);
    hipLaunchKernelGGL(fine_apply_kernel<double>, dim3(ew_grid(total)), dim3(256), 0, ctx().stream, (const c64*)dx,
                       (c64*)dout, nfft, n_symb, time_desync, freq_desync, (const double*)dest);
  } else {
    PilotView<float> pv{(const c32*)dx, (const c32*)dtx, (const int32_t*)dpc, nfft, n_pilots, (int64_t)n_pilots * n_symb};
    hipLaunchKernelGGL(fine_tau_kernel<float>, dim3(1), dim3(FS_THREADS), 0, ctx().stream, pv, deltak, variant, (double*)dest);
    hipLaunchKernelGGL(fine_phase_kernel<float>, dim3(1), dim3(FS_THREADS), 0, ctx().stream, pv, time_desync, (double*)dest);
    hipLaunchKernelGGL(fine_apply_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, ctx().stream, (const c32*)dx,
                       (c32*)dout, nfft, n_symb, time_desync, freq_desync, (const double*)dest);
  }
  OFDM_TRY(check_launch("fine_sync kernels"));
  OFDM_TRY(st.finish());
  if (tau_out) *tau_out = est[0];
  if (phase_out) *phase_out = est[1];
  return OFDM_OK;
}

}  // extern "C"
