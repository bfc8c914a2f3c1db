// PAPR study of Task 2: calculatePAPR, calculate_window_PAPR, calculateCCDF.
//
//   calculatePAPR          one pass: max |x|^2 and sum |x|^2 per workgroup, finished on the host
//   calculate_window_PAPR  the reference recomputes max and mean of every Nfft-sample window (O(L Nfft));
//                          here a workgroup owns W = Nfft consecutive window starts, keeps the 2 W samples
//                          they touch in LDS and gets every window in O(1): the maximum from a suffix
//                          maximum of the first block and a prefix maximum of the second (van Herk /
//                          Gil-Werman), the mean from one prefix sum over both.  HBM: each sample read
//                          twice (second time from L2), one double written per window.
//   calculateCCDF          ecdf = sort + run-length encode + running count (hipCUB device primitives)
//
// Powers, sums and outputs are double whatever the input precision (the outputs feed a CCDF plot).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <limits>

#include "ofdm_common.hpp"

namespace ofdm {

__device__ __forceinline__ int lds_pad(int j) { return j + (j >> 5); }        // 1 double per 32: chunked scans stay conflict-free

template <typename T>
__device__ __forceinline__ double power_of(cx<T> v) { return (double)v.x * (double)v.x + (double)v.y * (double)v.y; }

__device__ __forceinline__ double wave_excl_sum(double v, int lane) {
  double inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  return inc - v;
}
__device__ __forceinline__ double wave_excl_max(double v, int lane) {       // identity 0: all values are powers >= 0
  double inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = __shfl_up(inc, d, 64);
    if (lane >= d) inc = fmax(inc, o);
  }
  const double prev = __shfl_up(inc, 1, 64);
  return lane ? prev : 0.0;
}

// In-place inclusive scan (sum or max, identity 0) of the `len` padded LDS entries starting at logical index `off`,
// towards higher indices or (REV) towards lower ones.  Every thread of the workgroup calls it; blockDim.x is a
// multiple of 64 and at most 1024.  tot = blockDim.x doubles of LDS.
template <bool MAX, bool REV>
__device__ void seg_scan(double* __restrict__ a, int off, int len, double* __restrict__ tot) {
  const int nt = blockDim.x, tid = threadIdx.x;
  const int c = (len + nt - 1) / nt;
  const int q0 = min(len, tid * c), q1 = min(len, q0 + c);
  double acc = 0.0;
  for (int q = q0; q < q1; ++q) {
    const int j = lds_pad(REV ? off + len - 1 - q : off + q);
    acc = MAX ? fmax(acc, a[j]) : acc + a[j];
    a[j] = acc;
  }
  tot[tid] = acc;
  __syncthreads();
  if (tid < 64) {                                    // exclusive scan of the per-thread totals by the first wavefront
    const int per = nt >> 6;
    double run = 0.0;
    for (int e = 0; e < per; ++e) {
      const double v = tot[tid * per + e];
      tot[tid * per + e] = run;
      run = MAX ? fmax(run, v) : run + v;
    }
    const double base = MAX ? wave_excl_max(run, tid) : wave_excl_sum(run, tid);
    for (int e = 0; e < per; ++e) {
      const double v = tot[tid * per + e];
      tot[tid * per + e] = MAX ? fmax(v, base) : v + base;
    }
  }
  __syncthreads();
  const double o = tot[tid];
  for (int q = q0; q < q1; ++q) {
    const int j = lds_pad(REV ? off + len - 1 - q : off + q);
    a[j] = MAX ? fmax(a[j], o) : a[j] + o;
  }
  __syncthreads();
}

constexpr int WIN_PER_THREAD = 8;

// Workgroup b: window starts i = b W + r, r < W (calculate_window_PAPR.m:8-14).
template <typename T>
__global__ void window_papr_kernel(const cx<T>* __restrict__ x, int64_t n, int W, int64_t n_out, double* __restrict__ out) {
  extern __shared__ double a[];                      // lds_pad(2 W) powers, then scans of them
  __shared__ double tot[1024];
  const int nt = blockDim.x, tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * W;
  auto load = [&]() {
    for (int j = tid; j < 2 * W; j += nt) {
      const int64_t g = base + j;
      a[lds_pad(j)] = g < n ? power_of(x[g]) : 0.0;
    }
    __syncthreads();
  };
  // ---- window maximum: suffix maxima of [0, W), prefix maxima of [W, 2 W)
  load();
  seg_scan<true, true>(a, 0, W, tot);
  seg_scan<true, false>(a, W, W, tot);
  double mx[WIN_PER_THREAD];
#pragma unroll
  for (int k = 0; k < WIN_PER_THREAD; ++k) {
    const int r = tid + k * nt;
    mx[k] = 0.0;
    if (r < W) mx[k] = r ? fmax(a[lds_pad(r)], a[lds_pad(r + W - 1)]) : a[lds_pad(0)];
  }
  __syncthreads();
  // ---- window mean from one prefix sum over both blocks
  load();
  seg_scan<false, false>(a, 0, 2 * W, tot);
#pragma unroll
  for (int k = 0; k < WIN_PER_THREAD; ++k) {
    const int r = tid + k * nt;
    if (r < W && base + r < n_out) {
      const double s = a[lds_pad(r + W - 1)] - (r ? a[lds_pad(r - 1)] : 0.0);
      const double v = 10.0 * log10(mx[k] / (s / (double)W));          // calculatePAPR.m:4-10
      out[base + r] = v == v ? v : std::numeric_limits<double>::quiet_NaN();
    }
  }
}

// Power-of-two windows (every Nfft of the reference): the scans run in registers.  The first half of the workgroup owns
// the first block mirrored (q <-> sample W-1-q), the second half the second block (q <-> sample W+q), C = 2 W / threads
// consecutive q per thread; a prefix scan over q is then the suffix scan the first block needs and the prefix scan the
// second one needs.  Window r = max / sum of suffix(first block, r) and prefix(second block, r-1): no subtraction of
// large prefix sums.  LDS only carries the coalesced load (chunks are read back conflict-free through lds_pad) and the
// finished scans; TWO_PHASE (W = 8192) reuses one pair of arrays for the maxima and then the sums.
template <typename T, int C, bool TWO_PHASE>
__global__ void window_papr_reg_kernel(const cx<T>* __restrict__ x, int64_t n, int W, int64_t n_out, double* __restrict__ out) {
  extern __shared__ double a[];
  __shared__ double wtot_m[16], wtot_s[16];
  const int nt = blockDim.x, half = nt >> 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = tid >= half, u = tid - blk * half;
  const int wp = W + (W >> 5) + 1;
  double* const M = a;                                // [2][wp]
  double* const S = TWO_PHASE ? a : a + 2 * wp;       // [2][wp]
  const int64_t base = (int64_t)blockIdx.x * W;
  cx<T> xv[C];                                        // all C loads of a thread in flight together
#pragma unroll
  for (int e = 0; e < C; ++e) {
    const int64_t g = base + tid + e * nt;
    xv[e] = g < n ? x[g] : mk<T>(0, 0);
  }
#pragma unroll
  for (int e = 0; e < C; ++e) {
    const int j = tid + e * nt;
    const int b = j >= W, q = b ? j - W : W - 1 - j;
    M[b * wp + lds_pad(q)] = power_of(xv[e]);
  }
  __syncthreads();
  double m[C], s[C];
  double mrun = 0.0, srun = 0.0;
#pragma unroll
  for (int e = 0; e < C; ++e) {
    const double p = M[blk * wp + lds_pad(u * C + e)];
    mrun = fmax(mrun, p);
    srun += p;
    m[e] = mrun;
    s[e] = srun;
  }
  double em = wave_excl_max(mrun, lane), es = wave_excl_sum(srun, lane);
  if (lane == 63) { wtot_m[wave] = fmax(em, mrun); wtot_s[wave] = es + srun; }
  __syncthreads();                                    // also: every chunk has been read, M may be overwritten
  for (int w = blk * (half >> 6); w < wave; ++w) { em = fmax(em, wtot_m[w]); es += wtot_s[w]; }
#pragma unroll
  for (int e = 0; e < C; ++e) {
    M[blk * wp + lds_pad(u * C + e)] = fmax(m[e], em);
    if (!TWO_PHASE) S[blk * wp + lds_pad(u * C + e)] = s[e] + es;
  }
  __syncthreads();
  constexpr int NOUT = C / 2;                         // W / threads
  double mx[NOUT];
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    const int r = tid + k * nt;
    mx[k] = M[lds_pad(W - 1 - r)];
    if (r) mx[k] = fmax(mx[k], M[wp + lds_pad(r - 1)]);
  }
  if (TWO_PHASE) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < C; ++e) S[blk * wp + lds_pad(u * C + e)] = s[e] + es;
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    const int r = tid + k * nt;
    if (base + r < n_out) {
      double sum = S[lds_pad(W - 1 - r)];
      if (r) sum += S[wp + lds_pad(r - 1)];
      const double v = 10.0 * log10(mx[k] / (sum / (double)W));          // calculatePAPR.m:4-10
      out[base + r] = v == v ? v : std::numeric_limits<double>::quiet_NaN();
    }
  }
}

// Short windows: every thread walks its own window (the samples come from L1 / L2).
template <typename T>
__global__ void window_papr_direct_kernel(const cx<T>* __restrict__ x, int W, int64_t n_out, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  double m = 0.0, s = 0.0;
  for (int e = 0; e < W; ++e) {
    const double p = power_of(x[i + e]);
    m = fmax(m, p);
    s += p;
  }
  const double v = 10.0 * log10(m / (s / (double)W));
  out[i] = v == v ? v : std::numeric_limits<double>::quiet_NaN();
}

// part[2 b] = max |x|^2, part[2 b + 1] = sum |x|^2 of the samples workgroup b strides over
template <typename T>
__global__ __launch_bounds__(256) void papr_reduce_kernel(const cx<T>* __restrict__ x, int64_t n, double* __restrict__ part) {
  __shared__ double sm[4], ss[4];
  double m = 0.0, s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double p = power_of(x[i]);
    m = fmax(m, p);
    s += p;
  }
#pragma unroll
  for (int d = 32; d; d >>= 1) {
    m = fmax(m, __shfl_xor(m, d, 64));
    s += __shfl_xor(s, d, 64);
  }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; ss[threadIdx.x >> 6] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
    part[2 * blockIdx.x + 1] = (ss[0] + ss[1]) + (ss[2] + ss[3]);
  }
}

// ecdf preparation: NaN -> one canonical NaN (sorts last), -0 -> +0 (equal values must form one run)
__global__ void ccdf_canon_kernel(const double* __restrict__ v, int64_t n, double* __restrict__ keys, unsigned long long* __restrict__ n_nan) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool isn = false;
  if (i < n) {
    double k = v[i];
    isn = k != k;
    if (isn) k = __longlong_as_double(0x7ff8000000000000LL);
    else if (k == 0.0) k = 0.0;
    keys[i] = k;
  }
  const unsigned long long b = __ballot(isn);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_nan, (unsigned long long)__popcll(b));
}

// x = [u_0, u_0, u_1, ...], CCDF = 1 - [0, cum_0, cum_1, ...] / n     (calculateCCDF.m:4-5 on ecdf's outputs)
__global__ void ccdf_emit_kernel(const double* __restrict__ uniq, const int* __restrict__ cum, const int* __restrict__ n_runs,
                                 double n_valid, double* __restrict__ x_out, double* __restrict__ c_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int runs = *n_runs;
  if (i > runs || runs == 0) return;
  x_out[i] = uniq[i ? i - 1 : 0];
  c_out[i] = 1.0 - (i ? (double)cum[i - 1] / n_valid : 0.0);
}

}  // namespace ofdm

using namespace ofdm;

extern "C" {

int ofdm_calculatePAPR(const void* x, int64_t n, double* papr_db_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(n >= 0 && papr_db_out, "calculatePAPR: bad arguments");
  Stage st(flags);
  const void* dx; void* dpart;
  OFDM_TRY(st.in(x, csize(flags) * (size_t)n, &dx));
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4 * (int64_t)ctx().num_cu));
  std::vector<double> part(2 * (size_t)grid, 0.0);
  OFDM_TRY(st.fetch(part.data(), sizeof(double) * part.size(), &dpart));
  if (is_f64(flags)) hipLaunchKernelGGL(papr_reduce_kernel<double>, dim3(grid), dim3(256), 0, ctx().stream, (const c64*)dx, n, (double*)dpart);
  else hipLaunchKernelGGL(papr_reduce_kernel<float>, dim3(grid), dim3(256), 0, ctx().stream, (const c32*)dx, n, (double*)dpart);
  OFDM_TRY(check_launch("papr_reduce_kernel"));
  OFDM_TRY(st.finish());
  double m = 0.0, s = 0.0;
  for (unsigned b = 0; b < grid; ++b) { m = std::fmax(m, part[2 * b]); s += part[2 * b + 1]; }
  *papr_db_out = 10.0 * std::log10(m / (s / (double)n));                 // calculatePAPR.m:4-10 (n = 0 -> NaN, like mean([]))
  return OFDM_OK;
}

int ofdm_calculate_window_PAPR(const void* x, int64_t n, int nfft, double* paprs_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(n >= 0 && nfft >= 1, "calculate_window_PAPR: bad arguments");
  OFDM_ARG(nfft <= 8192, "calculate_window_PAPR: window of %d samples (limit 8192)", nfft);
  const int64_t n_out = n - nfft + 1;                                     // calculate_window_PAPR.m:4
  if (n_out <= 0) return OFDM_OK;                                         // zeros(1, <=0): empty
  OFDM_ARG(paprs_out, "calculate_window_PAPR: null output");
  Stage st(flags);
  const void* dx; void* dout;
  OFDM_TRY(st.in(x, csize(flags) * (size_t)n, &dx));
  OFDM_TRY(st.out(paprs_out, sizeof(double) * (size_t)n_out, &dout));
  hipStream_t s = ctx().stream;
  if (nfft < 256) {
    const unsigned grid = cdiv_u(n_out, 256);
    if (is_f64(flags)) hipLaunchKernelGGL(window_papr_direct_kernel<double>, dim3(grid), dim3(256), 0, s, (const c64*)dx, nfft, n_out, (double*)dout);
    else hipLaunchKernelGGL(window_papr_direct_kernel<float>, dim3(grid), dim3(256), 0, s, (const c32*)dx, nfft, n_out, (double*)dout);
  } else if ((nfft & (nfft - 1)) == 0 && !getenv("OFDM_PAPR_GENERIC")) {
    const int nt = nfft == 256 ? 128 : nfft <= 1024 ? 256 : nfft == 2048 ? 512 : 1024;
    const int c = 2 * nfft / nt;                                          // 4, 8 or 16 samples per thread
    const bool two = nfft == 8192;
    const size_t dyn = sizeof(double) * (size_t)(nfft + (nfft >> 5) + 1) * (two ? 2 : 4);
    const unsigned grid = cdiv_u(n_out, nfft);
    auto launch = [&](auto kern, auto xp) -> int {
      OFDM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), dyn, s, xp, n, nfft, n_out, (double*)dout);
      return OFDM_OK;
    };
    if (is_f64(flags)) {
      if (c == 4) OFDM_TRY(launch(window_papr_reg_kernel<double, 4, false>, (const c64*)dx));
      else if (c == 8) OFDM_TRY(launch(window_papr_reg_kernel<double, 8, false>, (const c64*)dx));
      else OFDM_TRY(launch(window_papr_reg_kernel<double, 16, true>, (const c64*)dx));
    } else {
      if (c == 4) OFDM_TRY(launch(window_papr_reg_kernel<float, 4, false>, (const c32*)dx));
      else if (c == 8) OFDM_TRY(launch(window_papr_reg_kernel<float, 8, false>, (const c32*)dx));
      else OFDM_TRY(launch(window_papr_reg_kernel<float, 16, true>, (const c32*)dx));
    }
  } else {
    const int nt = nfft <= 2048 ? 256 : 1024;                             // <= WIN_PER_THREAD windows per thread
    const size_t dyn = sizeof(double) * (size_t)(2 * nfft + (2 * nfft >> 5) + 1);
    const unsigned grid = cdiv_u(n_out, nfft);
    if (is_f64(flags)) {
      OFDM_HIP(hipFuncSetAttribute((const void*)window_papr_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
      hipLaunchKernelGGL(window_papr_kernel<double>, dim3(grid), dim3(nt), dyn, s, (const c64*)dx, n, nfft, n_out, (double*)dout);
    } else {
      OFDM_HIP(hipFuncSetAttribute((const void*)window_papr_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
      hipLaunchKernelGGL(window_papr_kernel<float>, dim3(grid), dim3(nt), dyn, s, (const c32*)dx, n, nfft, n_out, (double*)dout);
    }
  }
  OFDM_TRY(check_launch("window_papr_kernel"));
  return st.finish();
}

int ofdm_calculateCCDF(const double* papr_values, int64_t n, double* papr_ccdf_out, double* ccdf_out, int64_t* n_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(n >= 0 && n < (int64_t)1 << 31 && n_out, "calculateCCDF: bad arguments");
  *n_out = 0;
  if (n == 0) return OFDM_OK;
  OFDM_ARG(papr_ccdf_out && ccdf_out, "calculateCCDF: null output");
  Stage st(flags);
  hipStream_t s = ctx().stream;
  const void* dv; void *dx, *dc, *dkeys, *dsorted, *duniq, *dcnt, *dcum, *dmeta;
  OFDM_TRY(st.in(papr_values, sizeof(double) * (size_t)n, &dv));
  OFDM_TRY(st.out(papr_ccdf_out, sizeof(double) * (size_t)(n + 1), &dx));
  OFDM_TRY(st.out(ccdf_out, sizeof(double) * (size_t)(n + 1), &dc));
  OFDM_TRY(st.scratch(sizeof(double) * (size_t)n, &dkeys));
  OFDM_TRY(st.scratch(sizeof(double) * (size_t)n, &dsorted));
  OFDM_TRY(st.scratch(sizeof(double) * (size_t)n, &duniq));
  OFDM_TRY(st.scratch(sizeof(int) * (size_t)n, &dcnt));
  OFDM_TRY(st.scratch(sizeof(int) * (size_t)n, &dcum));
  OFDM_TRY(st.scratch(16, &dmeta));                                       // [0] NaN count (u64), [2] run count (int)
  unsigned long long* d_nan = (unsigned long long*)dmeta;
  int* d_runs = (int*)dmeta + 2;
  OFDM_HIP(hipMemsetAsync(dmeta, 0, 16, s));
  hipLaunchKernelGGL(ccdf_canon_kernel, dim3(cdiv_u(n, 256)), dim3(256), 0, s, (const double*)dv, n, (double*)dkeys, d_nan);
  OFDM_TRY(check_launch("ccdf_canon_kernel"));
  unsigned long long n_nan = 0;
  OFDM_HIP(hipMemcpyAsync(&n_nan, d_nan, sizeof(n_nan), hipMemcpyDeviceToHost, s));
  size_t tb_sort = 0, tb_rle = 0, tb_scan = 0;
  const int ni = (int)n;
  OFDM_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb_sort, (const double*)dkeys, (double*)dsorted, ni, 0, 64, s));
  void* dtmp;
  OFDM_TRY(st.scratch(tb_sort, &dtmp));
  OFDM_HIP(hipcub::DeviceRadixSort::SortKeys(dtmp, tb_sort, (const double*)dkeys, (double*)dsorted, ni, 0, 64, s));
  OFDM_HIP(hipStreamSynchronize(s));                                      // n_nan
  const int n_valid = (int)(n - (int64_t)n_nan);                          // ecdf ignores NaN
  if (n_valid == 0) return st.finish();
  OFDM_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb_rle, (const double*)dsorted, (double*)duniq, (int*)dcnt, d_runs, n_valid, s));
  OFDM_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb_scan, (const int*)dcnt, (int*)dcum, n_valid, s));
  void* dtmp2;
  OFDM_TRY(st.scratch(std::max(tb_rle, tb_scan), &dtmp2));
  OFDM_HIP(hipcub::DeviceRunLengthEncode::Encode(dtmp2, tb_rle, (const double*)dsorted, (double*)duniq, (int*)dcnt, d_runs, n_valid, s));
  int runs = 0;
  OFDM_HIP(hipMemcpyAsync(&runs, d_runs, sizeof(int), hipMemcpyDeviceToHost, s));
  OFDM_HIP(hipStreamSynchronize(s));
  OFDM_HIP(hipcub::DeviceScan::InclusiveSum(dtmp2, tb_scan, (const int*)dcnt, (int*)dcum, runs, s));
  hipLaunchKernelGGL(ccdf_emit_kernel, dim3(cdiv_u((int64_t)runs + 1, 256)), dim3(256), 0, s, (const double*)duniq, (const int*)dcum,
                     (const int*)d_runs, (double)n_valid, (double*)dx, (double*)dc);
  OFDM_TRY(check_launch("ccdf_emit_kernel"));
  *n_out = (int64_t)runs + 1;
  return st.finish();
}

}  // extern "C"
