// Receiver synchronisation: AutoCorrFunction (sliding CP autocorrelation + plateau search),
// remove_IFO, fine_sync.
#include <algorithm>

#include "rx_plan.hpp"
#include "spline_op.hpp"

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
// the three running sums as tile-local exclusive prefix sums in LDS (double accumulation: every thread
// scans its own run of <= 8 consecutive products serially, ONE wave-shuffle scan of the 256 run totals
// ties them together) and takes S[n+W]-S[n]: O(L) work, each sample read twice (once as x[m], once as
// x[m+N]) and rho written once.
// ---------------------------------------------------------------------------------------------
constexpr int ACF_TILE = 1024;
constexpr int ACF_THREADS = 256;
constexpr int ACF_MAXW = 1024;                       // WidthWindow limit (T_guard <= 1024 <-> Nfft <= 8192)
constexpr int ACF_ELEMS = ACF_TILE + ACF_MAXW;       // prefix entries 0..ACF_ELEMS

// accumulation type A: double in parity mode; float in fp32 mode (tile-local prefixes of <= 2048 terms: the window sums
// keep ~1e-6 relative accuracy, the fp32 tolerance of rho is 1e-4) -- 16-byte table entries, 20 KB per workgroup instead of
// 64 KB (seven resident workgroups per CU instead of two) and a float square root / reciprocal per output
template <typename A> struct acf4 { A pr, pi, e1, e2; };

template <typename A> __device__ __forceinline__ acf4<A> acf_add(acf4<A> a, acf4<A> b) { return acf4<A>{a.pr + b.pr, a.pi + b.pi, a.e1 + b.e1, a.e2 + b.e2}; }
template <typename A> __device__ __forceinline__ acf4<A> acf_shfl_up(acf4<A> v, int d) {
  return acf4<A>{__shfl_up(v.pr, d, 64), __shfl_up(v.pi, d, 64), __shfl_up(v.e1, d, 64), __shfl_up(v.e2, d, 64)};
}
template <typename T> inline size_t acf_lds_bytes(int W) { return sizeof(acf4<T>) * (size_t)(ACF_TILE + W + 1); }

// One tile of ACF_TILE outputs starting at n0 of one stream x: emit(i, rho) for i < n_here.  All ACF_THREADS threads call it;
// S = the dynamic LDS table, wtot = ACF_THREADS / 64 entries of static LDS.  Ends with the table still valid (the caller
// synchronises before the next tile overwrites it).
template <typename T, typename Emit>
__device__ __forceinline__ void acf_tile(const cx<T>* __restrict__ x, int64_t n0, int n_here, int W, int nfft,
                                         acf4<T>* __restrict__ S, acf4<T>* __restrict__ wtot, Emit emit) {
  using A = T;
  using acc = acf4<A>;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m_cnt = n_here + W - 1;                  // elements needed: m = n0 .. n0+n_here+W-2
  // (1) every thread forms the products of its own run of E consecutive elements and their inclusive prefix in
  //     registers (a wavefront still reads one contiguous span of x; the lines are reused across the run) ...
  constexpr int EMAX = ACF_ELEMS / ACF_THREADS;                   // 8
  const int E = (m_cnt + ACF_THREADS - 1) / ACF_THREADS;          // <= EMAX
  const int lo_i = tid * E;
  acc pre[EMAX];
  acc run{0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < EMAX; ++j) {
    const int i = lo_i + j;
    if (j < E && i < m_cnt) {
      const int64_t m = n0 + i;                      // m + nfft < len is guaranteed by n_out
      const cx<T> a = x[m], b = x[m + nfft];
      const A ar = a.x, ai = a.y, br = b.x, bi = b.y;
      run = acf_add(run, acc{ar * br + ai * bi,      // a * conj(b)
                             ai * br - ar * bi, ar * ar + ai * ai, br * br + bi * bi});
    }
    pre[j] = run;
  }
  // (2) ... one exclusive scan of the 256 run totals (wave shuffle scan + carry of the earlier wavefronts) ...
  acc v = run;
  for (int d = 1; d < 64; d <<= 1) {
    acc u = acf_shfl_up(v, d);
    if (lane >= d) v = acf_add(v, u);
  }
  if (lane == 63) wtot[wid] = v;
  if (tid == 0) S[0] = acc{0, 0, 0, 0};
  __syncthreads();
  acc off{0, 0, 0, 0};
  for (int w = 0; w < wid; ++w) off = acf_add(off, wtot[w]);
  off = acf_add(off, acc{v.pr - run.pr, v.pi - run.pi, v.e1 - run.e1, v.e2 - run.e2});
  // (3) ... and the prefix table is written once: S[i+1] = sum_{m <= i}
#pragma unroll
  for (int j = 0; j < EMAX; ++j) {
    const int i = lo_i + j;
    if (j < E && i < m_cnt) S[i + 1] = acf_add(pre[j], off);
  }
  __syncthreads();
  for (int i = tid; i < n_here; i += ACF_THREADS) {
    const acc hi = S[i + W], lo = S[i];
    const A pr = hi.pr - lo.pr, pi = hi.pi - lo.pi, e1 = hi.e1 - lo.e1, e2 = hi.e2 - lo.e2;
    const A den = sqrt(e1 * e2);                     // AutoCorrFunction.m:6
    if constexpr (sizeof(A) == 4) {
      const A inv = A(1) / den;
      emit(i, mk<T>(pr * inv, pi * inv));
    } else {
      emit(i, mk<T>((T)(pr / den), (T)(pi / den)));
    }
  }
}

template <typename T>
__global__ __launch_bounds__(ACF_THREADS) void acf_kernel(const cx<T>* __restrict__ x, int64_t len, int W, int nfft,
                                                          cx<T>* __restrict__ rho, int64_t n_out, int64_t rho_stride = 0) {
  // batched callers: grid.y = frame (one stream per frame); rho rows rho_stride apart (0: n_out)
  using acc = acf4<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char acf_smem[];
  acc* const S = (acc*)acf_smem;                     // [ACF_TILE + W + 1] exclusive prefix: S[i] = sum_{m<i}
  __shared__ acc wtot[ACF_THREADS / 64];
  const int64_t fr = blockIdx.y;
  x += fr * len;
  rho += fr * (rho_stride ? rho_stride : n_out);
  const int64_t n0 = (int64_t)blockIdx.x * ACF_TILE;
  const int n_here = (int)((n_out - n0 < ACF_TILE) ? (n_out - n0) : ACF_TILE);
  cx<T>* const dst = rho + n0;
  acf_tile<T>(x, n0, n_here, W, nfft, S, wtot, [dst](int i, cx<T> r) { dst[i] = r; });
}

// ---------------------------------------------------------------------------------------------
// Batched receiver (ofdm_rx_chain_task4): AutoCorrFunction.m:3-24 for one frame per workgroup WITHOUT the curve -- the
// receiver needs TgPosition and AutoCorr(TgPosition) only (T4/Main_model_Task_4.m:278-303).  The tiles of acf_tile are
// walked in order (so every rho value is bit-identical to acf_kernel's), each tile's rho stays in LDS, and the search of
// :10-20 -- first index > WidthWindow above the threshold (f), first one below after it (g), first one above after that
// (h: a second run exists) -- advances through the tile as a three-state machine; the walk stops at h (a symbol and a
// half into the frame on average) or at the end of the stream (the catch branch, TgPosition = 65).  Replaces five
// launches (rho of a three-symbol prefix -> plateau search -> list of unresolved frames -> rho of those over the whole
// stream -> search again -> per-frame scalars) and the 0.5 GB round trip of the prefix curve.
//   tg[f] = TgPosition, fo[f] = -angle(AutoCorr(TgPosition)) / (2 pi)  (:27), status[f] = 0 / 1 (catch branch) / -2 (index error at :27)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(ACF_THREADS) void t4_acf_search_kernel(const cx<T>* __restrict__ x_all, int64_t len, int W, int nfft,
                                                                    int64_t n_out, double thr, int64_t* __restrict__ tg,
                                                                    double* __restrict__ fo, int32_t* __restrict__ status) {
  using acc = acf4<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char acf_smem[];
  acc* const S = (acc*)acf_smem;
  cx<T>* const rt = (cx<T>*)(S + ACF_TILE + W + 1);  // rho of the current tile
  __shared__ acc wtot[ACF_THREADS / 64];
  __shared__ int wmin[ACF_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t fr = blockIdx.x;
  const cx<T>* const x = x_all + fr * len;
  auto tile = [&](int64_t n0) {
    const int n_here = (int)((n_out - n0 < ACF_TILE) ? (n_out - n0) : ACF_TILE);
    acf_tile<T>(x, n0, n_here, W, nfft, S, wtot, [rt](int i, cx<T> r) { rt[i] = r; });
    __syncthreads();
    return n_here;
  };
  // the smallest i in [lo, n_here) of the tile in LDS whose |rho| is above (want_above) / not above the threshold, or -1
  auto first_in_tile = [&](int lo, int n_here, bool want_above) -> int {
    int best = 0x7fffffff;
    for (int i = lo + tid; i < n_here; i += ACF_THREADS) {        // ascending per thread: its first hit is its smallest
      const cx<T> v = rt[i];
      const bool ab = sqrt((double)v.x * v.x + (double)v.y * v.y) > thr;
      if (ab == want_above) { best = i; break; }
    }
    for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
    if (lane == 0) wmin[wid] = best;
    __syncthreads();
    int r = wmin[0];
    for (int w = 1; w < ACF_THREADS / 64; ++w) r = min(r, wmin[w]);
    __syncthreads();
    return r == 0x7fffffff ? -1 : r;
  };
  int stage = 0;                                      // 0: looking for f, 1: g, 2: h, 3: found
  int64_t from = W, f0 = -1, g0 = -1;                 // 1-based index idx = i + 1 must be > W  <=>  i >= W
  int64_t cur_n0 = -1;
  for (int64_t n0 = (from / ACF_TILE) * ACF_TILE; n0 < n_out && stage < 3; n0 += ACF_TILE) {
    const int n_here = tile(n0);
    cur_n0 = n0;
    while (stage < 3) {
      const int lo = from > n0 ? (int)(from - n0) : 0;
      const int r = lo < n_here ? first_in_tile(lo, n_here, stage != 1) : -1;
      if (r < 0) break;
      if (stage == 0) f0 = n0 + r; else if (stage == 1) g0 = n0 + r;
      from = n0 + r + 1;
      ++stage;
    }
  }
  const int ok = stage == 3;
  const int64_t pos = ok ? ((f0 + 1) + (g0 - 1 + 1)) / 2 : 65;   // :20 floor of the mean of the run's 1-based ends; :23
  double vr = NAN, vi = NAN;
  if (pos >= 1 && pos <= n_out) {
    const int64_t pn0 = ((pos - 1) / ACF_TILE) * ACF_TILE;
    if (pn0 != cur_n0) tile(pn0);                    // the plateau's tile is no longer (or was never) the one in LDS
    const cx<T> v = rt[pos - 1 - pn0];
    vr = (double)v.x; vi = (double)v.y;
  }
  if (tid == 0) {
    tg[fr] = pos;
    fo[fr] = -atan2(vi, vr) / (2.0 * M_PI);
    status[fr] = pos > n_out ? -2 : (ok ? 0 : 1);
  }
}

template <typename T> inline size_t acf_search_lds_bytes(int W) { return acf_lds_bytes<T>(W) + sizeof(cx<T>) * ACF_TILE; }

// dynamic LDS beyond 64 KB (double table at WidthWindow = 1024) has to be allowed for the kernel (per device: set on every call)
template <typename T>
static int acf_prepare() {
  OFDM_HIP(hipFuncSetAttribute((const void*)acf_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acf_lds_bytes<T>(ACF_MAXW)));
  OFDM_HIP(hipFuncSetAttribute((const void*)t4_acf_search_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)acf_search_lds_bytes<T>(ACF_MAXW)));
  return OFDM_OK;
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
                                                                   double* __restrict__ outv, int64_t rho_stride = 0) {
  rho += (int64_t)blockIdx.x * (rho_stride ? rho_stride : n);   // grid.x = frame
  out += 2 * blockIdx.x;
  outv += 2 * blockIdx.x;
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
  spec += (int64_t)blockIdx.x * n;                   // grid.x = frame
  out += blockIdx.x;
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
// angle of a pilot product: parity mode in double; throughput mode (fp32 data) with the float atan2 -- the double one is
// ~150 double-rate instructions and made the two fine-sync reductions compute-bound once their gathers were gone
template <typename T>
__device__ __forceinline__ double angle0_t(double re, double im) {
  if constexpr (sizeof(T) == 4) return (re == 0.0 && im == 0.0) ? 0.0 : (double)atan2f((float)im, (float)re);
  else return angle0(re, im);
}

template <typename T>
struct PilotView {
  const cx<T>* rx;          // [nfft x n_symb]
  const cx<T>* tx;          // [np x n_symb]
  const int32_t* pc0;       // 0-based pilot rows
  int nfft, np;
  int64_t M;                // np * n_symb, < 2^31 (checked on the host)
  int64_t rx_fstride;       // elements between the frames of a batch (0 for a single frame)
  int compact;              // rx is already X(pilotCarriers, :) = [np x n_symb] (the batched receiver: written by its demodulator)
  __device__ void q(unsigned i, double& qr, double& qi) const {    // q = tx * conj(rx)
    const unsigned s = i / (unsigned)np, p = i - s * (unsigned)np;
    const cx<T> r = compact ? rx[i] : rx[(size_t)s * nfft + pc0[p]];
    const cx<T> t = tx[i];
    qr = (double)t.x * r.x + (double)t.y * r.y;
    qi = (double)t.y * r.x - (double)t.x * r.y;
  }
};

// taus(i) = angle(q(i+1) conj(q(i))) / (2 pi dk)
template <typename T>
__device__ __forceinline__ double tau_of(double ar, double ai, double br, double bi, double inv2pidk) {
  return angle0_t<T>(br * ar + bi * ai, bi * ar - br * ai) * inv2pidk;
}

// Every thread owns FS_U CONSECUTIVE entries of `taus` per trip: their FS_U + 2 pilot products are requested together
// (the kernel is bound by the latency of these gathers, not by arithmetic: one workgroup per frame), each product and
// each angle is formed once, and a frame of 6700 pilots takes 7 trips instead of 27.
constexpr int FS_U = 4;

// scratch of the two reductions (static LDS of the calling kernel)
struct FsScratch {
  int wcnt[FS_THREADS / 64];
  double wsum[FS_THREADS / 64];
  int64_t wn[FS_THREADS / 64];
  double result;
};

// the residual timing estimate tau of one frame, returned to every thread of the FS_THREADS-thread workgroup
template <typename T>
__device__ double fine_tau_body(const PilotView<T>& pv, double deltak, int variant, FsScratch& sc) {
  int (&wcnt)[FS_THREADS / 64] = sc.wcnt;
  double (&wsum)[FS_THREADS / 64] = sc.wsum;
  int64_t (&wn)[FS_THREADS / 64] = sc.wn;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double inv = 1.0 / (2.0 * M_PI * deltak);
  const unsigned M = (unsigned)pv.M;
  int64_t rank_base = 0;       // kept elements before this trip
  double sum = 0.0;
  int64_t cnt = 0;
  // length of `taus`: T5/fine_sync.m:8 allocates numel(pilotValues) entries (the loop leaves the last one 0);
  // T4/fine_sync.m:8 allocates Np and lets the loop of :25-30 grow it to numel-1 entries -- no trailing 0
  const unsigned L = (variant == 1 && pv.M - 1 >= pv.np) ? M - 1 : M;
  for (unsigned base = 0; base < L; base += FS_THREADS * FS_U) {
    const unsigned i0 = base + FS_U * tid;                       // entries i0 .. i0 + FS_U - 1
    // products q(i0 - 1) .. q(i0 + FS_U): taus(i) needs q(i), q(i + 1); the first difference needs taus(i0 - 1)
    double qr[FS_U + 2], qi[FS_U + 2];
#pragma unroll
    for (int u = 0; u < FS_U + 2; ++u) {
      const long long i = (long long)i0 + u - 1;
      qr[u] = qi[u] = 0.0;
      if (i >= 0 && i < (long long)M && i0 < L) pv.q((unsigned)i, qr[u], qi[u]);
    }
    double tp = 0.0;                                             // taus(i0 - 1)
    if (i0 >= 1 && i0 - 1 < M - 1) tp = tau_of<T>(qr[0], qi[0], qr[1], qi[1], inv);
    bool keep[FS_U];
    double tv[FS_U];
    int mine = 0;
#pragma unroll
    for (int u = 0; u < FS_U; ++u) {
      const unsigned i = i0 + u;
      keep[u] = false;
      tv[u] = 0.0;
      if (i < L) {
        tv[u] = i < M - 1 ? tau_of<T>(qr[u + 1], qi[u + 1], qr[u + 2], qi[u + 2], inv) : 0.0;      // taus(M) = 0 (T5 form)
        if (i >= 1) {
          const double d = tv[u] - tp;
          keep[u] = fabs(d) < 1e-3;                              // fine_sync.m:18
          if (variant == 1) keep[u] = keep[u] && (d != 0.0);     // T4/fine_sync.m:33
        }
      }
      tp = tv[u];
      mine += keep[u] ? 1 : 0;
    }
    // rank among the kept entries, in entry order (thread-major, then u)
    int before = 0, wtot = 0;
#pragma unroll
    for (int u = 0; u < FS_U; ++u) {
      const unsigned long long bal = __ballot(keep[u]);
      before += __popcll(bal & ((1ull << lane) - 1ull));
      wtot += __popcll(bal);
    }
    if (lane == 0) wcnt[wid] = wtot;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < FS_THREADS / 64; ++w) { if (w < wid) woff += wcnt[w]; tot += wcnt[w]; }
    int64_t rank = rank_base + woff + before;                    // 0-based rank of this thread's first kept entry
#pragma unroll
    for (int u = 0; u < FS_U; ++u)
      if (keep[u]) {
        if (rank >= pv.np) { sum += tv[u]; cnt += 1; }           // :20 taus_result(Np+1:end)
        rank += 1;
      }
    rank_base += tot;
    __syncthreads();
  }
  for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off, 64); cnt += __shfl_down(cnt, off, 64); }
  if (lane == 0) { wsum[wid] = sum; wn[wid] = cnt; }
  __syncthreads();
  if (tid == 0) {
    double s = 0; int64_t n = 0;
    for (int w = 0; w < FS_THREADS / 64; ++w) { s += wsum[w]; n += wn[w]; }
    sc.result = n > 0 ? s / (double)n : NAN;                     // mean([]) = NaN
  }
  __syncthreads();
  const double r = sc.result;
  __syncthreads();
  return r;
}

template <typename T>
__global__ __launch_bounds__(FS_THREADS) void fine_tau_kernel(PilotView<T> pv, double deltak, int variant,
                                                              double* __restrict__ out /* [0]=tau */) {
  pv.rx += (int64_t)blockIdx.x * pv.rx_fstride;      // grid.x = frame
  __shared__ FsScratch sc;
  const double tau = fine_tau_body<T>(pv, deltak, variant, sc);
  if (threadIdx.x == 0) out[2 * blockIdx.x] = tau;
}

// fine_sync.m:32-37 -- common phase after the (optional) timing derotation; out[1] = phase_shift
// rotation exp(-2 pi j tau k) of pilot row k as (cos, sin): the arithmetic of fine_phase_body's per-entry form
template <typename T>
__device__ __forceinline__ void fine_rot(double tau, int k, double& cs, double& sn) {
  const double t = tau * (double)k;
  if constexpr (sizeof(T) == 4) {                                  // throughput mode: phase reduced in double, sine / cosine in float
    float fs, fc;
    sincospif(2.0f * (float)(t - floor(t)), &fs, &fc);
    sn = fs; cs = fc;
  } else {
    sincospi(2.0 * (t - floor(t)), &sn, &cs);
  }
}

// rot_tab (optional, [np] (cos, sin) of fine_rot per pilot row): the rotation depends on the pilot row only, a caller that
// holds a whole frame computes it once per row instead of once per (row, symbol) -- same values
template <typename T>
__device__ double fine_phase_body(const PilotView<T>& pv, int time_desync, double tau, FsScratch& sc,
                                  const double2* rot_tab = nullptr) {
  double (&wsum)[FS_THREADS / 64] = sc.wsum;
  int64_t (&wn)[FS_THREADS / 64] = sc.wn;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned M = (unsigned)pv.M;
  // i mod np by a multiply (exact while i * np < 2^32: frames of up to 65535 pilot samples), else the division
  const unsigned npu = (unsigned)pv.np, magic = (M < 65536u && npu < 65536u) ? 0xFFFFFFFFu / npu + 1u : 0u;
  auto mod_np = [&](unsigned i) { return magic ? i - __umulhi(i, magic) * npu : i % npu; };
  double sum = 0.0;
  int64_t cnt = 0;
  for (unsigned base = 0; base < M; base += FS_THREADS * FS_U) {
    double qr[FS_U], qi[FS_U];
#pragma unroll
    for (int u = 0; u < FS_U; ++u) {                              // FS_U gathers in flight per thread
      const unsigned i = base + u * FS_THREADS + tid;
      qr[u] = qi[u] = 0.0;
      if (i < M) pv.q(i, qr[u], qi[u]);
    }
#pragma unroll
    for (int u = 0; u < FS_U; ++u) {
      const unsigned i = base + u * FS_THREADS + tid;
      if (i < M) {
        if (time_desync) {
          // rx' = rx * exp(+2 pi j tau k)  =>  q' = q * exp(-2 pi j tau k)   (fine_sync.m:25-27, nn_exp')
          double sn, cs;
          if (rot_tab) { const double2 r = rot_tab[mod_np(i)]; cs = r.x; sn = r.y; }
          else fine_rot<T>(tau, pv.pc0[mod_np(i)], cs, sn);
          const double r2 = qr[u] * cs + qi[u] * sn, i2 = qi[u] * cs - qr[u] * sn;
          qr[u] = r2; qi[u] = i2;
        }
        const double a = angle0_t<T>(qr[u], qi[u]);              // :35
        if (fabs(a) > 1e-3) { sum += a; cnt += 1; }              // :37
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off, 64); cnt += __shfl_down(cnt, off, 64); }
  if (lane == 0) { wsum[wid] = sum; wn[wid] = cnt; }
  __syncthreads();
  if (tid == 0) {
    double s = 0; int64_t n = 0;
    for (int w = 0; w < FS_THREADS / 64; ++w) { s += wsum[w]; n += wn[w]; }
    sc.result = n > 0 ? s / (double)n : NAN;
  }
  __syncthreads();
  const double r = sc.result;
  __syncthreads();
  return r;
}

template <typename T>
__global__ __launch_bounds__(FS_THREADS) void fine_phase_kernel(PilotView<T> pv, int time_desync,
                                                                double* __restrict__ out) {
  pv.rx += (int64_t)blockIdx.x * pv.rx_fstride;      // grid.x = frame
  __shared__ FsScratch sc;
  const double ph = fine_phase_body<T>(pv, time_desync, out[2 * blockIdx.x], sc);
  if (threadIdx.x == 0) out[2 * blockIdx.x + 1] = ph;
}

// fine_sync.m:23-29,:39-43 -- apply both corrections in one pass
template <typename T>
__global__ void fine_apply_kernel(const cx<T>* __restrict__ x, cx<T>* __restrict__ y, int nfft, int64_t n_symb,
                                  int time_desync, int freq_desync, const double* __restrict__ est) {
  x += (int64_t)blockIdx.y * nfft * n_symb;          // grid.y = frame
  y += (int64_t)blockIdx.y * nfft * n_symb;
  est += 2 * blockIdx.y;
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


// ---------------------------------------------------------------------------------------------
// Task-4 receiver for a batch of frames (ofdm_rx_chain_task4): every stage above with grid.x / grid.y = frame, the
// per-frame scalars (TgPosition, FreqOffset, IFO, fine-sync estimates) never leave the device.
// ---------------------------------------------------------------------------------------------
// plateau result -> TgPosition, FreqOffset = -angle(rho(TgPosition)) / (2 pi) (AutoCorrFunction.m:27), status
__global__ void t4_scalars_kernel(const int64_t* __restrict__ res, const double* __restrict__ resv, int64_t n_out,
                                  int64_t* __restrict__ tg, double* __restrict__ fo, int32_t* __restrict__ status,
                                  int64_t n_frames) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  const int64_t pos = res[2 * f];
  tg[f] = pos;
  fo[f] = -atan2(resv[2 * f + 1], resv[2 * f]) / (2.0 * M_PI);
  status[f] = pos > n_out ? -2 : (res[2 * f + 1] ? 0 : 1);     // -2: index error at :27, 1: catch branch (65)
}

// add_STO(rx, TgPosition), add_STO(., -(Nfft+Tg)) (T4/Main_model_Task_4.m:292-294) and add_CFO(., -FreqOffset) (:301) in
// one pass: the shifts are copies, the rotation is the arithmetic of cfo_kernel.
template <typename T>
__global__ void t4_align_kernel(const cx<T>* __restrict__ rx, cx<T>* __restrict__ y, int64_t len, int sym_len, int nfft,
                                int time_desync, int freq_desync, const int64_t* __restrict__ tg,
                                const double* __restrict__ fo) {
  const int64_t f = blockIdx.y;
  const cx<T>* x = rx + f * len;
  cx<T>* o = y + f * len;
  const int64_t pos = time_desync ? tg[f] : 0;
  const double cfo = freq_desync ? -fo[f] : 0.0;
  const double inv_nfft = 1.0 / (double)nfft;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    cx<T> v = mk<T>(0, 0);
    if (time_desync) {
      const int64_t i1 = i - sym_len;                              // index in add_STO(rx, pos)
      if (i1 >= 0 && i1 < len - pos && i1 + pos < len) v = x[i1 + pos];
    } else {
      v = x[i];
    }
    if (freq_desync) {
      const double t = cfo * (double)i * inv_nfft;
      const double fr = t - floor(t);
      double sn, cs;
      sincospi(2.0 * fr, &sn, &cs);
      v = mk<T>((T)((double)v.x * cs - (double)v.y * sn), (T)((double)v.x * sn + (double)v.y * cs));
    }
    o[i] = v;
  }
}

// rx_signal(Nfft+1 : 2*Nfft) of every frame, contiguous (remove_IFO.m:5)
template <typename T>
__global__ void t4_segment_kernel(const cx<T>* __restrict__ y, cx<T>* __restrict__ seg, int64_t len, int nfft) {
  const int64_t f = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nfft; k += gridDim.x * blockDim.x)
    seg[f * nfft + k] = y[f * len + nfft + k];
}

// add_CFO(., -IFO) in place (remove_IFO.m:9); frames without a line above 0.77 are marked (MATLAB would abort)
template <typename T>
__global__ void t4_ifo_kernel(cx<T>* __restrict__ y, int64_t len, int nfft, const int64_t* __restrict__ first,
                              int32_t* __restrict__ ifo_out, int32_t* __restrict__ status) {
  const int64_t f = blockIdx.y;
  const int64_t fi = first[f];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ifo_out[f] = (int32_t)fi;
    if (fi < 0 && status[f] >= 0) status[f] = -1;
  }
  if (fi <= 0) return;                                             // nothing to rotate
  const double cfo = -(double)fi, inv_nfft = 1.0 / (double)nfft;
  cx<T>* o = y + f * len;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    const double t = cfo * (double)i * inv_nfft;
    const double fr = t - floor(t);
    double sn, cs;
    sincospi(2.0 * fr, &sn, &cs);
    const cx<T> v = o[i];
    o[i] = mk<T>((T)((double)v.x * cs - (double)v.y * sn), (T)((double)v.x * sn + (double)v.y * cs));
  }
}

// ---- the same three stages without the intermediate streams (Nfft 512..4096): sample i of the aligned, CFO- and
// IFO-corrected stream of a frame, computed from rx on the fly with the roundings of the separate stages
// (round to T after the add_CFO(-FreqOffset) rotation and again after the add_CFO(-IFO) rotation).
template <typename T>
__device__ __forceinline__ cx<T> t4_raw(const cx<T>* __restrict__ x, int64_t i, int64_t len, int sym_len, int time_desync,
                                        int64_t pos) {
  if (!time_desync) return nt_load(x + i);
  const int64_t i1 = i - sym_len;
  return (i1 >= 0 && i1 < len - pos && i1 + pos < len) ? nt_load(x + i1 + pos) : mk<T>(0, 0);
}
template <typename T>
__device__ __forceinline__ cx<T> t4_rotate(cx<T> v, double cfo, int64_t i, double inv_nfft) {
  const double t = cfo * (double)i * inv_nfft;
  double sn, cs;
  sincospi(2.0 * (t - floor(t)), &sn, &cs);
  return mk<T>((T)((double)v.x * cs - (double)v.y * sn), (T)((double)v.x * sn + (double)v.y * cs));
}

// rx_signal(Nfft+1 : 2*Nfft) after the STO fix and add_CFO(-FreqOffset), straight from rx (remove_IFO.m:5)
template <typename T>
__global__ void t4_segment_direct_kernel(const cx<T>* __restrict__ rx, cx<T>* __restrict__ seg, int64_t len, int sym_len, int nfft,
                                         int time_desync, const int64_t* __restrict__ tg, const double* __restrict__ fo) {
  const int64_t f = blockIdx.y;
  const int64_t pos = time_desync ? tg[f] : 0;
  const double cfo = -fo[f], inv = 1.0 / (double)nfft;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nfft; k += gridDim.x * blockDim.x) {
    const int64_t i = (int64_t)nfft + k;
    seg[f * nfft + k] = t4_rotate<T>(t4_raw<T>(rx + f * len, i, len, sym_len, time_desync, pos), cfo, i, inv);
  }
}

__global__ void t4_ifo_finalize_kernel(const int64_t* __restrict__ first, int32_t* __restrict__ ifo_out,
                                       int32_t* __restrict__ status, int64_t n_frames) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  ifo_out[f] = (int32_t)first[f];
  if (first[f] < 0 && status[f] >= 0) status[f] = -1;
}

// OFDM_demodulator of every symbol of every frame on the wave-local transform, its samples taken from rx through
// t4_raw / t4_rotate: no aligned / corrected copy of the batch is ever written.
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void t4_demod_kernel(const cx<T>* __restrict__ rx, cx<T>* __restrict__ X,
                                                           const cx<T>* __restrict__ tw, int64_t len, int t_guard, int n_symb,
                                                           int64_t n_frames, int time_desync, int freq_desync,
                                                           const int64_t* __restrict__ tg, const double* __restrict__ fo,
                                                           const int32_t* __restrict__ ifo, int n_keep,
                                                           cx<T>* __restrict__ xp, const int16_t* __restrict__ prole, int np) {
  constexpr int N = 512 * NW;
  constexpr int BPT = NW > 1 ? 8 / NW : 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* lwv = (cx<T>*)smem;
  cx<T>* const ex = lwv;
  cx<T>* twl = lwv + NW * WAVE_LDS_ELEMS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gid = threadIdx.x;
  DifTw<T, NW> dt;
  wave_tw_fill<T, NW>(twl, tw);
  dif_tw_init<T, NW>(dt, gid, tw);
  cx<T> twb[7];
#pragma unroll
  for (int t = 1; t < 8; ++t) twb[t - 1] = tw[(t * (lane & 7) * 8) * NW];
  __syncthreads();
  const int sym_len = N + t_guard;
  const double inv = 1.0 / (double)N;
  const int64_t total = n_frames * n_symb;
  // slot e of this thread <-> sample m_e of the symbol (the layout frame_load uses)
  auto slot_m = [&](int e) -> int {
    if constexpr (NW == 1) return lane + 64 * e;
    else return gid * BPT + (e / NW) + 512 * (e % NW);
  };
  auto load_raw = [&](cx<T> (&dst)[8], int64_t sg) {
    const int64_t f = sg / n_symb;
    const int sy = (int)(sg - f * n_symb);
    const int64_t pos = time_desync ? tg[f] : 0;
    const cx<T>* xf = rx + f * len;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e] = t4_raw<T>(xf, (int64_t)sy * sym_len + t_guard + slot_m(e), len, sym_len, time_desync, pos);
  };
  // phase (in turns) of the merged rotation add_CFO(-FreqOffset) . add_CFO(-IFO) at stream index i: the fractional part in
  // double, the integer-offset part reduced exactly in integers (N is a power of two)
  auto turns_at = [&](double c1, int32_t fi, int64_t i) -> double {
    double t = c1 * (double)i * inv;
    t -= floor(t);
    if (fi > 0) {
      t -= (double)(((int64_t)fi * i) & (N - 1)) * inv;
      t += t < 0.0 ? 1.0 : 0.0;
    }
    return t;
  };
  cx<T> v[8], nx[8];
  if ((int64_t)blockIdx.x < total) load_raw(nx, blockIdx.x);
  for (int64_t sg = blockIdx.x; sg < total; sg += gridDim.x) {
    const int64_t f = sg / n_symb;
    const int sy = (int)(sg - f * n_symb);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = nx[e];
    if (sg + gridDim.x < total) load_raw(nx, sg + gridDim.x);
    if (freq_desync) {
      const double c1 = -fo[f];
      const int32_t fi = ifo[f];
      if constexpr (std::is_same<T, float>::value) {
        // throughput mode: ONE rotation per sample.  The rotor of the thread's first sample comes from a sine / cosine of the
        // exactly reduced phase; its other samples sit at fixed distances (e / NW) + 512 (e % NW), whose rotors are
        // frame constants (wave-uniform: a handful of scalar sine / cosine pairs per symbol): rotor_e = rotor_0 * step_e.
        // (Each sample used to pay two double multiplies, a floor, a 64-bit product and a sincospif of its own.)
        const int64_t i0 = (int64_t)sy * sym_len + t_guard + slot_m(0);
        auto rotor = [&](int64_t i) {
          float sn, cs;
          sincospif(2.0f * (float)turns_at(c1, fi, i), &sn, &cs);
          return mk<float>(cs, sn);
        };
        const cx<float> r0 = rotor(i0);
        // distances (e / NW) + 512 (e % NW) [or 64 e for one wavefront per symbol]: powers of two base steps
        constexpr int SA = NW > 1 ? 512 : 64, NA = NW > 1 ? NW : 8, NB = NW > 1 ? BPT : 1;
        cx<float> pa[NA];                                         // rotor(SA a)
        pa[0] = mk<float>(1.0f, 0.0f);
        if (NA > 1) pa[1] = rotor(SA);
#pragma unroll
        for (int a = 2; a < NA; ++a) pa[a] = pa[a - 1] * pa[1];
        cx<float> pb = r0;                                        // r0 * rotor(b)
        const cx<float> r1 = NB > 1 ? rotor(1) : mk<float>(1.0f, 0.0f);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            const int e = NW > 1 ? b * NW + a : a;
            v[e] = v[e] * (a == 0 ? pb : pb * pa[a]);
          }
          pb = pb * r1;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int64_t i = (int64_t)sy * sym_len + t_guard + slot_m(e);
          v[e] = t4_rotate<T>(v[e], c1, i, inv);
          if (fi > 0) v[e] = t4_rotate<T>(v[e], -(double)fi, i, inv);
        }
      }
    }
    if constexpr (NW > 1) {
      dif_stage<T, NW>(v, dt);
      __syncthreads();
      dif_scatter<T, NW>(v, gid, ex);
      __syncthreads();
      dif_gather<T>(v, wave, lane, ex);
    } else {
      __syncthreads();
    }
    wave_fft512<T, false>(v, lane, twb, twl, lwv + wave * WAVE_LDS_ELEMS);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) lwv[NW * (lane + 64 * t) + wave] = v[t];
    __syncthreads();
    // rows 1..n_keep of the column (everything downstream reads carriers 1..N_carrier only) + its pilot rows, compact
    cx<T>* dst = X + sg * (int64_t)n_keep;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = gid + 64 * NW * e;
      if (i < n_keep) {
        const cx<T> xv = lwv[i];
        nt_store(dst + i, xv);
        if (xp) {
          const int pr = prole[i];
          if (pr >= 0) xp[sg * (int64_t)np + pr] = xv;
        }
      }
    }
  }
}

template <typename T, int NW>
static int t4_demod_launch(const void* rx, void* X, const void* tw, int64_t len, int t_guard, int n_symb, int64_t F, int td, int fd,
                           const int64_t* tg, const double* fo, const int32_t* ifo, int n_keep, void* xp, const void* prole, int np) {
  const size_t dyn = sizeof(cx<T>) * ((size_t)NW * WAVE_LDS_ELEMS + WAVE_TW_ELEMS);
  auto kern = t4_demod_kernel<T, NW>;
  const int per_cu = resident_blocks_per_cu((const void*)kern, 64 * NW, dyn);
  const unsigned grid = (unsigned)std::min<int64_t>(F * n_symb, (int64_t)ctx().num_cu * per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), dyn, ctx().stream, (const cx<T>*)rx, (cx<T>*)X, (const cx<T>*)tw, len, t_guard,
                     n_symb, F, td, fd, tg, fo, ifo, n_keep, (cx<T>*)xp, (const int16_t*)prole, np);
  return check_launch("t4_demod_kernel");
}

// estimate_channel.m:6 per frame, then the spline operator of :8 restricted to rows 1..N_carrier (what equalize_signal reads)
template <typename T>
__global__ void t4_mean_pilots_kernel(const cx<T>* __restrict__ X, const cx<T>* __restrict__ tx, const int32_t* __restrict__ pc0,
                                      cx<T>* __restrict__ hp, int nfft, int np, int n_symb,
                                      const double* __restrict__ fine_est /* rotation not yet applied to X, or null */,
                                      int time_desync, int freq_desync, int compact /* X = X(pilotCarriers,:) [np x S] per frame */) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t f = blockIdx.y;
  if (p >= np) return;
  const cx<T>* rx = X + f * (int64_t)(compact ? np : nfft) * n_symb;
  double rc = 1.0, rs = 0.0;
  if (fine_est) {                                                  // the rotation of fine_apply_kernel for row pc0[p]
    const double tau = fine_est[2 * f], ph = fine_est[2 * f + 1];
    double psn = 0.0, pcs = 1.0, cs = 1.0, sn = 0.0;
    if (freq_desync) sincos(ph, &psn, &pcs);
    if (time_desync) {
      const double t = tau * (double)pc0[p];
      sincospi(2.0 * (t - floor(t)), &sn, &cs);
    }
    rc = cs * pcs - sn * psn;
    rs = sn * pcs + cs * psn;
  }
  double ar = 0, ai = 0;
  const int row = pc0[p];
  for (int s0 = 0; s0 < n_symb; s0 += 10) {                       // ten strided samples requested together
    cx<T> xs[10], ts[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int s = s0 + u < n_symb ? s0 + u : n_symb - 1;
      xs[u] = compact ? rx[(int64_t)s * np + p] : rx[(int64_t)s * nfft + row];
      ts[u] = tx[s * np + p];
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      if (s0 + u < n_symb) {
        cx<T> xv = xs[u];
        if (fine_est) xv = mk<T>((T)((double)xv.x * rc - (double)xv.y * rs), (T)((double)xv.x * rs + (double)xv.y * rc));
        const cx<T> q = cdiv(xv, ts[u]);
        ar += (double)q.x;
        ai += (double)q.y;
      }
    }
  }
  hp[f * np + p] = mk<T>((T)(ar / (double)n_symb), (T)(ai / (double)n_symb));
}

// fine_sync's two estimates (T4/fine_sync.m:10-20, :32-37) and the pilot means of estimate_channel.m:6 of one frame in ONE
// workgroup: the frame's pilot rows (compact, written by the demodulator) come from HBM once and from L2 after that, and the
// rotation of fine_sync.m:24-27 is formed once per pilot row instead of once per (row, symbol) -- they used to be three
// launches, each one sweep through HBM.  The arithmetic is the bodies of fine_tau_kernel /
// fine_phase_kernel / t4_mean_pilots_kernel unchanged.  sync == 0: only the means (no estimates, no rotation).
template <typename T>
__global__ __launch_bounds__(FS_THREADS) void t4_fine_est_kernel(const cx<T>* __restrict__ xp, const cx<T>* __restrict__ tx,
                                                                 const int32_t* __restrict__ pc0, int nfft, int np, int n_symb,
                                                                 double deltak, int sync, int time_desync, int freq_desync,
                                                                 int mp_desync, double* __restrict__ est, cx<T>* __restrict__ hp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fe_smem[];
  double2* const rot = (double2*)fe_smem;            // [np] rotation of fine_sync.m:24-27 per pilot row
  __shared__ FsScratch sc;
  const int64_t f = blockIdx.x;
  const int M = np * n_symb;
  // the frame's pilot rows [np x n_symb] stay where the demodulator wrote them: 48 KB per frame at the C3 geometry, read three
  // times by this workgroup -- from L2 after the first touch.  (Staging them in LDS was slower: 3 instead of 8 resident
  // workgroups per CU and flat loads in the shared reduction bodies, 0.45 against 0.35 ms per 4096 frames.)
  const cx<T>* const lp = xp + f * (int64_t)M;
  double tau = 0.0, ph = 0.0;
  if (sync) {
    PilotView<T> pv{lp, tx, pc0, nfft, np, (int64_t)M, 0, 1};
    tau = fine_tau_body<T>(pv, deltak, 1 /* T4 variant */, sc);
    if (time_desync) {
      for (int p = threadIdx.x; p < np; p += FS_THREADS) {
        double cs, sn;
        fine_rot<T>(tau, pc0[p], cs, sn);
        rot[p] = make_double2(cs, sn);
      }
      __syncthreads();
    }
    ph = fine_phase_body<T>(pv, time_desync, tau, sc, time_desync ? rot : nullptr);
    if (threadIdx.x == 0) { est[2 * f] = tau; est[2 * f + 1] = ph; }
  }
  if (!mp_desync) return;
  for (int p = threadIdx.x; p < np; p += FS_THREADS) {           // t4_mean_pilots_kernel for pilot p
    double rc = 1.0, rs = 0.0;
    if (sync) {
      double psn = 0.0, pcs = 1.0, cs = 1.0, sn = 0.0;
      if (freq_desync) sincos(ph, &psn, &pcs);
      if (time_desync) {
        const double t = tau * (double)pc0[p];
        sincospi(2.0 * (t - floor(t)), &sn, &cs);
      }
      rc = cs * pcs - sn * psn;
      rs = sn * pcs + cs * psn;
    }
    double ar = 0, ai = 0;
    for (int s0 = 0; s0 < n_symb; s0 += 10) {                     // ten strided samples requested together
      cx<T> xs[10], ts[10];
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        const int s = s0 + u < n_symb ? s0 + u : n_symb - 1;
        xs[u] = lp[s * np + p];
        ts[u] = tx[s * np + p];
      }
#pragma unroll
      for (int u = 0; u < 10; ++u) {
        if (s0 + u < n_symb) {
          cx<T> xv = xs[u];
          if (sync) xv = mk<T>((T)((double)xv.x * rc - (double)xv.y * rs), (T)((double)xv.x * rs + (double)xv.y * rc));
          const cx<T> q = cdiv(xv, ts[u]);
          ar += (double)q.x;
          ai += (double)q.y;
        }
      }
    }
    hp[f * np + p] = mk<T>((T)(ar / (double)n_symb), (T)(ai / (double)n_symb));
  }
}

// H(1..N_carrier) = W * Hp for T4_FT frames per workgroup: a weight is read once and used for all of them (one frame per
// workgroup re-read the 429 KB operator from L2 for every frame: 70 us per 1024 frames); the pilot means are wave-uniform
// reads.  double accumulation: the not-a-knot weights alternate in sign.
constexpr int T4_FT = 8;
template <typename T>
__global__ __launch_bounds__(128) void t4_apply_operator_kernel(const T* __restrict__ W, const cx<T>* __restrict__ hp,
                                                                cx<T>* __restrict__ hout, int n_out, int n_in, int64_t n_frames) {
  extern __shared__ __attribute__((aligned(16))) unsigned char t4_smem[];
  cx<T>* hs = (cx<T>*)t4_smem;                                     // [n_in][T4_FT]: the tile's pilot means
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.y * T4_FT;
  for (int i = threadIdx.x; i < n_in * T4_FT; i += blockDim.x) {
    const int f = i / n_in, j = i - f * n_in;
    hs[j * T4_FT + f] = f0 + f < n_frames ? hp[(f0 + f) * n_in + j] : mk<T>(0, 0);
  }
  __syncthreads();
  if (m >= n_out) return;
  double ar[T4_FT], ai[T4_FT];
#pragma unroll
  for (int f = 0; f < T4_FT; ++f) ar[f] = ai[f] = 0.0;
  for (int j0 = 0; j0 < n_in; j0 += 8) {                           // eight weights of the column in flight
    T w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = j0 + u < n_in ? W[(size_t)(j0 + u) * n_out + m] : T(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (j0 + u < n_in) {
        const double wd = (double)w[u];
#pragma unroll
        for (int f = 0; f < T4_FT; ++f) {
          const cx<T> z = hs[(j0 + u) * T4_FT + f];
          ar[f] += wd * (double)z.x;
          ai[f] += wd * (double)z.y;
        }
      }
    }
  }
#pragma unroll
  for (int f = 0; f < T4_FT; ++f)
    if (f0 + f < n_frames) hout[(f0 + f) * n_out + m] = mk<T>((T)ar[f], (T)ai[f]);
}

__global__ void t4_init_kernel(int32_t* __restrict__ stat, int64_t* __restrict__ tg, double* __restrict__ fo, int32_t* __restrict__ ifo,
                               int64_t n_frames) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f < n_frames) { stat[f] = 0; tg[f] = 0; fo[f] = 0.0; ifo[f] = 0; }
}

template <typename T>
__global__ void t4_fill_ones_kernel(cx<T>* __restrict__ h, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) h[i] = mk<T>(1, 0);
}

// ofdm_t4_wave.hip
bool t4_demod_wave_supported(int nfft, int n_keep, int np, bool f64);
int t4_ifo_wave_launch(const void* rx, const void* tw, int64_t len, int t_guard, int64_t F, int td, const int64_t* tg,
                       const double* fo, int32_t* ifo_out, int32_t* status);
int t4_demod_wave_launch(const void* rx, void* X, const void* tw, int64_t len, int t_guard, int n_symb, int64_t F, int td, int fd,
                         const int64_t* tg, const double* fo, const int32_t* ifo, int n_keep, void* xp, const void* prole, int np);

template <typename T>
static int task4_run(ofdm_rx_plan* pl, const void* drx, int64_t F, int time_desync, int freq_desync, int mp_desync,
                     void* dbits, const void* dref, void* derr, int64_t* dtg, double* dfo, int32_t* difo, int32_t* dstat,
                     void* dh_out, Stage& st) {
  const int N = pl->nfft, Tg = pl->t_guard, S = pl->n_symb, np = pl->np, nc = pl->n_carrier;
  const int64_t len = (int64_t)(N + Tg) * S;
  const int64_t n_out = len - Tg - N;
  const bool f64 = std::is_same<T, double>::value;
  hipStream_t s = ctx().stream;
  const bool sync = time_desync || freq_desync;
  // intermediates live in one plan-owned arena (grown on demand): no allocation on the steady-state path
  void *drho = nullptr, *dres, *dresv, *dy, *dseg = nullptr, *dspec = nullptr, *dfirst = nullptr, *dX, *dest, *dhp, *dH;
  size_t need = 0;
  auto reserve = [&](size_t bytes) { const size_t o = need; need += (bytes + 255) & ~size_t(255); return o; };
  const size_t o_res = reserve(sizeof(int64_t) * 2 * F), o_resv = reserve(sizeof(double) * 2 * F);
  const size_t o_y = reserve(sizeof(cx<T>) * (size_t)len * F), o_X = reserve(sizeof(cx<T>) * (size_t)N * S * F);
  const size_t o_est = reserve(sizeof(double) * 2 * F), o_hp = reserve(sizeof(cx<T>) * (size_t)np * F);
  const size_t o_H = reserve(sizeof(cx<T>) * (size_t)nc * F);
  const size_t o_Xp = reserve(sizeof(cx<T>) * (size_t)np * S * F);
  const bool full_acf = getenv("OFDM_T4_FULL_ACF") != nullptr;       // the curve in HBM + the search over it (A/B, tests)
  const size_t o_rho = reserve(sync && full_acf && n_out > 0 ? sizeof(cx<T>) * (size_t)n_out * F : 0);
  const size_t o_seg = reserve(freq_desync ? sizeof(cx<T>) * (size_t)N * F : 0);
  const size_t o_spec = reserve(freq_desync ? sizeof(cx<T>) * (size_t)N * F : 0);
  const size_t o_first = reserve(freq_desync ? sizeof(int64_t) * F : 0);
  if (pl->ws_t4_bytes < need) {
    OFDM_HIP(hipStreamSynchronize(s));
    if (pl->ws_t4) { (void)hipFree(pl->ws_t4); pl->ws_t4 = nullptr; pl->ws_t4_bytes = 0; }
    OFDM_HIP(hipMalloc(&pl->ws_t4, need));
    pl->ws_t4_bytes = need;
  }
  unsigned char* arena = (unsigned char*)pl->ws_t4;
  dres = arena + o_res; dresv = arena + o_resv; dy = arena + o_y; dX = arena + o_X; dest = arena + o_est;
  dhp = arena + o_hp; dH = arena + o_H;
  void* dXp = arena + o_Xp;
  // stage brackets (ofdm_rx_plan_set_timing): [0] start, [1] AutoCorrFunction stage, [2] remove_IFO stage, [3] demodulator,
  // [4] fine_sync + estimate_channel, [5] equalise / demap / (DeScrambler)
  const bool timed = pl->timing && pl->ev_t4[0];
  pl->t4_timed = timed ? 1 : 0;
  auto mark = [&](int i) { if (timed) (void)hipEventRecord(pl->ev_t4[i], s); };
  mark(0);
  // per-frame scalars start at zero: one launch instead of four memsets (4.7 us each)
  hipLaunchKernelGGL(t4_init_kernel, dim3(cdiv_u(F, 256)), dim3(256), 0, s, dstat, dtg, dfo, difo, F);
  if (sync) {
    OFDM_ARG(n_out > 0 && Tg >= 1 && Tg <= ACF_MAXW, "rx_chain_task4: frame shorter than T_guard + Nfft, or T_guard outside 1..%d", ACF_MAXW);
    OFDM_TRY(acf_prepare<T>());
    if (!full_acf) {
      // one workgroup per frame walks the tiles of the autocorrelation until the search of AutoCorrFunction.m:10-20 is
      // decided (t4_acf_search_kernel): no curve in HBM, one launch
      hipLaunchKernelGGL(t4_acf_search_kernel<T>, dim3((unsigned)F), dim3(ACF_THREADS), acf_search_lds_bytes<T>(Tg), s, (const cx<T>*)drx, len,
                         Tg, N, n_out, 0.77, dtg, dfo, dstat);
    } else {
      // OFDM_T4_FULL_ACF: the whole curve per frame, then the search over it (the kernels of ofdm_AutoCorrFunction)
      drho = arena + o_rho;
      hipLaunchKernelGGL(acf_kernel<T>, dim3(cdiv_u(n_out, ACF_TILE), (unsigned)F), dim3(ACF_THREADS), acf_lds_bytes<T>(Tg), s, (const cx<T>*)drx, len, Tg,
                         N, (cx<T>*)drho, n_out, n_out);
      hipLaunchKernelGGL(acf_plateau_kernel<T>, dim3((unsigned)F), dim3(PLAT_THREADS), 0, s, (const cx<T>*)drho, n_out, Tg, 0.77,
                         (int64_t*)dres, (double*)dresv, n_out);
      hipLaunchKernelGGL(t4_scalars_kernel, dim3(cdiv_u(F, 256)), dim3(256), 0, s, (const int64_t*)dres, (const double*)dresv, n_out,
                         dtg, dfo, dstat, F);
    }
    OFDM_TRY(check_launch("AutoCorrFunction stage"));
  }
  mark(1);
  const unsigned gl = (unsigned)std::min<int64_t>((len + 255) / 256, 64);
  const int td_eff = sync ? time_desync : 0, fd_eff = sync ? freq_desync : 0;
  const bool direct = (N == 512 || N == 1024 || N == 2048 || N == 4096) && !getenv("OFDM_T4_STAGED");
  if (freq_desync) OFDM_ARG(len >= 2 * (int64_t)N, "rx_chain_task4: rx_signal(Nfft+1:2*Nfft) exceeds the frame");
  if (direct) {
    // no aligned / corrected copy of the batch: the IFO search reads its segment, the demodulator its samples, from rx
    const void* twd = nullptr;
    OFDM_TRY(get_twiddles(N, f64, &twd));
    const bool wave = t4_demod_wave_supported(N, nc, np, f64);
    if (freq_desync && wave) {                       // segment, spectrum and search of remove_IFO.m:5-8 in one launch
      OFDM_TRY(t4_ifo_wave_launch(drx, twd, len, Tg, F, td_eff, dtg, dfo, difo, dstat));
    } else if (freq_desync) {
      dseg = arena + o_seg; dspec = arena + o_spec; dfirst = arena + o_first;
      hipLaunchKernelGGL(t4_segment_direct_kernel<T>, dim3(cdiv_u(N, 256), (unsigned)F), dim3(256), 0, s, (const cx<T>*)drx, (cx<T>*)dseg,
                         len, N + Tg, N, td_eff, (const int64_t*)dtg, (const double*)dfo);
      OFDM_TRY(demod_device(dseg, dspec, N, F, 0, f64));
      hipLaunchKernelGGL(first_above_kernel<T>, dim3((unsigned)F), dim3(PLAT_THREADS), 0, s, (const cx<T>*)dspec, (int64_t)N, 0.77,
                         (int64_t*)dfirst);
      hipLaunchKernelGGL(t4_ifo_finalize_kernel, dim3(cdiv_u(F, 256)), dim3(256), 0, s, (const int64_t*)dfirst, difo, dstat, F);
      OFDM_TRY(check_launch("remove_IFO stage"));
    }
    mark(2);
    if (wave) {         // Nfft 2048, fp32, N_carrier <= 1024: one wavefront per symbol run
      OFDM_TRY(t4_demod_wave_launch(drx, dX, twd, len, Tg, S, F, td_eff, fd_eff, dtg, dfo, difo, nc, dXp, pl->d_prole, np));
    } else
    switch (N / 512) {
      case 1: OFDM_TRY((t4_demod_launch<T, 1>(drx, dX, twd, len, Tg, S, F, td_eff, fd_eff, dtg, dfo, difo, nc, dXp, pl->d_prole, np))); break;
      case 2: OFDM_TRY((t4_demod_launch<T, 2>(drx, dX, twd, len, Tg, S, F, td_eff, fd_eff, dtg, dfo, difo, nc, dXp, pl->d_prole, np))); break;
      case 4: OFDM_TRY((t4_demod_launch<T, 4>(drx, dX, twd, len, Tg, S, F, td_eff, fd_eff, dtg, dfo, difo, nc, dXp, pl->d_prole, np))); break;
      default: OFDM_TRY((t4_demod_launch<T, 8>(drx, dX, twd, len, Tg, S, F, td_eff, fd_eff, dtg, dfo, difo, nc, dXp, pl->d_prole, np))); break;
    }
  } else {
    hipLaunchKernelGGL(t4_align_kernel<T>, dim3(gl, (unsigned)F), dim3(256), 0, s, (const cx<T>*)drx, (cx<T>*)dy, len, N + Tg, N,
                       td_eff, fd_eff, (const int64_t*)dtg, (const double*)dfo);
    OFDM_TRY(check_launch("t4_align_kernel"));
    if (freq_desync) {
      dseg = arena + o_seg; dspec = arena + o_spec; dfirst = arena + o_first;
      hipLaunchKernelGGL(t4_segment_kernel<T>, dim3(cdiv_u(N, 256), (unsigned)F), dim3(256), 0, s, (const cx<T>*)dy, (cx<T>*)dseg, len, N);
      OFDM_TRY(demod_device(dseg, dspec, N, F, 0, f64));
      hipLaunchKernelGGL(first_above_kernel<T>, dim3((unsigned)F), dim3(PLAT_THREADS), 0, s, (const cx<T>*)dspec, (int64_t)N, 0.77,
                         (int64_t*)dfirst);
      hipLaunchKernelGGL(t4_ifo_kernel<T>, dim3(gl, (unsigned)F), dim3(256), 0, s, (cx<T>*)dy, len, N, (const int64_t*)dfirst, difo, dstat);
      OFDM_TRY(check_launch("remove_IFO stage"));
    }
    mark(2);
    OFDM_TRY(demod_device(dy, dX, N, (int64_t)S * F, Tg, f64));                       // T4:308-310
  }
  mark(3);
  // pilot matrix [np x S] = the plan's pilot column on every symbol (T4:28-31) and the spline operator of
  // estimate_channel.m:8 for rows 1..N_carrier: built once per plan, kept on the device
  if (!pl->d_t4_tx) {
    std::vector<cx<T>> col(np), txm((size_t)np * S);
    OFDM_HIP(hipMemcpyAsync(col.data(), pl->d_pilots, sizeof(cx<T>) * np, hipMemcpyDeviceToHost, s));
    OFDM_HIP(hipStreamSynchronize(s));
    for (int sy = 0; sy < S; ++sy) std::copy(col.begin(), col.end(), txm.begin() + (size_t)sy * np);
    OFDM_HIP(hipMalloc(&pl->d_t4_tx, sizeof(cx<T>) * txm.size()));
    OFDM_HIP(hipMemcpy(pl->d_t4_tx, txm.data(), sizeof(cx<T>) * txm.size(), hipMemcpyHostToDevice));
    std::vector<double> xk(np), xq(nc), W;
    for (int i = 0; i < np; ++i) xk[i] = pl->pilot_loc[i];
    for (int i = 0; i < nc; ++i) xq[i] = i + 1.0;
    OFDM_TRY(build_spline_operator(xk, xq, W));                                       // estimate_channel.m:8, rows 1..N_carrier
    std::vector<T> Wt(W.begin(), W.end());
    OFDM_HIP(hipMalloc(&pl->d_t4_w, sizeof(T) * Wt.size()));
    OFDM_HIP(hipMemcpy(pl->d_t4_w, Wt.data(), sizeof(T) * Wt.size(), hipMemcpyHostToDevice));
    if (!f64) {                                      // fp32: the operator's band (weights below 1e-10 of a row's largest dropped)
      std::vector<float> bw_w;
      std::vector<int32_t> bw_c0;
      mmse_band_spline(W, nc, np, bw_w, bw_c0, pl->t4_bw, pl->t4_span);
      OFDM_HIP(hipMalloc(&pl->d_t4_bw, sizeof(float) * bw_w.size()));
      OFDM_HIP(hipMemcpy(pl->d_t4_bw, bw_w.data(), sizeof(float) * bw_w.size(), hipMemcpyHostToDevice));
      OFDM_HIP(hipMalloc(&pl->d_t4_bc0, sizeof(int32_t) * bw_c0.size()));
      OFDM_HIP(hipMemcpy(pl->d_t4_bc0, bw_c0.data(), sizeof(int32_t) * bw_c0.size(), hipMemcpyHostToDevice));
    }
  }
  const void *dtx = pl->d_t4_tx, *dW = pl->d_t4_w;
  if (sync) OFDM_ARG(np >= 2, "rx_chain_task4: fine_sync needs at least two pilot carriers");
  const double deltak = np >= 2 ? (double)pl->pilot_loc[1] - (double)pl->pilot_loc[0] : 1.0;      // fine_sync.m:6
  // direct form: fine_sync's two estimates and the pilot means in ONE launch on the frame's compact pilot rows held in LDS
  const size_t fe_lds = sizeof(double2) * (size_t)np;
  const bool fused_est = direct && fe_lds <= 48 * 1024 && (sync || mp_desync) && !getenv("OFDM_T4_UNFUSED_SYNC");
  if (fused_est) {
    OFDM_HIP(hipFuncSetAttribute((const void*)t4_fine_est_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fe_lds));
    hipLaunchKernelGGL(t4_fine_est_kernel<T>, dim3((unsigned)F), dim3(FS_THREADS), fe_lds, s, (const cx<T>*)dXp, (const cx<T>*)dtx,
                       (const int32_t*)pl->d_pc0, N, np, S, deltak, sync ? 1 : 0, time_desync, freq_desync, mp_desync, (double*)dest,
                       (cx<T>*)dhp);
    OFDM_TRY(check_launch("t4_fine_est_kernel"));
  } else if (sync) {
    PilotView<T> pv{direct ? (const cx<T>*)dXp : (const cx<T>*)dX, (const cx<T>*)dtx, (const int32_t*)pl->d_pc0, N, np, (int64_t)np * S,
                    direct ? (int64_t)np * S : (int64_t)N * S, direct ? 1 : 0};
    hipLaunchKernelGGL(fine_tau_kernel<T>, dim3((unsigned)F), dim3(FS_THREADS), 0, s, pv, deltak, 1 /* T4 variant */, (double*)dest);
    hipLaunchKernelGGL(fine_phase_kernel<T>, dim3((unsigned)F), dim3(FS_THREADS), 0, s, pv, time_desync, (double*)dest);
    // staged form: rewrite X; otherwise the rotation is applied where X is read (pilot means, equaliser)
    if (!direct)
      hipLaunchKernelGGL(fine_apply_kernel<T>, dim3(64, (unsigned)F), dim3(256), 0, s, (const cx<T>*)dX, (cx<T>*)dX, N, (int64_t)S,
                         time_desync, freq_desync, (const double*)dest);
    OFDM_TRY(check_launch("fine_sync stage"));
  }
  const double* lazy_rot = (sync && direct) ? (const double*)dest : nullptr;
  if (mp_desync) {
    if (!fused_est)
      hipLaunchKernelGGL(t4_mean_pilots_kernel<T>, dim3(cdiv_u(np, 128), (unsigned)F), dim3(128), 0, s,
                         direct ? (const cx<T>*)dXp : (const cx<T>*)dX, (const cx<T>*)dtx, (const int32_t*)pl->d_pc0, (cx<T>*)dhp, N, np, S,
                         lazy_rot, time_desync, freq_desync, direct ? 1 : 0);
    int band = 1;                                    // fp32: banded product (0.084 -> 0.04 ms per 4096 frames at C3), else the dense tile product
    if (!f64 && pl->d_t4_bw && !getenv("OFDM_T4_DENSE_SPLINE"))
      band = spline_band_run((const float*)pl->d_t4_bw, (const int32_t*)pl->d_t4_bc0, pl->t4_bw, pl->t4_span, dhp, dH, np, nc, F);
    OFDM_TRY(band);
    if (band == 1)
      hipLaunchKernelGGL(t4_apply_operator_kernel<T>, dim3(cdiv_u(nc, 128), cdiv_u(F, T4_FT)), dim3(128), sizeof(cx<T>) * np * T4_FT, s, (const T*)dW,
                         (const cx<T>*)dhp, (cx<T>*)dH, nc, np, F);
  } else {
    hipLaunchKernelGGL(t4_fill_ones_kernel<T>, dim3(256), dim3(256), 0, s, (cx<T>*)dH, (int64_t)nc * F);
  }
  OFDM_TRY(check_launch("estimate_channel stage"));
  mark(4);
  FastPlanView pv2;
  make_plan_view(pl, pv2);
  pv2.ev = nullptr;
  FastParams<T> P;
  const void* tw = nullptr;
  OFDM_TRY(get_twiddles(N, f64, &tw));
  OFDM_TRY(fast_params_prepare<T>(pv2, tw, F, P));
  P.h_in = (const cx<T>*)dH;
  // the per-frame DeScrambler of T4:354-364 (ofdm_rx_plan_set_descrambler) as a pass over the packed decisions
  const bool descr_pass = (pl->descr & DESCR_ON) != 0;
  void* craw = nullptr;
  if (descr_pass) OFDM_TRY(descr_raw_workspace(pl, F, &craw));
  OFDM_TRY(eq_demap_run<T>(pv2, P, (const cx<T>*)dX, direct ? nc : N, true, F, descr_pass ? craw : dbits,
                           descr_pass ? nullptr : dref, descr_pass ? nullptr : derr, dh_out, nullptr, lazy_rot,
                           time_desync, freq_desync));                                                                   // T4:334-347
  if (descr_pass) OFDM_TRY(descr_pass_run(pl, craw, dbits, dref, derr, F));
  mark(5);
  return OFDM_OK;
}

}  // namespace ofdm

using namespace ofdm;

extern "C" {

int ofdm_rx_chain_task4(ofdm_rx_plan* pl, const void* rx, int64_t n_frames, int time_desync, int freq_desync, int mp_desync,
                        uint8_t* bits_out, const uint8_t* ref_bits, uint32_t* errors_out, int64_t* tg_position_out,
                        double* freq_offset_out, int32_t* ifo_out, int32_t* status_out, void* h_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(pl && rx && n_frames >= 0, "rx_chain_task4: bad arguments");
  OFDM_ARG(pl->nd >= 1, "rx_chain_task4: the plan has no data carriers");
  OFDM_PLAN_DEVICE(pl);
  OFDM_ARG((is_f64(flags) ? 1 : 0) == pl->f64, "rx_chain_task4: precision flag differs from the plan's");
  OFDM_ARG(pl->pilots_in_band, "rx_chain_task4: pilots outside 1..N_carrier are not supported");
  OFDM_ARG(!errors_out || ref_bits, "rx_chain_task4: errors_out needs ref_bits");
  OFDM_ARG(n_frames <= 65535, "rx_chain_task4: at most 65535 frames per call");
  if (n_frames == 0) return OFDM_OK;
  const size_t cs = csize(flags);
  const size_t frame_samples = (size_t)(pl->nfft + pl->t_guard) * pl->n_symb;
  const size_t fb = (size_t)pl->frame_words * 4;
  Stage st(flags);
  const void *drx, *dref; void *dbits, *derr, *dtg, *dfo, *difo, *dstat, *dh;
  OFDM_TRY(st.in(rx, cs * frame_samples * n_frames, &drx));
  OFDM_TRY(st.in(ref_bits, fb * n_frames, &dref));
  OFDM_TRY(st.out(bits_out, fb * n_frames, &dbits));
  OFDM_TRY(st.out(errors_out, sizeof(uint32_t) * n_frames, &derr));
  OFDM_TRY(st.out(h_out, cs * (size_t)pl->n_carrier * n_frames, &dh));
  // the per-frame scalars are always produced (scratch when the caller does not want them)
  if (tg_position_out) OFDM_TRY(st.out(tg_position_out, sizeof(int64_t) * n_frames, &dtg)); else OFDM_TRY(st.scratch(sizeof(int64_t) * n_frames, &dtg));
  if (freq_offset_out) OFDM_TRY(st.out(freq_offset_out, sizeof(double) * n_frames, &dfo)); else OFDM_TRY(st.scratch(sizeof(double) * n_frames, &dfo));
  if (ifo_out) OFDM_TRY(st.out(ifo_out, sizeof(int32_t) * n_frames, &difo)); else OFDM_TRY(st.scratch(sizeof(int32_t) * n_frames, &difo));
  if (status_out) OFDM_TRY(st.out(status_out, sizeof(int32_t) * n_frames, &dstat)); else OFDM_TRY(st.scratch(sizeof(int32_t) * n_frames, &dstat));
  if (pl->f64)
    OFDM_TRY(task4_run<double>(pl, drx, n_frames, time_desync, freq_desync, mp_desync, dbits, dref, derr, (int64_t*)dtg, (double*)dfo,
                               (int32_t*)difo, (int32_t*)dstat, dh, st));
  else
    OFDM_TRY(task4_run<float>(pl, drx, n_frames, time_desync, freq_desync, mp_desync, dbits, dref, derr, (int64_t*)dtg, (double*)dfo,
                              (int32_t*)difo, (int32_t*)dstat, dh, st));
  return st.finish();
}

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
  OFDM_TRY(f64 ? acf_prepare<double>() : acf_prepare<float>());
  if (f64) {
    hipLaunchKernelGGL(acf_kernel<double>, dim3(grid), dim3(ACF_THREADS), acf_lds_bytes<double>(width_window), ctx().stream, (const c64*)dx, len,
                       width_window, nfft, (c64*)drho, n_out);
    OFDM_TRY(check_launch("acf_kernel"));
    hipLaunchKernelGGL(acf_plateau_kernel<double>, dim3(1), dim3(PLAT_THREADS), 0, ctx().stream, (const c64*)drho,
                       n_out, width_window, 0.77, (int64_t*)dres, (double*)dresv);
  } else {
    hipLaunchKernelGGL(acf_kernel<float>, dim3(grid), dim3(ACF_THREADS), acf_lds_bytes<float>(width_window), ctx().stream, (const c32*)dx, len,
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
    PilotView<double> pv{(const c64*)dx, (const c64*)dtx, (const int32_t*)dpc, nfft, n_pilots, (int64_t)n_pilots * n_symb, 0, 0};
    hipLaunchKernelGGL(fine_tau_kernel<double>, dim3(1), dim3(FS_THREADS), 0, ctx().stream, pv, deltak, variant, (double*)dest);
    hipLaunchKernelGGL(fine_phase_kernel<double>, dim3(1), dim3(FS_THREADS), 0, ctx().stream, pv, time_desync, (double*)dest);
    hipLaunchKernelGGL(fine_apply_kernel<double>, dim3(ew_grid(total)), dim3(256), 0, ctx().stream, (const c64*)dx,
                       (c64*)dout, nfft, n_symb, time_desync, freq_desync, (const double*)dest);
  } else {
    PilotView<float> pv{(const c32*)dx, (const c32*)dtx, (const int32_t*)dpc, nfft, n_pilots, (int64_t)n_pilots * n_symb, 0, 0};
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
