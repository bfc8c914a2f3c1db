// Channel estimators: interpolate / estimate_channel / LS_CE / MMSE_CE (spline as a precomputed
// linear operator, Levinson solve of the Hermitian-Toeplitz MMSE system) and the single-problem
// MP_estimate / OMP_estimate (the batched OMP of the benchmark chain lives in ofdm_chain.hip).
#include "spline_op.hpp"
#include "pursuit_core.hpp"

namespace ofdm {

int demod_device(const void* y, void* x, int nfft, int64_t n_symb, int t_guard, bool f64);   // ofdm_modem.hip

// ---------------------------------------------------------------------------------------------
// out[m] = sum_j W[m,j] * v[j]   (W real, column-major [n_out x n_in], v complex)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void apply_operator_kernel(const T* __restrict__ W, const cx<T>* __restrict__ v, cx<T>* __restrict__ out,
                                      int n_out, int n_in) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_out) return;
  // accumulate in double: the not-a-knot weights alternate in sign
  double ar = 0, ai = 0;
  for (int j = 0; j < n_in; ++j) {
    const double w = (double)W[(size_t)j * n_out + m];
    const cx<T> z = v[j];
    ar += w * (double)z.x;
    ai += w * (double)z.y;
  }
  out[m] = mk<T>((T)ar, (T)ai);
}

// LS at the pilots of symbol `sym` : out[p] = Y[pilot_p, sym] / Xp[p, sym]     (LS_CE.m:27-28)
template <typename T>
__global__ void ls_pilots_kernel(const cx<T>* __restrict__ y, const cx<T>* __restrict__ xp,
                                 const int32_t* __restrict__ pc0, cx<T>* __restrict__ out, int np) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < np) out[p] = cdiv(y[pc0[p]], xp[p]);
}

// mean over symbols of rx_p ./ tx_p                                           (estimate_channel.m:6)
template <typename T>
__global__ void mean_pilots_kernel(const cx<T>* __restrict__ rx, const cx<T>* __restrict__ tx,
                                   const int32_t* __restrict__ pc0, cx<T>* __restrict__ out, int nfft, int np,
                                   int64_t n_symb) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  double ar = 0, ai = 0;
  for (int64_t s = 0; s < n_symb; ++s) {
    const cx<T> q = cdiv(rx[s * nfft + pc0[p]], tx[s * np + p]);
    ar += (double)q.x;
    ai += (double)q.y;
  }
  out[p] = mk<T>((T)(ar / (double)n_symb), (T)(ai / (double)n_symb));
}

// ---------------------------------------------------------------------------------------------
// MMSE_CE.m:19-36 in one workgroup: tau_rms from h, Hermitian-Toeplitz Rpp = rf2 + I/snr,
// z = Rpp \ H_tilde by the Levinson recursion (double), H_p = rf2 * z (only the first Np rows of
// Rhp are ever used because of MMSE_CE.m:38).
// ---------------------------------------------------------------------------------------------
constexpr int MMSE_THREADS = 256;
constexpr int MMSE_MAXNP = 1024;

__device__ __forceinline__ c64 block_sum_c64(c64 v, c64* sh /* [MMSE_THREADS/64] */) {
  for (int off = 32; off > 0; off >>= 1) {
    v.x += __shfl_down(v.x, off, 64);
    v.y += __shfl_down(v.y, off, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  c64 t = sh[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = t + sh[w];
  return t;
}

template <typename T>
__global__ __launch_bounds__(MMSE_THREADS) void mmse_kernel(const cx<T>* __restrict__ h, int h_len,
                                                            const cx<T>* __restrict__ h_tilde, int np, double nps,
                                                            double df, double inv_snr, cx<T>* __restrict__ hp_out,
                                                            double* __restrict__ tau_out) {
  __shared__ c64 red[MMSE_THREADS / 64];
  __shared__ c64 tcol[MMSE_MAXNP];     // first column of rf2: t[k] = 1/(1 + j c k)
  __shared__ c64 fv[MMSE_MAXNP], bv[MMSE_MAXNP], xv[MMSE_MAXNP];
  const int tid = threadIdx.x;
  // ---- rms delay spread (:19-24)
  c64 s0{0, 0}, s1{0, 0};
  for (int k = tid; k < h_len; k += MMSE_THREADS) {
    const double p = (double)h[k].x * h[k].x + (double)h[k].y * h[k].y;
    s0.x += p;
    s0.y += p * (double)k;
    s1.x += p * (double)k * (double)k;
  }
  s0 = block_sum_c64(s0, red);
  s1 = block_sum_c64(s1, red);
  const double hh = s0.x, r = s0.y / hh, r2 = s1.x / hh;
  const double tau_rms = sqrt(r2 - r * r);
  if (tid == 0 && tau_out) tau_out[0] = tau_rms;
  const double c = 2.0 * M_PI * tau_rms * df * nps;              // j2pi_tau_df * Nps  (:26,:30)
  // ---- Toeplitz column: rf2[i][j] = 1/(1 + j c (i-j))  (:33) ; diag += 1/snr (:35)
  for (int k = tid; k < np; k += MMSE_THREADS) {
    const double d = 1.0 + (c * k) * (c * k);
    tcol[k] = c64{1.0 / d, -(c * k) / d};
  }
  __syncthreads();
  const double t0 = tcol[0].x + inv_snr;
  auto Tm = [&](int i, int j) -> c64 {                            // Rpp[i][j]
    const int k = i - j;
    if (k == 0) return c64{t0, 0.0};
    return k > 0 ? tcol[k] : conj(tcol[-k]);
  };
  // ---- Levinson: f (T_n f = e_1), b (T_n b = e_n), x (T_n x = y_1..n)
  if (tid == 0) {
    fv[0] = c64{1.0 / t0, 0};
    bv[0] = c64{1.0 / t0, 0};
    const c64 y0{(double)h_tilde[0].x, (double)h_tilde[0].y};
    xv[0] = c64{y0.x / t0, y0.y / t0};
  }
  __syncthreads();
  for (int n = 1; n < np; ++n) {
    // eps_f = sum_i T[n][i] f[i], eps_x = sum_i T[n][i] x[i], eps_b = sum_i T[0][i+1] b[i],  i < n
    c64 ef{0, 0}, ex{0, 0}, eb{0, 0};
    for (int i = tid; i < n; i += MMSE_THREADS) {
      const c64 tn = Tm(n, i);
      ef = ef + tn * fv[i];
      ex = ex + tn * xv[i];
      eb = eb + Tm(0, i + 1) * bv[i];
    }
    ef = block_sum_c64(ef, red);
    ex = block_sum_c64(ex, red);
    eb = block_sum_c64(eb, red);
    const c64 one{1, 0};
    const c64 den = one - eb * ef;
    const c64 inv = cdiv(one, den);
    const c64 yn{(double)h_tilde[n].x, (double)h_tilde[n].y};
    const c64 dx = yn - ex;
    // new f = inv*[f;0] - ef*inv*[0;b] ; new b = inv*[0;b] - eb*inv*[f;0]
    constexpr int PER = MMSE_MAXNP / MMSE_THREADS;
    c64 nf[PER], nb[PER];
#pragma unroll
    for (int c2 = 0; c2 < PER; ++c2) {
      const int i = tid + c2 * MMSE_THREADS;
      if (i <= n) {
        const c64 fe = (i < n) ? fv[i] : c64{0, 0};
        const c64 be = (i > 0) ? bv[i - 1] : c64{0, 0};
        nf[c2] = inv * fe - (ef * inv) * be;
        nb[c2] = inv * be - (eb * inv) * fe;
      }
    }
    __syncthreads();
#pragma unroll
    for (int c2 = 0; c2 < PER; ++c2) {
      const int i = tid + c2 * MMSE_THREADS;
      if (i <= n) {
        fv[i] = nf[c2];
        bv[i] = nb[c2];
        const c64 xo = (i < n) ? xv[i] : c64{0, 0};
        xv[i] = xo + dx * nb[c2];
      }
    }
    __syncthreads();
  }
  // ---- H_p = rf2 * z   (Rhp rows 0..Np-1 ; no 1/snr on this diagonal)
  for (int i = tid; i < np; i += MMSE_THREADS) {
    c64 acc{0, 0};
    for (int j = 0; j < np; ++j) {
      const int k = i - j;
      const c64 t = k >= 0 ? tcol[k] : conj(tcol[-k]);
      acc = acc + t * xv[j];
    }
    hp_out[i] = mk<T>((T)acc.x, (T)acc.y);
  }
}

// ---------------------------------------------------------------------------------------------
// sensing matrix S[p,k] = exp(-2 pi i (pilot_p - 1)(k - 1)/Nfft)     (Main_model_Task_5.m:182-190)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void sensing_kernel(const int32_t* __restrict__ pc0, int np, int nfft, int k_atoms, cx<T>* __restrict__ s) {
  const int64_t total = (int64_t)np * k_atoms;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % np);
    const int k = (int)(i / np);
    const int64_t ph = ((int64_t)pc0[p] * k) % nfft;
    double sn, cs;
    sincospi(2.0 * (double)ph / (double)nfft, &sn, &cs);
    s[i] = mk<T>((T)cs, (T)(-sn));
  }
}

// ---------------------------------------------------------------------------------------------
// Single-problem MP / OMP.  One workgroup of 1024 threads (16 waves); correlations are formed one
// column per wave with lanes striding over the pilot rows (coalesced) and accumulated in double
// (argmax stability, SURVEY.md section 7 "hard parts").  y/r live in LDS as complex double.
// ---------------------------------------------------------------------------------------------
constexpr int PUR_THREADS = 1024;
constexpr int PUR_WAVES = PUR_THREADS / 64;
constexpr int PUR_MAXNP = 2048;
constexpr int PUR_MAXT = 32;

__device__ __forceinline__ c64 wave_sum_c64(c64 v) {
  for (int off = 32; off > 0; off >>= 1) {
    v.x += __shfl_down(v.x, off, 64);
    v.y += __shfl_down(v.y, off, 64);
  }
  return v;      // valid in lane 0
}

// conj(S[:,k])^T * r
template <typename T>
__device__ __forceinline__ c64 col_dot(const cx<T>* __restrict__ col, const c64* __restrict__ r, int np, int lane) {
  c64 acc{0, 0};
  for (int p = lane; p < np; p += 64) {
    const c64 a{(double)col[p].x, (double)col[p].y};
    acc = acc + mulc(r[p], a);          // r * conj(a)
  }
  return wave_sum_c64(acc);
}

struct BestPick { double score; int idx; };

__device__ __forceinline__ void best_update(BestPick& b, double s, int idx) {
  if (s > b.score || (s == b.score && idx < b.idx)) { b.score = s; b.idx = idx; }
}

// argmax over columns [0, ncols) of score(k); excluded columns get `excl_score`.  Returns in all threads.
template <typename T, bool MP>
__device__ int pursuit_argmax(const cx<T>* __restrict__ S, const c64* __restrict__ r, int np, int ncols,
                              const int* __restrict__ picked, int n_picked, BestPick* sh_best) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  BestPick best{-INFINITY, 0x7fffffff};
  for (int k = wid; k < ncols; k += PUR_WAVES) {
    double score;
    bool excluded = false;
    if (MP) for (int j = 0; j < n_picked; ++j) excluded |= (picked[j] == k);
    if (excluded) {
      score = -100.0;                                           // MP_estimate.m:12
    } else {
      const cx<T>* col = S + (size_t)k * np;
      const c64 d = col_dot<T>(col, r, np, lane);
      if (MP) {
        c64 nn{0, 0};
        for (int p = lane; p < np; p += 64) nn.x += (double)col[p].x * col[p].x + (double)col[p].y * col[p].y;
        nn = wave_sum_c64(nn);
        score = (d.x * d.x + d.y * d.y) / nn.x;                 // MP_estimate.m:15
      } else {
        score = sqrt(d.x * d.x + d.y * d.y);                    // OMP_estimate.m:7,:14
      }
    }
    if (lane == 0) best_update(best, score, k);
  }
  __syncthreads();
  if (lane == 0) sh_best[wid] = best;
  __syncthreads();
  BestPick b = sh_best[0];
  for (int w = 1; w < PUR_WAVES; ++w) best_update(b, sh_best[w].score, sh_best[w].idx);
  __syncthreads();
  return b.idx < ncols ? b.idx : 0;     // all-NaN scores: MATLAB max returns index 1
}

template <typename T>
__global__ __launch_bounds__(PUR_THREADS) void mp_kernel(const cx<T>* __restrict__ y, const cx<T>* __restrict__ S,
                                                         int np, int k_atoms, int taps, int nfft,
                                                         cx<T>* __restrict__ h_out, int32_t* __restrict__ picks_out) {
  __shared__ c64 r[PUR_MAXNP];
  __shared__ BestPick sh_best[PUR_WAVES];
  __shared__ int picked[PUR_MAXT];
  __shared__ c64 coef_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int p = tid; p < np; p += PUR_THREADS) r[p] = c64{(double)y[p].x, (double)y[p].y};
  for (int i = tid; i < nfft; i += PUR_THREADS) h_out[i] = mk<T>(0, 0);     // MP_estimate.m:27
  __syncthreads();
  for (int it = 0; it < taps; ++it) {
    const int kp = pursuit_argmax<T, true>(S, r, np, np /* loop bound is Np, MP_estimate.m:10 */, picked, it, sh_best);
    const cx<T>* col = S + (size_t)kp * np;
    if (wid == 0) {
      const c64 d = col_dot<T>(col, r, np, lane);
      c64 nn{0, 0};
      for (int p = lane; p < np; p += 64) nn.x += (double)col[p].x * col[p].x + (double)col[p].y * col[p].y;
      nn = wave_sum_c64(nn);
      if (lane == 0) { coef_s = c64{d.x / nn.x, d.y / nn.x}; picked[it] = kp; }   // :22
    }
    __syncthreads();
    const c64 x = coef_s;
    for (int p = tid; p < np; p += PUR_THREADS) {
      const c64 a{(double)col[p].x, (double)col[p].y};
      r[p] = r[p] - a * x;                                                       // :23
    }
    if (tid == 0) {
      h_out[kp] = mk<T>((T)x.x, (T)x.y);                                         // :28-30
      if (picks_out) picks_out[it] = kp + 1;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(PUR_THREADS) void omp_kernel(const cx<T>* __restrict__ y, const cx<T>* __restrict__ S,
                                                          int np, int k_atoms, int taps, int nfft,
                                                          cx<T>* __restrict__ h_out, int32_t* __restrict__ index_out,
                                                          int32_t* __restrict__ n_index_out) {
  __shared__ c64 yv[PUR_MAXNP];
  __shared__ c64 r[PUR_MAXNP];
  __shared__ BestPick sh_best[PUR_WAVES];
  __shared__ int picked[PUR_MAXT];
  __shared__ c64 G[PUR_MAXT * PUR_MAXT], Lm[PUR_MAXT * PUR_MAXT];
  __shared__ c64 bvec[PUR_MAXT], xv[PUR_MAXT];
  __shared__ double nrm[2 * PUR_WAVES];
  __shared__ int stop_s, n_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int p = tid; p < np; p += PUR_THREADS) { yv[p] = c64{(double)y[p].x, (double)y[p].y}; r[p] = yv[p]; }
  for (int i = tid; i < nfft; i += PUR_THREADS) h_out[i] = mk<T>(0, 0);
  if (tid == 0) { stop_s = 0; n_s = 0; }
  __syncthreads();
  for (int it = 0; it < taps; ++it) {
    const int kp = pursuit_argmax<T, false>(S, r, np, k_atoms, picked, it, sh_best);
    // duplicate pick: pinv of a matrix with two identical columns splits the coefficient equally
    // between them and the residual does not change -> the loop breaks (OMP_estimate.m:17-22).
    int dup = -1;
    for (int j = 0; j < it; ++j) if (picked[j] == kp) dup = j;
    if (dup >= 0) {
      if (tid == 0) {
        const c64 half{xv[dup].x * 0.5, xv[dup].y * 0.5};
        xv[dup] = half;
        xv[it] = half;
        picked[it] = kp;
        n_s = it + 1;
        stop_s = 1;
      }
      __syncthreads();
      break;
    }
    const cx<T>* col = S + (size_t)kp * np;
    // new Gram row/column and right-hand side: one wave per entry
    for (int j = wid; j <= it + 1; j += PUR_WAVES) {
      if (j <= it) {
        const cx<T>* cj = (j == it) ? col : S + (size_t)picked[j] * np;
        c64 acc{0, 0};
        for (int p = lane; p < np; p += 64) {
          const c64 a{(double)cj[p].x, (double)cj[p].y}, b{(double)col[p].x, (double)col[p].y};
          acc = acc + mulc(b, a);                     // conj(a_j) * a_new  = G[j][it]
        }
        acc = wave_sum_c64(acc);
        if (lane == 0) { G[j * PUR_MAXT + it] = acc; G[it * PUR_MAXT + j] = conj(acc); }
      } else {
        const c64 d = col_dot<T>(col, yv, np, lane);   // a_new^H y
        if (lane == 0) bvec[it] = d;
      }
    }
    if (tid == 0) picked[it] = kp;
    __syncthreads();
    if (tid == 0) chol_solve(G, bvec, Lm, xv, it + 1, PUR_MAXT);       // x = pinv(A)*y  (:9,:17)
    __syncthreads();
    // residual (:11,:18) and the two norms of the stopping rule (:20)
    double dn = 0, pn = 0;
    for (int p = tid; p < np; p += PUR_THREADS) {
      c64 acc = yv[p];
      for (int j = 0; j <= it; ++j) {
        const cx<T> a = S[(size_t)picked[j] * np + p];
        acc = acc - c64{(double)a.x, (double)a.y} * xv[j];
      }
      const c64 d = acc - r[p];
      dn += norm2(d);
      pn += norm2(r[p]);
      r[p] = acc;
    }
    for (int off = 32; off > 0; off >>= 1) { dn += __shfl_down(dn, off, 64); pn += __shfl_down(pn, off, 64); }
    if (lane == 0) { nrm[wid] = dn; nrm[PUR_WAVES + wid] = pn; }
    __syncthreads();
    if (tid == 0) {
      double a = 0, b = 0;
      for (int w = 0; w < PUR_WAVES; ++w) { a += nrm[w]; b += nrm[PUR_WAVES + w]; }
      n_s = it + 1;
      if (it >= 1 && (sqrt(a) / sqrt(b) < 1e-2)) stop_s = 1;          // :20-22 (only inside the i1>=2 loop)
    }
    __syncthreads();
    if (stop_s) break;
  }
  __syncthreads();
  const int n = n_s;
  if (tid == 0) {
    for (int j = 0; j < n; ++j) {
      h_out[picked[j]] = mk<T>((T)xv[j].x, (T)xv[j].y);               // :31-33 (last write wins)
      if (index_out) index_out[j] = picked[j] + 1;
    }
    if (n_index_out) n_index_out[0] = n;
  }
}

static int upload_zero_based(Stage& st, const int32_t* idx1, int n, int limit, const char* what, const void** d) {
  std::vector<int32_t> v(n);
  for (int i = 0; i < n; ++i) {
    OFDM_ARG(idx1[i] >= 1 && idx1[i] <= limit, "%s: index %d outside 1..%d", what, (int)idx1[i], limit);
    v[i] = idx1[i] - 1;
  }
  return st.upload(v.data(), sizeof(int32_t) * n, d);
}

// applies a host-built real operator W (column-major [n_out x n_in], double) to a device vector
static int apply_operator(Stage& st, const std::vector<double>& W, int n_out, int n_in, const void* dv, void* dout,
                          bool f64) {
  const void* dW;
  if (f64) {
    OFDM_TRY(st.upload(W.data(), sizeof(double) * W.size(), &dW));
    hipLaunchKernelGGL(apply_operator_kernel<double>, dim3(cdiv_u(n_out, 128)), dim3(128), 0, ctx().stream,
                       (const double*)dW, (const c64*)dv, (c64*)dout, n_out, n_in);
  } else {
    std::vector<float> Wf(W.begin(), W.end());
    OFDM_TRY(st.upload(Wf.data(), sizeof(float) * Wf.size(), &dW));
    hipLaunchKernelGGL(apply_operator_kernel<float>, dim3(cdiv_u(n_out, 128)), dim3(128), 0, ctx().stream,
                       (const float*)dW, (const c32*)dv, (c32*)dout, n_out, n_in);
  }
  return check_launch("apply_operator_kernel");
}

}  // namespace ofdm

using namespace ofdm;

extern "C" {

int ofdm_interpolate(const void* h, const int32_t* pilot_loc, int n_pilots, int n_out, char method, void* out,
                     int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(n_pilots >= 2 && n_out >= 1, "interpolate: needs at least two pilots");
  std::vector<double> W;
  OFDM_TRY(build_interpolate_operator(pilot_loc, n_pilots, n_out, method, W));
  Stage st(flags);
  const void* dh; void* dout;
  OFDM_TRY(st.in(h, csize(flags) * (size_t)n_pilots, &dh));
  OFDM_TRY(st.out(out, csize(flags) * (size_t)n_out, &dout));
  OFDM_TRY(apply_operator(st, W, n_out, n_pilots, dh, dout, is_f64(flags)));
  return st.finish();
}

int ofdm_estimate_channel(const void* rx, int nfft, int64_t n_symb, const int32_t* all_carriers, int n_all,
                          const int32_t* pilot_carriers, int n_pilots, const void* pilot_values, void* h_est_out,
                          void* h_pilots_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(nfft > 0 && n_symb > 0 && n_pilots >= 2 && n_all >= 1, "estimate_channel: bad sizes");
  std::vector<double> xk(n_pilots), xq(n_all), W;
  for (int i = 0; i < n_pilots; ++i) xk[i] = pilot_carriers[i];
  for (int i = 0; i < n_all; ++i) xq[i] = all_carriers[i];
  OFDM_TRY(build_spline_operator(xk, xq, W));                                   // interp1(...,'spline') (:8)
  const bool f64 = is_f64(flags);
  Stage st(flags);
  const void *dx, *dtx, *dpc; void *dh, *dhp;
  OFDM_TRY(st.in(rx, csize(flags) * (size_t)nfft * n_symb, &dx));
  OFDM_TRY(st.in(pilot_values, csize(flags) * (size_t)n_pilots * n_symb, &dtx));
  OFDM_TRY(upload_zero_based(st, pilot_carriers, n_pilots, nfft, "estimate_channel(pilotCarriers)", &dpc));
  OFDM_TRY(st.out(h_est_out, csize(flags) * (size_t)n_all, &dh));
  if (h_pilots_out) OFDM_TRY(st.out(h_pilots_out, csize(flags) * (size_t)n_pilots, &dhp));
  else OFDM_TRY(st.scratch(csize(flags) * (size_t)n_pilots, &dhp));
  if (f64) hipLaunchKernelGGL(mean_pilots_kernel<double>, dim3(cdiv_u(n_pilots, 128)), dim3(128), 0, ctx().stream, (const c64*)dx, (const c64*)dtx, (const int32_t*)dpc, (c64*)dhp, nfft, n_pilots, n_symb);
  else hipLaunchKernelGGL(mean_pilots_kernel<float>, dim3(cdiv_u(n_pilots, 128)), dim3(128), 0, ctx().stream, (const c32*)dx, (const c32*)dtx, (const int32_t*)dpc, (c32*)dhp, nfft, n_pilots, n_symb);
  OFDM_TRY(check_launch("mean_pilots_kernel"));
  OFDM_TRY(apply_operator(st, W, n_all, n_pilots, dhp, dh, f64));
  return st.finish();
}

int ofdm_LS_CE(const void* y, int nfft, int64_t n_symb, const void* xp, const int32_t* pilot_loc, int n_pilots,
               int n_carrier, void* h_ls_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(nfft > 0 && n_symb > 0 && n_pilots >= 2 && n_carrier >= 1, "LS_CE: bad sizes");
  std::vector<double> W;
  OFDM_TRY(build_interpolate_operator(pilot_loc, n_pilots, n_carrier, 's', W));   // LS_CE.m:31
  const bool f64 = is_f64(flags);
  Stage st(flags);
  const void *dy, *dxp, *dpc; void *dls, *dout;
  // linear indexing Y(pilot_loc(k)) may legally reach beyond column 1 when pilot_loc > Nfft
  OFDM_TRY(st.in(y, csize(flags) * (size_t)nfft * n_symb, &dy));
  OFDM_TRY(st.in(xp, csize(flags) * (size_t)n_pilots * n_symb, &dxp));
  OFDM_TRY(upload_zero_based(st, pilot_loc, n_pilots, (int)std::min<int64_t>((int64_t)nfft * n_symb, INT32_MAX), "LS_CE(pilot_loc)", &dpc));
  OFDM_TRY(st.scratch(csize(flags) * (size_t)n_pilots, &dls));
  OFDM_TRY(st.out(h_ls_out, csize(flags) * (size_t)n_carrier, &dout));
  if (f64) hipLaunchKernelGGL(ls_pilots_kernel<double>, dim3(cdiv_u(n_pilots, 128)), dim3(128), 0, ctx().stream, (const c64*)dy, (const c64*)dxp, (const int32_t*)dpc, (c64*)dls, n_pilots);
  else hipLaunchKernelGGL(ls_pilots_kernel<float>, dim3(cdiv_u(n_pilots, 128)), dim3(128), 0, ctx().stream, (const c32*)dy, (const c32*)dxp, (const int32_t*)dpc, (c32*)dls, n_pilots);
  OFDM_TRY(check_launch("ls_pilots_kernel"));
  OFDM_TRY(apply_operator(st, W, n_carrier, n_pilots, dls, dout, f64));
  return st.finish();
}

int ofdm_MMSE_CE(const void* y, int nfft, int64_t n_symb, const void* xp, const int32_t* pilot_loc, int n_pilots,
                 int n_carrier, const void* h, int h_len, double snr_db, void* h_mmse_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(nfft > 0 && n_symb > 0 && n_pilots >= 2 && n_carrier >= 1 && h_len >= 1, "MMSE_CE: bad sizes");
  OFDM_ARG(n_pilots <= MMSE_MAXNP, "MMSE_CE: more than %d pilots not supported", MMSE_MAXNP);
  std::vector<double> W;
  OFDM_TRY(build_interpolate_operator(pilot_loc, n_pilots, n_carrier, 's', W));   // MMSE_CE.m:38
  const bool f64 = is_f64(flags);
  const double snr = std::pow(10.0, snr_db * 0.1);                                // :13
  const double nps = (double)pilot_loc[1] - (double)pilot_loc[0];                 // :15
  const double df = 1.0 / (double)n_carrier;                                      // :25
  Stage st(flags);
  const void *dy, *dxp, *dpc, *dh; void *dls, *dhp, *dout;
  OFDM_TRY(st.in(y, csize(flags) * (size_t)nfft * n_symb, &dy));
  OFDM_TRY(st.in(xp, csize(flags) * (size_t)n_pilots * n_symb, &dxp));
  OFDM_TRY(st.in(h, csize(flags) * (size_t)h_len, &dh));
  OFDM_TRY(upload_zero_based(st, pilot_loc, n_pilots, nfft, "MMSE_CE(pilot_loc)", &dpc));
  OFDM_TRY(st.scratch(csize(flags) * (size_t)n_pilots, &dls));
  OFDM_TRY(st.scratch(csize(flags) * (size_t)n_pilots, &dhp));
  OFDM_TRY(st.out(h_mmse_out, csize(flags) * (size_t)n_carrier, &dout));
  if (f64) {
    hipLaunchKernelGGL(ls_pilots_kernel<double>, dim3(cdiv_u(n_pilots, 128)), dim3(128), 0, ctx().stream, (const c64*)dy, (const c64*)dxp, (const int32_t*)dpc, (c64*)dls, n_pilots);   // :17
    hipLaunchKernelGGL(mmse_kernel<double>, dim3(1), dim3(MMSE_THREADS), 0, ctx().stream, (const c64*)dh, h_len, (const c64*)dls, n_pilots, nps, df, 1.0 / snr, (c64*)dhp, (double*)nullptr);
  } else {
    hipLaunchKernelGGL(ls_pilots_kernel<float>, dim3(cdiv_u(n_pilots, 128)), dim3(128), 0, ctx().stream, (const c32*)dy, (const c32*)dxp, (const int32_t*)dpc, (c32*)dls, n_pilots);
    hipLaunchKernelGGL(mmse_kernel<float>, dim3(1), dim3(MMSE_THREADS), 0, ctx().stream, (const c32*)dh, h_len, (const c32*)dls, n_pilots, nps, df, 1.0 / snr, (c32*)dhp, (double*)nullptr);
  }
  OFDM_TRY(check_launch("mmse_kernel"));
  OFDM_TRY(apply_operator(st, W, n_carrier, n_pilots, dhp, dout, f64));
  return st.finish();
}

int ofdm_sensing_matrix(const int32_t* pilot_carriers, int n_pilots, int nfft, int k, void* s_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(n_pilots >= 1 && nfft >= 1 && k >= 1, "sensing_matrix: bad sizes");
  Stage st(flags);
  const void* dpc; void* ds;
  OFDM_TRY(upload_zero_based(st, pilot_carriers, n_pilots, nfft, "sensing_matrix(pilotCarriers)", &dpc));
  OFDM_TRY(st.out(s_out, csize(flags) * (size_t)n_pilots * k, &ds));
  const int64_t total = (int64_t)n_pilots * k;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
  if (is_f64(flags)) hipLaunchKernelGGL(sensing_kernel<double>, dim3(grid), dim3(256), 0, ctx().stream, (const int32_t*)dpc, n_pilots, nfft, k, (c64*)ds);
  else hipLaunchKernelGGL(sensing_kernel<float>, dim3(grid), dim3(256), 0, ctx().stream, (const int32_t*)dpc, n_pilots, nfft, k, (c32*)ds);
  OFDM_TRY(check_launch("sensing_kernel"));
  return st.finish();
}

static int pursuit_common(bool omp, const void* y, const void* s, int n_pilots, int k, int nfft, int taps,
                          void* H_out, void* h_out, int32_t* picks_out, int* n_index_out, int flags) {
  OFDM_TRY(ensure_init());
  const char* who = omp ? "OMP_estimate" : "MP_estimate";
  OFDM_ARG(n_pilots >= 1 && k >= 1 && taps >= 1, "%s: bad sizes", who);
  OFDM_ARG(n_pilots <= PUR_MAXNP && taps <= PUR_MAXT, "%s: supports up to %d pilots and %d taps", who, PUR_MAXNP, PUR_MAXT);
  OFDM_ARG(fft_size_supported(nfft), "%s: unsupported Nfft %d", who, nfft);
  if (!omp) OFDM_ARG(k >= n_pilots, "MP_estimate: index exceeds matrix dimensions (loop runs over Np=%d columns, K=%d)", n_pilots, k);
  if (!omp) OFDM_ARG(taps <= n_pilots, "MP_estimate: more taps than candidate columns");
  OFDM_ARG(std::max(k, omp ? k : n_pilots) <= nfft, "%s: atom index exceeds Nfft", who);
  const bool f64 = is_f64(flags);
  Stage st(flags);
  const void *dy, *ds; void *dH, *dh, *dpicks, *dn;
  OFDM_TRY(st.in(y, csize(flags) * (size_t)n_pilots, &dy));
  OFDM_TRY(st.in(s, csize(flags) * (size_t)n_pilots * k, &ds));
  if (h_out) OFDM_TRY(st.out(h_out, csize(flags) * (size_t)nfft, &dh));
  else OFDM_TRY(st.scratch(csize(flags) * (size_t)nfft, &dh));
  OFDM_TRY(st.out(H_out, csize(flags) * (size_t)nfft, &dH));
  std::vector<int32_t> picks(taps, 0);
  int32_t n_idx = 0;
  OFDM_TRY(st.fetch(picks_out ? picks.data() : nullptr, sizeof(int32_t) * taps, &dpicks));
  OFDM_TRY(st.fetch(n_index_out ? &n_idx : nullptr, sizeof(int32_t), &dn));
  OFDM_HIP(hipMemsetAsync(dpicks, 0, sizeof(int32_t) * taps, ctx().stream));
  if (omp) {
    if (f64) hipLaunchKernelGGL(omp_kernel<double>, dim3(1), dim3(PUR_THREADS), 0, ctx().stream, (const c64*)dy, (const c64*)ds, n_pilots, k, taps, nfft, (c64*)dh, (int32_t*)dpicks, (int32_t*)dn);
    else hipLaunchKernelGGL(omp_kernel<float>, dim3(1), dim3(PUR_THREADS), 0, ctx().stream, (const c32*)dy, (const c32*)ds, n_pilots, k, taps, nfft, (c32*)dh, (int32_t*)dpicks, (int32_t*)dn);
  } else {
    if (f64) hipLaunchKernelGGL(mp_kernel<double>, dim3(1), dim3(PUR_THREADS), 0, ctx().stream, (const c64*)dy, (const c64*)ds, n_pilots, k, taps, nfft, (c64*)dh, (int32_t*)dpicks);
    else hipLaunchKernelGGL(mp_kernel<float>, dim3(1), dim3(PUR_THREADS), 0, ctx().stream, (const c32*)dy, (const c32*)ds, n_pilots, k, taps, nfft, (c32*)dh, (int32_t*)dpicks);
  }
  OFDM_TRY(check_launch(who));
  OFDM_TRY(demod_device(dh, dH, nfft, 1, 0, f64));          // fft(h_impulse_est)  (MP :33 / OMP :36)
  OFDM_TRY(st.finish());
  if (picks_out) memcpy(picks_out, picks.data(), sizeof(int32_t) * taps);
  if (n_index_out) *n_index_out = n_idx;
  return OFDM_OK;
}

int ofdm_MP_estimate(const void* y, const void* s, int n_pilots, int k, int nfft, int dominant_taps, void* H_out,
                     void* h_out, int32_t* picks_out, int flags) {
  return pursuit_common(false, y, s, n_pilots, k, nfft, dominant_taps, H_out, h_out, picks_out, nullptr, flags);
}

int ofdm_OMP_estimate(const void* y, const void* s, int n_pilots, int k, int nfft, int dominant_taps, double snr_db,
                      void* H_out, void* h_out, int32_t* index_out, int* n_index_out, int flags) {
  (void)snr_db;   // OMP_estimate.m:4 computes noise_pw and never uses it
  return pursuit_common(true, y, s, n_pilots, k, nfft, dominant_taps, H_out, h_out, index_out, n_index_out, flags);
}

}  // extern "C"
