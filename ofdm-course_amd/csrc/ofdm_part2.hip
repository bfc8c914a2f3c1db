// One (kk, jj-range) tile of Task 5/Task5_part2.m as a device-resident pass: the Monte-Carlo realisations jj of one pilot
// scenario kk share the noisy TX stream and differ in their channel (Task5_part2.m:148-205, :269-304).  Per realisation
//   conv(Tx_noised, h_jj) truncated                (:160-166)      fir_multi_kernel
//   OFDM_demodulator, rows 1..N_carrier            (:169-172)      demod_keep_device
//   Y = X(pilots, 1) ./ pilotValues(:, 1)          (:190)          p2_pilot_ls_kernel
//   LS_CE                                          (:174)          spline operator (interpolate.m) as ONE real GEMM over the tile
//   MMSE_CE(h = true CIR, SNR)                     (:176-177)      mmse_wave_kernel: one wavefront per realisation (Levinson),
//                                                                  then the same spline operator
//   MP_estimate                                    (:192)          mp_batch_kernel: S^H residue on the matrix cores per iteration
//   OMP_estimate                                   (:193)          omp_batch_kernel (ofdm_chain_fast.hip)
//   NMSE of the four estimates                     (:202-205)      p2_nmse_kernel against H = fft(h_jj)
//   equalize_signal -> get_payload -> demapping -> BER_func, four times (:269-304)   eq_demap_kernel (ofdm_chain_split.hip)
// Nothing returns to the host but 4 x n NMSE values and 4 x n error counts.
#include <algorithm>

#include "rx_plan.hpp"
#include "spline_op.hpp"

namespace ofdm {

int demod_keep_device(const void* y, void* x, int nfft, int64_t n_symb, int t_guard, int n_keep, bool f64);   // ofdm_modem.hip
bool chain_split_supported(int nfft, int n_carrier, int taps, int bps, int64_t nd_nsymb, bool f64);           // ofdm_chain_split.hip

// ---- conv(x, h_f.', 'full')(1:len) for every realisation f: sparse taps (delay, amplitude) per realisation
template <typename T>
__global__ void fir_multi_kernel(const cx<T>* __restrict__ x, int64_t len, const int32_t* __restrict__ delay,
                                 const c64* __restrict__ amp, int n_taps, cx<T>* __restrict__ out) {
  const int64_t f = blockIdx.y;
  const int32_t* d = delay + f * n_taps;
  const c64* a = amp + f * n_taps;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < len; n += (int64_t)gridDim.x * blockDim.x) {
    double sr = 0, si = 0;
    for (int t = 0; t < n_taps; ++t) {
      const int64_t m = n - d[t];
      if (m >= 0) {
        const cx<T> v = x[m];
        sr += a[t].x * (double)v.x - a[t].y * (double)v.y;
        si += a[t].x * (double)v.y + a[t].y * (double)v.x;
      }
    }
    out[f * len + n] = mk<T>((T)sr, (T)si);
  }
}

template <typename T>
__global__ void p2_pilot_ls_kernel(const cx<T>* __restrict__ xk, const int32_t* __restrict__ pc0, const cx<T>* __restrict__ pilots,
                                   cx<T>* __restrict__ ypil, int np, int n_symb, int nc, int64_t n_frames) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_frames * np) return;
  const int64_t f = i / np;
  const int p = (int)(i - f * np);
  ypil[i] = cdiv(xk[f * n_symb * (int64_t)nc + pc0[p]], pilots[p]);
}

template <typename T>
__global__ void p2_replicate_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int words, int64_t n_frames) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_frames * words) dst[i] = src[i % words];
}

// ---- out[f][m] = sum_j W[m, j] v[f][j]: real operator [n_out x n_in] (column-major), P2_FT realisations per workgroup
constexpr int P2_FT = 8;
template <typename T>
__global__ __launch_bounds__(128) void p2_apply_operator_kernel(const double* __restrict__ W, const cx<T>* __restrict__ v,
                                                                cx<T>* __restrict__ out, int n_out, int n_in, int64_t n_frames) {
  extern __shared__ __attribute__((aligned(16))) unsigned char p2_smem[];
  cx<T>* hs = (cx<T>*)p2_smem;                                     // [n_in][P2_FT]
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.y * P2_FT;
  for (int i = threadIdx.x; i < n_in * P2_FT; i += blockDim.x) {
    const int f = i / n_in, j = i - f * n_in;
    hs[j * P2_FT + f] = f0 + f < n_frames ? v[(f0 + f) * n_in + j] : mk<T>(0, 0);
  }
  __syncthreads();
  if (m >= n_out) return;
  double ar[P2_FT], ai[P2_FT];
#pragma unroll
  for (int f = 0; f < P2_FT; ++f) ar[f] = ai[f] = 0.0;
  for (int j0 = 0; j0 < n_in; j0 += 8) {
    double w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = j0 + u < n_in ? W[(size_t)(j0 + u) * n_out + m] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + u < n_in) {
#pragma unroll
        for (int f = 0; f < P2_FT; ++f) {
          const cx<T> z = hs[(j0 + u) * P2_FT + f];
          ar[f] += w[u] * (double)z.x;
          ai[f] += w[u] * (double)z.y;
        }
      }
  }
#pragma unroll
  for (int f = 0; f < P2_FT; ++f)
    if (f0 + f < n_frames) out[(f0 + f) * n_out + m] = mk<T>((T)ar[f], (T)ai[f]);
}

// ---- MMSE_CE.m:25-36 for one realisation per WAVEFRONT: Rpp = rf2 + I/snr is Hermitian Toeplitz with first column
// 1/(1 + j c k), c = 2 pi tau_rms df Nps (tau_rms of the realisation's CIR comes with it); z = Rpp \ H_tilde by the Levinson
// recursion in double, all state in wave-private LDS, no workgroup barrier; out = rf2 * z (the first Np rows of Rhp/Rpp*H_tilde,
// which is all MMSE_CE.m:38 keeps).
__device__ __forceinline__ double p2_wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void mmse_wave_kernel(const cx<T>* __restrict__ ypil, const double* __restrict__ cvals,
                                                        double inv_snr, int np, cx<T>* __restrict__ vout, int64_t n_frames,
                                                        const double* __restrict__ inv_snr_v = nullptr) {
  // blockDim.x / 64 realisations per workgroup (4 while their state fits the LDS, fewer for many pilots);
  // inv_snr_v (optional): every realisation at its own SNR (the SNR sweep of Main_model_Task_5.m:305-315)
  extern __shared__ __attribute__((aligned(16))) unsigned char p2_smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t f = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (f >= n_frames) return;                                       // no workgroup barrier below
  if (inv_snr_v) inv_snr = inv_snr_v[f];
  c64* tcol = (c64*)p2_smem + (size_t)wave * 4 * np;               // [np] each: tcol, fv, bv, xv
  c64 *fv = tcol + np, *bv = fv + np, *xv = bv + np;
  const cx<T>* y = ypil + f * np;
  const double c = cvals[f];
  for (int k = lane; k < np; k += 64) {
    const double d = 1.0 + (c * k) * (c * k);
    tcol[k] = c64{1.0 / d, -(c * k) / d};
  }
  wave_sync();
  const double t0 = tcol[0].x + inv_snr;
  if (lane == 0) {
    fv[0] = c64{1.0 / t0, 0};
    bv[0] = c64{1.0 / t0, 0};
    xv[0] = c64{(double)y[0].x / t0, (double)y[0].y / t0};
  }
  wave_sync();
  for (int n = 1; n < np; ++n) {
    // eps_f = sum_i T[n][i] f[i], eps_x = sum_i T[n][i] x[i], eps_b = sum_i T[0][i+1] b[i],  i < n   (T[i][j] = t(i - j))
    c64 ef{0, 0}, ex{0, 0}, eb{0, 0};
    for (int i = lane; i < n; i += 64) {
      const c64 tn = tcol[n - i];
      ef = ef + tn * fv[i];
      ex = ex + tn * xv[i];
      eb = eb + conj(tcol[i + 1]) * bv[i];
    }
    ef = c64{p2_wave_sum(ef.x), p2_wave_sum(ef.y)};
    ex = c64{p2_wave_sum(ex.x), p2_wave_sum(ex.y)};
    eb = c64{p2_wave_sum(eb.x), p2_wave_sum(eb.y)};
    const c64 one{1, 0};
    const c64 inv = cdiv(one, one - eb * ef);
    const c64 dx = c64{(double)y[n].x, (double)y[n].y} - ex;
    // new f = inv [f; 0] - ef inv [0; b] ; new b = inv [0; b] - eb inv [f; 0] ; x += dx * new b.  Entry i reads the old f[i]
    // and b[i - 1] and writes index i: 64-entry chunks from the top down, each chunk reading before it writes, never
    // overwrite an input of a chunk still to come.
    for (int i0 = (n / 64) * 64; i0 >= 0; i0 -= 64) {
      const int i = i0 + lane;
      c64 nf{0, 0}, nb{0, 0}, nx{0, 0};
      if (i <= n) {
        const c64 fe = (i < n) ? fv[i] : c64{0, 0};
        const c64 be = (i > 0) ? bv[i - 1] : c64{0, 0};
        nf = inv * fe - (ef * inv) * be;
        nb = inv * be - (eb * inv) * fe;
        nx = ((i < n) ? xv[i] : c64{0, 0}) + dx * nb;
      }
      wave_sync();
      if (i <= n) {
        fv[i] = nf;
        bv[i] = nb;
        xv[i] = nx;
      }
      wave_sync();
    }
  }
  // ---- out = rf2 * z (no 1/snr on this diagonal)
  for (int i = lane; i < np; i += 64) {
    c64 acc{0, 0};
    for (int j = 0; j < np; ++j) {
      const int k = i - j;
      const c64 t = k >= 0 ? tcol[k] : conj(tcol[-k]);
      acc = acc + t * xv[j];
    }
    vout[f * np + i] = mk<T>((T)acc.x, (T)acc.y);
  }
}

// ---- MP_estimate.m:8-24 for a tile of realisations.  A workgroup of four wavefronts owns FB = 4 * fpw realisations:
//   per iteration  (1) C = S(:, 1:Np)^H * residue for all FB at once: fp32 on the matrix cores (v_mfma_f32_16x16x4_f32 through
//                      corr_mfma_f32; this is the dense product a random pilot mask needs, Task5_part2.m:58-64), else scalar
//                  (2) per realisation (a group of 64 / fpw lanes): projection |c|^2 / norm(a)^2 over the FIRST Np columns
//                      (:10), picked columns at -100 (:11-12), first maximum (:18), x = c(kp) / norm(a)^2 (:22),
//                      residue -= a_kp x (:21, :23)
// The picks / coefficients land in the plan's tap workspace, where the equalise stage turns them into H = fft(h) (:27-33).
struct MpLayout { unsigned off_r, off_c, off_pick, total; int fpw; };

template <typename T>
static MpLayout mp_layout(int np, int kc, int taps) {
  MpLayout o;
  int fpw = 4;
  auto bytes = [&](int f) { return sizeof(cx<T>) * (size_t)(4 * f) * (np + 1 + kc) + sizeof(int) * (size_t)(4 * f) * taps; };
  while (fpw > 1 && bytes(fpw) > 72 * 1024) fpw >>= 1;
  o.fpw = fpw;
  const int fb = 4 * fpw;
  unsigned b = 0;
  o.off_r = b;    b += (unsigned)((sizeof(cx<T>) * fb * (np + 1) + 15) & ~15u);
  o.off_c = b;    b += (unsigned)((sizeof(cx<T>) * fb * kc + 15) & ~15u);
  o.off_pick = b; b += (unsigned)((sizeof(int) * fb * taps + 15) & ~15u);
  o.total = b;
  return o;
}

template <typename T, bool MFMA>
__global__ __launch_bounds__(256) void mp_batch_kernel(FastParams<T> P, MpLayout lay, int kc, int64_t n_frames) {
  extern __shared__ __attribute__((aligned(16))) unsigned char p2_smem[];
  cx<T>* Rl = (cx<T>*)(p2_smem + lay.off_r);      // [FB][np + 1] residue
  cx<T>* Cl = (cx<T>*)(p2_smem + lay.off_c);      // [FB][kc]     S^H residue
  int* picks = (int*)(p2_smem + lay.off_pick);    // [FB][taps]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int FPW = lay.fpw, FB = 4 * FPW;
  const int64_t f0 = (int64_t)blockIdx.x * FB;
  const int np = P.np, K = P.k_atoms, taps = P.taps, YS = np + 1;
  for (int i = tid; i < FB * np; i += 256) {
    const int f = i / np, p = i - f * np;
    Rl[f * YS + p] = (f0 + f < n_frames) ? P.ypil[(f0 + f) * np + p] : mk<T>(0, 0);
  }
  const int LPF = 64 / FPW, grp = lane / LPF, sl = lane - grp * LPF;
  const int fi = wave * FPW + grp;
  const int64_t f = f0 + fi;
  const bool live = f < n_frames;
  const double nrm2 = (double)np;                 // norm(a)^2 of a column of exp(-2 pi i ..): Np
  __syncthreads();
  for (int it = 0; it < taps; ++it) {
    // (1) correlations of every realisation of the workgroup
    if constexpr (MFMA) {
      corr_mfma_f32(P.sct, K, kc / 16, Rl, YS, np, FB, (float*)Cl, kc, wave, lane);
    } else {
      for (int i = tid; i < FB * kc; i += 256) {
        const int ff = i / kc, k = i - ff * kc;
        double ar = 0, ai = 0;
        for (int p = 0; p < np; ++p) {
          const cx<T> a = P.sct[(size_t)p * K + k], r = Rl[ff * YS + p];
          ar += (double)a.x * r.x - (double)a.y * r.y;
          ai += (double)a.x * r.y + (double)a.y * r.x;
        }
        Cl[i] = mk<T>((T)ar, (T)ai);
      }
    }
    __syncthreads();
    // (2) projection, first maximum over the group's lanes, update
    const cx<T>* cf = Cl + fi * kc;
    double bs = -1e300;
    int bi = 0x7fffffff;
    for (int k = sl; k < kc; k += LPF) {
      double sc = ((double)cf[k].x * cf[k].x + (double)cf[k].y * cf[k].y) / nrm2;
      for (int q = 0; q < it; ++q) if (picks[fi * taps + q] == k) sc = -100.0;
      if (sc > bs) { bs = sc; bi = k; }           // ascending k inside a lane: strict > keeps the first
    }
    double gm = bs;
    for (int off = LPF >> 1; off > 0; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off, 64));
    int kp = bs == gm ? bi : 0x7fffffff;
    for (int off = LPF >> 1; off > 0; off >>= 1) kp = min(kp, __shfl_xor(kp, off, 64));
    if (kp >= kc) kp = 0;                         // all-NaN projections: MATLAB max returns index 1
    const cx<T> ck = cf[kp];
    const c64 x{(double)ck.x / nrm2, (double)ck.y / nrm2};
    cx<T>* rf = Rl + fi * YS;
    for (int p = sl; p < np; p += LPF) {
      const cx<T> a = conj(P.sct[(size_t)p * K + kp]);             // S(p, kp)
      const double rr = (double)rf[p].x - ((double)a.x * x.x - (double)a.y * x.y);
      const double ri = (double)rf[p].y - ((double)a.x * x.y + (double)a.y * x.x);
      rf[p] = mk<T>((T)rr, (T)ri);
    }
    if (sl == 0) {
      picks[fi * taps + it] = kp;
      if (live) {
        P.tap_idx[f * taps + it] = kp;
        P.tap_x[f * taps + it] = x;
        // h_impulse_est(kp(i1)) = x(i1) (:28-30): a later pick of the same column (possible once all Np columns are used
        // up: every projection is -100 and max returns column 1) overwrites the earlier coefficient
        for (int q = 0; q < it; ++q)
          if (picks[fi * taps + q] == kp) P.tap_x[f * taps + q] = c64{0, 0};
      }
    }
    __syncthreads();
  }
}

// ---- (H_f - H_est)(H_f - H_est)' / N_carrier for the four estimates (:202-205); H_f = fft(h_jj)(1..N_carrier) from the taps
template <typename T>
__global__ __launch_bounds__(256) void p2_nmse_kernel(const int32_t* __restrict__ delay, const c64* __restrict__ amp, int n_taps,
                                                      int nfft, int nc, const cx<T>* __restrict__ h0, const cx<T>* __restrict__ h1,
                                                      const cx<T>* __restrict__ h2, const cx<T>* __restrict__ h3,
                                                      double* __restrict__ nmse /* [4][n_frames] */, int64_t n_frames) {
  const int64_t f = blockIdx.x;
  __shared__ double red[4][4];
  double acc[4] = {0, 0, 0, 0};
  const cx<T>* hs[4] = {h0, h1, h2, h3};
  for (int k = threadIdx.x; k < nc; k += 256) {
    double hr = 0, hi = 0;
    for (int t = 0; t < n_taps; ++t) {
      const int e = (int)(((int64_t)delay[f * n_taps + t] * k) % nfft);
      double sn, cs;
      sincospi(2.0 * (double)e / (double)nfft, &sn, &cs);
      const c64 a = amp[f * n_taps + t];
      hr += a.x * cs + a.y * sn;                                   // a * exp(-2 pi i e / N)
      hi += a.y * cs - a.x * sn;
    }
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const cx<T> he = hs[e4][f * nc + k];
      const double dr = hr - (double)he.x, di = hi - (double)he.y;
      acc[e4] += dr * dr + di * di;
    }
  }
#pragma unroll
  for (int e4 = 0; e4 < 4; ++e4) {
    double v = acc[e4];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[e4][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int e4 = threadIdx.x;
    nmse[e4 * n_frames + f] = (red[e4][0] + red[e4][1] + red[e4][2] + red[e4][3]) / (double)nc;
  }
}

template <typename T>
static int part2_tile_run(ofdm_rx_plan* pl, const void* dtx, const int32_t* ddelay, const c64* damp, int n_ch, int64_t F,
                          const double* dcvals, double inv_snr, const void* dref1, double* dnmse, uint32_t* derrs) {
  const int N = pl->nfft, Tg = pl->t_guard, S = pl->n_symb, np = pl->np, nc = pl->n_carrier, taps = pl->taps;
  const int64_t len = (int64_t)(N + Tg) * S;
  const bool f64 = std::is_same<T, double>::value;
  hipStream_t st = ctx().stream;
  // arena
  size_t need = 0;
  auto reserve = [&](size_t bytes) { const size_t o = need; need += (bytes + 255) & ~size_t(255); return o; };
  const size_t o_rx = reserve(sizeof(cx<T>) * (size_t)len * F), o_x = reserve(sizeof(cx<T>) * (size_t)nc * S * F);
  const size_t o_v = reserve(sizeof(cx<T>) * (size_t)np * F), o_ref = reserve((size_t)pl->frame_words * 4 * F);
  size_t o_h[4];
  for (int e = 0; e < 4; ++e) o_h[e] = reserve(sizeof(cx<T>) * (size_t)nc * F);
  if (pl->ws_t4_bytes < need) {                                     // the Task-4 arena doubles as this entry's
    OFDM_HIP(hipStreamSynchronize(st));
    if (pl->ws_t4) { (void)hipFree(pl->ws_t4); pl->ws_t4 = nullptr; pl->ws_t4_bytes = 0; }
    OFDM_HIP(hipMalloc(&pl->ws_t4, need));
    pl->ws_t4_bytes = need;
  }
  unsigned char* arena = (unsigned char*)pl->ws_t4;
  cx<T>* drx = (cx<T>*)(arena + o_rx);
  cx<T>* dxk = (cx<T>*)(arena + o_x);
  cx<T>* dv = (cx<T>*)(arena + o_v);
  uint32_t* dref = (uint32_t*)(arena + o_ref);
  cx<T>* dh[4];
  for (int e = 0; e < 4; ++e) dh[e] = (cx<T>*)(arena + o_h[e]);
  // spline operator of interpolate.m (LS_CE.m:31, MMSE_CE.m:38): built once per plan, double
  if (!pl->d_p2_sop) {
    std::vector<double> W;
    OFDM_TRY(build_interpolate_operator(pl->pilot_loc.data(), np, nc, 's', W));
    OFDM_HIP(hipMalloc(&pl->d_p2_sop, sizeof(double) * W.size()));
    OFDM_HIP(hipMemcpy(pl->d_p2_sop, W.data(), sizeof(double) * W.size(), hipMemcpyHostToDevice));
  }
  FastPlanView pv;
  make_plan_view(pl, pv);
  pv.ev = nullptr;
  pv.d_wt = nullptr;
  FastParams<T> P;
  const void* tw = nullptr;
  OFDM_TRY(get_twiddles(N, f64, &tw));
  OFDM_TRY(fast_params_prepare<T>(pv, tw, F, P));
  // channel, demodulator, pilot LS values
  hipLaunchKernelGGL(fir_multi_kernel<T>, dim3((unsigned)std::min<int64_t>((len + 255) / 256, 256), (unsigned)F), dim3(256), 0, st,
                     (const cx<T>*)dtx, len, ddelay, damp, n_ch, drx);
  OFDM_TRY(check_launch("fir_multi_kernel"));
  OFDM_TRY(demod_keep_device(drx, dxk, N, F * S, Tg, nc, f64));
  hipLaunchKernelGGL(p2_pilot_ls_kernel<T>, dim3(cdiv_u(F * np, 256)), dim3(256), 0, st, (const cx<T>*)dxk, (const int32_t*)pl->d_pc0,
                     (const cx<T>*)pl->d_pilots, P.ypil, np, S, nc, F);
  hipLaunchKernelGGL(p2_replicate_kernel<T>, dim3(cdiv_u(F * pl->frame_words, 256)), dim3(256), 0, st, (const uint32_t*)dref1, dref,
                     pl->frame_words, F);
  // LS: H = Sop * Y
  const dim3 og(cdiv_u(nc, 128), cdiv_u(F, P2_FT));
  hipLaunchKernelGGL(p2_apply_operator_kernel<T>, og, dim3(128), sizeof(cx<T>) * np * P2_FT, st, (const double*)pl->d_p2_sop,
                     (const cx<T>*)P.ypil, dh[0], nc, np, F);
  // MMSE: v = rf2 (rf2 + I/snr)^-1 Y per realisation, H = Sop * v
  {
    const size_t dyn = sizeof(c64) * 4 * (size_t)np * 4;
    OFDM_ARG(dyn <= 150 * 1024, "task5_part2_tile: MMSE stage supports at most 585 pilots");
    OFDM_HIP(hipFuncSetAttribute((const void*)mmse_wave_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(mmse_wave_kernel<T>, dim3(cdiv_u(F, 4)), dim3(256), dyn, st, (const cx<T>*)P.ypil, dcvals, inv_snr, np, dv, F);
    hipLaunchKernelGGL(p2_apply_operator_kernel<T>, og, dim3(128), sizeof(cx<T>) * np * P2_FT, st, (const double*)pl->d_p2_sop,
                       (const cx<T>*)dv, dh[1], nc, np, F);
  }
  OFDM_TRY(check_launch("LS / MMSE stage"));
  // MP: taps -> equalise (H written to dh[2])
  {
    OFDM_ARG(pl->k_atoms >= np, "MP_estimate: the loop bound is Np columns (MP_estimate.m:10) but the dictionary has fewer");
    const int kc = np;
    const MpLayout lay = mp_layout<T>(np, kc, taps);
    OFDM_ARG(lay.total <= 150 * 1024, "task5_part2_tile: MP stage needs %u bytes of LDS", lay.total);
    const unsigned grid = cdiv_u(F, 4 * lay.fpw);
    const bool mfma = !f64 && (kc % 16 == 0) && (np % 4 == 0) && !getenv("OFDM_MP_NO_MFMA");
    if (mfma) {
      if constexpr (!f64) {
        OFDM_HIP(hipFuncSetAttribute((const void*)mp_batch_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total));
        hipLaunchKernelGGL((mp_batch_kernel<float, true>), dim3(grid), dim3(256), lay.total, st, P, lay, kc, F);
      }
    } else {
      OFDM_HIP(hipFuncSetAttribute((const void*)mp_batch_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total));
      hipLaunchKernelGGL((mp_batch_kernel<T, false>), dim3(grid), dim3(256), lay.total, st, P, lay, kc, F);
    }
    OFDM_TRY(check_launch("mp_batch_kernel"));
    OFDM_TRY(eq_demap_run<T>(pv, P, dxk, nc, false, F, nullptr, dref, derrs + 2 * F, dh[2], nullptr, nullptr, 0, 0));
  }
  // OMP
  OFDM_TRY(omp_batch_run<T>(P, F));
  OFDM_TRY(eq_demap_run<T>(pv, P, dxk, nc, false, F, nullptr, dref, derrs + 3 * F, dh[3], nullptr, nullptr, 0, 0));
  // LS, MMSE: equalise with the given H
  P.h_in = dh[0];
  OFDM_TRY(eq_demap_run<T>(pv, P, dxk, nc, true, F, nullptr, dref, derrs + 0 * F, nullptr, nullptr, nullptr, 0, 0));
  P.h_in = dh[1];
  OFDM_TRY(eq_demap_run<T>(pv, P, dxk, nc, true, F, nullptr, dref, derrs + 1 * F, nullptr, nullptr, nullptr, 0, 0));
  hipLaunchKernelGGL(p2_nmse_kernel<T>, dim3((unsigned)F), dim3(256), 0, st, ddelay, damp, n_ch, N, nc, (const cx<T>*)dh[0],
                     (const cx<T>*)dh[1], (const cx<T>*)dh[2], (const cx<T>*)dh[3], dnmse, F);
  return check_launch("p2_nmse_kernel");
}


// ---- MMSE_CE.m:18-25 with h = ifft(H_est_LS) (Main_model_Task_5.m:314-315): the N_carrier-point inverse DFT of the LS estimate
// (any N_carrier: direct sums over a table of the N_carrier-th roots of unity, exact index arithmetic), then
// hh = h h', r = sum(|h|^2 k) / hh, r2 = sum(|h|^2 k^2) / hh, tau_rms = sqrt(r2 - r^2); cvals[f] = 2 pi tau_rms df Nps.
template <typename T>
__global__ __launch_bounds__(256) void mse_tau_kernel(const cx<T>* __restrict__ hls, int nc, double nps, double* __restrict__ cvals) {
  extern __shared__ __attribute__((aligned(16))) unsigned char p2_smem[];
  c64* const root = (c64*)p2_smem;                                  // exp(+2 pi i j / nc)
  c64* const hl = root + nc;                                        // the frame's H_est_LS in double
  __shared__ double red[3][4];
  const int64_t f = blockIdx.x;
  for (int j = threadIdx.x; j < nc; j += 256) {
    double sn, cs;
    sincospi(2.0 * (double)j / (double)nc, &sn, &cs);
    root[j] = c64{cs, sn};
    const cx<T> v = hls[f * nc + j];
    hl[j] = c64{(double)v.x, (double)v.y};
  }
  __syncthreads();
  double hh = 0, s1 = 0, s2 = 0;
  for (int k = threadIdx.x; k < nc; k += 256) {
    double ar = 0, ai = 0;
    int e = 0;                                                      // (k m) mod nc
    for (int m = 0; m < nc; ++m) {
      const c64 w = root[e], x = hl[m];
      ar += x.x * w.x - x.y * w.y;
      ai += x.x * w.y + x.y * w.x;
      e += k;
      if (e >= nc) e -= nc;
    }
    ar /= (double)nc; ai /= (double)nc;                             // ifft scaling
    const double p = ar * ar + ai * ai;
    hh += p; s1 += p * (double)k; s2 += p * (double)k * (double)k;
  }
  double v[3] = {hh, s1, s2};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, 64);
    if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double H = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double r = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / H, r2 = (red[2][0] + red[2][1] + red[2][2] + red[2][3]) / H;
    cvals[f] = 2.0 * M_PI * sqrt(r2 - r * r) * (1.0 / (double)nc) * nps;      // :23-26, df = 1 / N_carrier
  }
}

// ---- H(1..N_carrier) = fft(h_est) from the picked taps of MP / OMP (MP_estimate.m:27-33, OMP_estimate.m:25-36): a later pick of
// the same atom overwrites the earlier one (est(index(i)) = x(i)).
template <typename T>
__global__ __launch_bounds__(256) void mse_taps_to_h_kernel(const int32_t* __restrict__ tap_idx, const c64* __restrict__ tap_x, int taps,
                                                            int nfft, int nc, cx<T>* __restrict__ hout) {
  const int64_t f = blockIdx.x;
  for (int k = threadIdx.x; k < nc; k += 256) {
    double hr = 0, hi = 0;
    for (int q = 0; q < taps; ++q) {
      const int idx = tap_idx[f * taps + q];
      if (idx < 0) continue;
      bool later = false;
      for (int q2 = q + 1; q2 < taps; ++q2) later = later || tap_idx[f * taps + q2] == idx;
      if (later) continue;
      const int e = (int)(((int64_t)idx * k) % nfft);
      double sn, cs;
      sincospi(2.0 * (double)e / (double)nfft, &sn, &cs);
      const c64 a = tap_x[f * taps + q];
      hr += a.x * cs + a.y * sn;
      hi += a.y * cs - a.x * sn;
    }
    hout[f * nc + k] = mk<T>((T)hr, (T)hi);
  }
}

template <typename T>
static int mse_tile_run(ofdm_rx_plan* pl, const void* dtx, const void* h_dense, int h_len, const int32_t* ddelay, const c64* damp,
                        int n_ch, const double* snr_db_host, const double* dinv_snr, int64_t F, uint64_t seed, uint32_t stream0,
                        double* dmse, int flags) {
  const int N = pl->nfft, Tg = pl->t_guard, S = pl->n_symb, np = pl->np, nc = pl->n_carrier, taps = pl->taps;
  const int64_t len = (int64_t)(N + Tg) * S;
  const bool f64 = std::is_same<T, double>::value;
  hipStream_t st = ctx().stream;
  const int devflags = (flags & ~OFDM_DEVICE) | OFDM_DEVICE;
  size_t need = 0;
  auto reserve = [&](size_t bytes) { const size_t o = need; need += (bytes + 255) & ~size_t(255); return o; };
  const size_t o_a = reserve(sizeof(cx<T>) * (size_t)len * F), o_b = reserve(sizeof(cx<T>) * (size_t)len * F);
  const size_t o_x = reserve(sizeof(cx<T>) * (size_t)nc * S * F), o_v = reserve(sizeof(cx<T>) * (size_t)np * F);
  const size_t o_c = reserve(sizeof(double) * (size_t)F);
  size_t o_h[4];
  for (int e = 0; e < 4; ++e) o_h[e] = reserve(sizeof(cx<T>) * (size_t)nc * F);
  if (pl->ws_t4_bytes < need) {                                     // the Task-4 arena doubles as this entry's
    OFDM_HIP(hipStreamSynchronize(st));
    if (pl->ws_t4) { (void)hipFree(pl->ws_t4); pl->ws_t4 = nullptr; pl->ws_t4_bytes = 0; }
    OFDM_HIP(hipMalloc(&pl->ws_t4, need));
    pl->ws_t4_bytes = need;
  }
  unsigned char* arena = (unsigned char*)pl->ws_t4;
  cx<T>* da = (cx<T>*)(arena + o_a);
  cx<T>* db = (cx<T>*)(arena + o_b);
  cx<T>* dxk = (cx<T>*)(arena + o_x);
  cx<T>* dv = (cx<T>*)(arena + o_v);
  double* dcv = (double*)(arena + o_c);
  cx<T>* dh[4];
  for (int e = 0; e < 4; ++e) dh[e] = (cx<T>*)(arena + o_h[e]);
  if (!pl->d_p2_sop) {
    std::vector<double> W;
    OFDM_TRY(build_interpolate_operator(pl->pilot_loc.data(), np, nc, 's', W));
    OFDM_HIP(hipMalloc(&pl->d_p2_sop, sizeof(double) * W.size()));
    OFDM_HIP(hipMemcpy(pl->d_p2_sop, W.data(), sizeof(double) * W.size(), hipMemcpyHostToDevice));
  }
  FastPlanView pv;
  make_plan_view(pl, pv);
  pv.ev = nullptr;
  pv.d_wt = nullptr;
  FastParams<T> P;
  const void* tw = nullptr;
  OFDM_TRY(get_twiddles(N, f64, &tw));
  OFDM_TRY(fast_params_prepare<T>(pv, tw, F, P));
  // Tx for every point -> Noise at the point's SNR (:307) -> conv truncated (:308-309) -> demodulator (:311) -> Y (:328)
  const int words = (int)(len * (int64_t)(sizeof(cx<T>) / 4));
  OFDM_ARG((int64_t)words * F < ((int64_t)1 << 40), "task5_mse_tile: tile too large");
  hipLaunchKernelGGL(p2_replicate_kernel<T>, dim3(cdiv_u((int64_t)F * words, 256)), dim3(256), 0, st, (const uint32_t*)dtx, (uint32_t*)da,
                     words, F);
  OFDM_TRY(check_launch("p2_replicate_kernel"));
  OFDM_TRY(ofdm_Noise_frames_snr(snr_db_host, da, len, F, seed, stream0, da, devflags));
  OFDM_TRY(ofdm_channel_conv_frames(da, len, F, h_dense, h_len, db, devflags));
  OFDM_TRY(demod_keep_device(db, dxk, N, F * S, Tg, nc, f64));
  hipLaunchKernelGGL(p2_pilot_ls_kernel<T>, dim3(cdiv_u(F * np, 256)), dim3(256), 0, st, (const cx<T>*)dxk, (const int32_t*)pl->d_pc0,
                     (const cx<T>*)pl->d_pilots, P.ypil, np, S, nc, F);
  // LS_CE (:313): H = Sop * Y
  const dim3 og(cdiv_u(nc, 128), cdiv_u(F, P2_FT));
  hipLaunchKernelGGL(p2_apply_operator_kernel<T>, og, dim3(128), sizeof(cx<T>) * np * P2_FT, st, (const double*)pl->d_p2_sop,
                     (const cx<T>*)P.ypil, dh[0], nc, np, F);
  // MMSE_CE (:314-315): tau_rms of ifft(H_est_LS), then v = rf2 (rf2 + I/snr_i)^-1 Y, H = Sop * v
  {
    const size_t tdyn = sizeof(c64) * 2 * (size_t)nc;
    OFDM_ARG(tdyn <= 150 * 1024, "task5_mse_tile: N_carrier beyond 4800");
    OFDM_HIP(hipFuncSetAttribute((const void*)mse_tau_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tdyn));
    const double nps = (double)pl->pilot_loc[1] - (double)pl->pilot_loc[0];                    // MMSE_CE.m:15
    hipLaunchKernelGGL(mse_tau_kernel<T>, dim3((unsigned)F), dim3(256), tdyn, st, (const cx<T>*)dh[0], nc, nps, dcv);
    int wpw = 4;
    while (wpw > 1 && sizeof(c64) * 4 * (size_t)np * wpw > 150 * 1024) wpw >>= 1;
    const size_t dyn = sizeof(c64) * 4 * (size_t)np * wpw;
    OFDM_ARG(dyn <= 150 * 1024, "task5_mse_tile: MMSE stage supports at most 2343 pilots");
    OFDM_HIP(hipFuncSetAttribute((const void*)mmse_wave_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(mmse_wave_kernel<T>, dim3(cdiv_u(F, wpw)), dim3(64 * wpw), dyn, st, (const cx<T>*)P.ypil, (const double*)dcv, 0.0, np,
                       dv, F, dinv_snr);
    hipLaunchKernelGGL(p2_apply_operator_kernel<T>, og, dim3(128), sizeof(cx<T>) * np * P2_FT, st, (const double*)pl->d_p2_sop,
                       (const cx<T>*)dv, dh[1], nc, np, F);
  }
  OFDM_TRY(check_launch("LS / MMSE stage"));
  // MP_estimate (:330)
  {
    OFDM_ARG(pl->k_atoms >= np, "MP_estimate: the loop bound is Np columns (MP_estimate.m:10) but the dictionary has fewer");
    const int kc = np;
    const MpLayout lay = mp_layout<T>(np, kc, taps);
    OFDM_ARG(lay.total <= 150 * 1024, "task5_mse_tile: MP stage needs %u bytes of LDS", lay.total);
    const unsigned grid = cdiv_u(F, 4 * lay.fpw);
    const bool mfma = !f64 && (kc % 16 == 0) && (np % 4 == 0) && !getenv("OFDM_MP_NO_MFMA");
    if (mfma) {
      if constexpr (!f64) {
        OFDM_HIP(hipFuncSetAttribute((const void*)mp_batch_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total));
        hipLaunchKernelGGL((mp_batch_kernel<float, true>), dim3(grid), dim3(256), lay.total, st, P, lay, kc, F);
      }
    } else {
      OFDM_HIP(hipFuncSetAttribute((const void*)mp_batch_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total));
      hipLaunchKernelGGL((mp_batch_kernel<T, false>), dim3(grid), dim3(256), lay.total, st, P, lay, kc, F);
    }
    hipLaunchKernelGGL(mse_taps_to_h_kernel<T>, dim3((unsigned)F), dim3(256), 0, st, (const int32_t*)P.tap_idx, (const c64*)P.tap_x, taps, N, nc,
                       dh[2]);
    OFDM_TRY(check_launch("MP stage"));
  }
  // OMP_estimate (:331)
  OFDM_TRY(omp_batch_run<T>(P, F));
  hipLaunchKernelGGL(mse_taps_to_h_kernel<T>, dim3((unsigned)F), dim3(256), 0, st, (const int32_t*)P.tap_idx, (const c64*)P.tap_x, taps, N, nc,
                     dh[3]);
  // the four errors against fft(h)(1..N_carrier) (:334-344)
  hipLaunchKernelGGL(p2_nmse_kernel<T>, dim3((unsigned)F), dim3(256), 0, st, ddelay, damp, n_ch, N, nc, (const cx<T>*)dh[0],
                     (const cx<T>*)dh[1], (const cx<T>*)dh[2], (const cx<T>*)dh[3], dmse, F);
  return check_launch("p2_nmse_kernel");
}

}  // namespace ofdm

using namespace ofdm;

extern "C" int ofdm_task5_part2_tile(ofdm_rx_plan* pl, const void* tx_noised, const int32_t* tap_delay, const double* tap_amp,
                                     int n_ch_taps, int64_t n_frames, double snr_db, const uint8_t* ref_bits, double* nmse_out,
                                     uint32_t* errors_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(!pl || !(pl->descr & DESCR_ON), "task5_part2_tile: the study runs unscrambled (Task5_part2.m:104-114,:285-299 are commented out); clear the plan's DeScrambler");
  OFDM_ARG(pl && tx_noised && tap_delay && tap_amp && ref_bits && nmse_out && errors_out, "task5_part2_tile: null argument");
  OFDM_ARG(pl->nd >= 1, "task5_part2_tile: the plan has no data carriers");
  OFDM_PLAN_DEVICE(pl);
  OFDM_ARG((is_f64(flags) ? 1 : 0) == pl->f64, "task5_part2_tile: precision flag differs from the plan's");
  OFDM_ARG(pl->pilots_in_band && pl->np >= 2, "task5_part2_tile: needs at least two pilots, all inside 1..N_carrier");
  OFDM_ARG(n_ch_taps >= 1 && n_ch_taps <= 64 && n_frames >= 0 && n_frames <= 65535, "task5_part2_tile: 1..64 channel taps, at most 65535 realisations");
  OFDM_ARG(chain_split_supported(pl->nfft, pl->n_carrier, pl->taps, pl->bps, (int64_t)pl->nd * pl->n_symb, pl->f64 != 0),
           "task5_part2_tile: the frame's decisions do not fit the equalise stage's LDS");
  if (n_frames == 0) return OFDM_OK;
  const size_t cs = csize(flags);
  const int64_t len = (int64_t)(pl->nfft + pl->t_guard) * pl->n_symb;
  // rms delay spread of every realisation's CIR (MMSE_CE.m:19-24 on h_t(1:N_carrier), Task5_part2.m:176) -> c = 2 pi tau df Nps
  std::vector<double> cvals((size_t)n_frames);
  const double nps = (double)pl->pilot_loc[1] - (double)pl->pilot_loc[0], df = 1.0 / (double)pl->n_carrier;   // MMSE_CE.m:15, :25
  for (int64_t f = 0; f < n_frames; ++f) {
    double hh = 0, s1 = 0, s2 = 0;
    for (int t = 0; t < n_ch_taps; ++t) {
      const int d = tap_delay[f * n_ch_taps + t];
      OFDM_ARG(d >= 0 && d < pl->nfft, "task5_part2_tile: tap delay outside 0..Nfft-1");
      if (d >= pl->n_carrier) continue;                            // h_t(1:N_carrier)
      const double ar = tap_amp[2 * (f * n_ch_taps + t)], ai = tap_amp[2 * (f * n_ch_taps + t) + 1];
      const double p = ar * ar + ai * ai;
      hh += p; s1 += p * d; s2 += p * (double)d * d;
    }
    const double r = s1 / hh, r2 = s2 / hh;
    cvals[f] = 2.0 * M_PI * sqrt(r2 - r * r) * df * nps;
  }
  Stage st(flags);
  const void *dtx, *ddel, *damp, *dcv, *dref;
  void *dnm, *der;
  OFDM_TRY(st.in(tx_noised, cs * (size_t)len, &dtx));
  OFDM_TRY(st.upload(tap_delay, sizeof(int32_t) * (size_t)n_frames * n_ch_taps, &ddel));
  OFDM_TRY(st.upload(tap_amp, sizeof(double) * 2 * (size_t)n_frames * n_ch_taps, &damp));
  OFDM_TRY(st.upload(cvals.data(), sizeof(double) * (size_t)n_frames, &dcv));
  OFDM_TRY(st.in(ref_bits, (size_t)pl->frame_words * 4, &dref));
  OFDM_TRY(st.out(nmse_out, sizeof(double) * 4 * (size_t)n_frames, &dnm));
  OFDM_TRY(st.out(errors_out, sizeof(uint32_t) * 4 * (size_t)n_frames, &der));
  const double inv_snr = 1.0 / pow(10.0, snr_db * 0.1);            // MMSE_CE.m:13
  if (pl->f64) OFDM_TRY(part2_tile_run<double>(pl, dtx, (const int32_t*)ddel, (const c64*)damp, n_ch_taps, n_frames, (const double*)dcv,
                                                inv_snr, dref, (double*)dnm, (uint32_t*)der));
  else OFDM_TRY(part2_tile_run<float>(pl, dtx, (const int32_t*)ddel, (const c64*)damp, n_ch_taps, n_frames, (const double*)dcv, inv_snr,
                                      dref, (double*)dnm, (uint32_t*)der));
  return st.finish();
}

extern "C" int ofdm_task5_mse_tile(ofdm_rx_plan* pl, const void* tx, const int32_t* tap_delay, const double* tap_amp, int n_ch_taps,
                                   const double* snr_db, int64_t n_points, uint64_t seed, uint32_t stream0, double* mse_out,
                                   int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(pl && tx && tap_delay && tap_amp && snr_db && mse_out, "task5_mse_tile: null argument");
  OFDM_PLAN_DEVICE(pl);
  OFDM_ARG((is_f64(flags) ? 1 : 0) == pl->f64, "task5_mse_tile: precision flag differs from the plan's");
  OFDM_ARG(pl->pilots_in_band && pl->np >= 2, "task5_mse_tile: needs at least two pilots, all inside 1..N_carrier");
  OFDM_ARG(n_ch_taps >= 1 && n_ch_taps <= 64 && n_points >= 0 && n_points <= 65535, "task5_mse_tile: 1..64 channel taps, at most 65535 points");
  OFDM_ARG((int64_t)stream0 + n_points < ((int64_t)1 << 32), "task5_mse_tile: stream index outside 32 bits");
  if (n_points == 0) return OFDM_OK;
  const size_t cs = csize(flags);
  const int64_t len = (int64_t)(pl->nfft + pl->t_guard) * pl->n_symb;
  // dense impulse response for conv (get_MP_channel_resp.m: amplitudes at their delays), the channel per point for the error
  // kernel, 1 / snr per point (MMSE_CE.m:13)
  int h_len = 0;
  for (int t = 0; t < n_ch_taps; ++t) {
    OFDM_ARG(tap_delay[t] >= 0 && tap_delay[t] < pl->nfft, "task5_mse_tile: tap delay outside 0..Nfft-1");
    h_len = std::max(h_len, tap_delay[t] + 1);
  }
  std::vector<c64> h64((size_t)h_len, c64{0, 0});
  for (int t = 0; t < n_ch_taps; ++t) h64[tap_delay[t]] = c64{tap_amp[2 * t], tap_amp[2 * t + 1]};   // a later tap at the same delay wins
  std::vector<c32> h32;
  const void* h_dense = h64.data();
  if (!pl->f64) {
    h32.resize(h64.size());
    for (size_t i = 0; i < h64.size(); ++i) h32[i] = c32{(float)h64[i].x, (float)h64[i].y};
    h_dense = h32.data();
  }
  std::vector<int32_t> del((size_t)n_points * n_ch_taps);
  std::vector<double> amp(2 * (size_t)n_points * n_ch_taps), inv_snr((size_t)n_points);
  for (int64_t f = 0; f < n_points; ++f) {
    for (int t = 0; t < n_ch_taps; ++t) {
      // the dense vector is what the script convolves with: the error is measured against ITS transform
      del[f * n_ch_taps + t] = tap_delay[t];
      const c64 a = h64[tap_delay[t]];
      bool dup = false;
      for (int t2 = t + 1; t2 < n_ch_taps; ++t2) dup = dup || tap_delay[t2] == tap_delay[t];
      amp[2 * (f * n_ch_taps + t)] = dup ? 0.0 : a.x;
      amp[2 * (f * n_ch_taps + t) + 1] = dup ? 0.0 : a.y;
    }
    inv_snr[f] = 1.0 / pow(10.0, snr_db[f] * 0.1);
  }
  Stage st(flags);
  const void *dtx, *ddel, *damp, *dinv;
  void* dms;
  OFDM_TRY(st.in(tx, cs * (size_t)len, &dtx));
  OFDM_TRY(st.upload(del.data(), sizeof(int32_t) * del.size(), &ddel));
  OFDM_TRY(st.upload(amp.data(), sizeof(double) * amp.size(), &damp));
  OFDM_TRY(st.upload(inv_snr.data(), sizeof(double) * inv_snr.size(), &dinv));
  OFDM_TRY(st.out(mse_out, sizeof(double) * 4 * (size_t)n_points, &dms));
  if (pl->f64) OFDM_TRY(mse_tile_run<double>(pl, dtx, h_dense, h_len, (const int32_t*)ddel, (const c64*)damp, n_ch_taps, snr_db,
                                              (const double*)dinv, n_points, seed, stream0, (double*)dms, flags));
  else OFDM_TRY(mse_tile_run<float>(pl, dtx, h_dense, h_len, (const int32_t*)ddel, (const c64*)damp, n_ch_taps, snr_db, (const double*)dinv,
                                    n_points, seed, stream0, (double*)dms, flags));
  return st.finish();
}
