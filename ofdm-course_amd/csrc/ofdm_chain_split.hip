// Split form of the fused Task-5 RX chain for the geometries the wave-local fast path does not take
// (Nfft = 8192: BASELINE config "sparse 32-tap channel, OMP_estimate"; or a frame whose state does not fit the
// single generic kernel's LDS).  Four stages, intermediates in HBM restricted to the carriers that are read:
//
//   demod_keep8192_kernel / demod_keep_kernel
//                       OFDM_demodulator of every symbol, rows 1..N_carrier kept        (OFDM_demodulator.m:2-10)
//   pilot_ls_kernel     Y = X(pilotCarriers, 1) ./ pilotValues(:, 1)                    (Task5_part2.m:190)
//   omp_batch_kernel    batch OMP with the MFMA dictionary correlation (ofdm_chain_fast.hip), or
//   mmse_apply_*        the plan's MMSE operator (ofdm_chain_mmse.hip)
//   eq_demap_kernel     H = fft(h)(1..N_carrier) from the taps -> equalize_signal -> get_payload -> demapping
//                       -> packed bits -> BER numerator                                 (OMP_estimate.m:25-36, ...)
//
// Against the generic single kernel this replaces the O(taps * K * Np) residual correlations by the Gram-table
// form (O(K * taps^2)) and runs the transform at full occupancy; it costs one write + read of X(1..N_carrier,:).
#include <algorithm>

#include "chain_fast_core.hpp"

namespace ofdm {

int demod_keep_device(const void* y, void* x, int nfft, int64_t n_symb, int t_guard, int n_keep, bool f64);   // ofdm_modem.hip

template <typename T>
__global__ __launch_bounds__(256) void pilot_ls_kernel(FastParams<T> P, const cx<T>* __restrict__ xk,
                                                       const int32_t* __restrict__ pc0, int64_t n_frames) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_frames * P.np) return;
  const int64_t f = i / P.np;
  const int p = (int)(i - f * P.np);
  P.ypil[i] = cdiv(xk[f * P.n_symb * (int64_t)P.n_carrier + pc0[p]], P.pilots[p]);
}

// one workgroup per frame (grid-stride): G = 1 ./ H in LDS, then every symbol's data carriers
// VEC (fp32, N_carrier and the row stride even): a thread takes PAIRS of neighbouring carriers with one 16-byte load per (pair,
// symbol), four symbols x two pairs requested together, and decides every carrier it loads -- a carrier without data (pilot, or
// past N_carrier) stores its decision to a dump byte behind the codes: no exec-mask branch per sample, half the load
// instructions, twice the bytes in flight of the 8-byte form.
template <typename T, int BA, bool HEXT, bool VEC = false>
__global__ __launch_bounds__(256) void eq_demap_kernel(FastParams<T> P, int nfft, const cx<T>* __restrict__ xk,
                                                       int x_stride /* rows per symbol column of xk */, int64_t n_frames, uint32_t* __restrict__ bits_out,
                                                       const uint32_t* __restrict__ ref_bits,
                                                       uint32_t* __restrict__ errors_out, cx<T>* __restrict__ h_out,
                                                       int32_t* __restrict__ index_out, DemapTable<T> tab,
                                                       const double* __restrict__ fine_est /* [n_frames][2] or null */,
                                                       int time_desync, int freq_desync) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // fine_est: the residual timing / phase estimates of fine_sync (tau, phase) per frame; its rotation
  // exp(j (2 pi tau k + phase)) (fine_sync.m:36-43) is applied to every sample as it is read, with the rounding of the
  // separate pass, instead of rewriting X
  cx<T>* geq = (cx<T>*)smem;                                            // [n_carrier]
  uint8_t* codes = (uint8_t*)(geq + P.n_carrier);                       // [n_symb * nd] (+ padding to 32)
  __shared__ unsigned int sh_err;
  __shared__ int sh_tidx[FAST_MAXT];
  __shared__ c64 sh_tx[FAST_MAXT];
  __shared__ T sh_lut[DEMAP_LUT_ELEMS];
  const int gid = threadIdx.x;
  if constexpr (BA >= 2) demap_lut_fill<T, BA>(tab, sh_lut, gid);
  const int taps = P.taps, nd = P.nd, nc = P.n_carrier;
  const int bps = BA > 0 ? 2 * BA : P.bps;
  const int n_codes = nd * P.n_symb;
  const int n_groups = (n_codes + 31) >> 5;
  if (gid < 32) codes[((n_codes + 31) & ~31) - 32 + gid] = 0;          // zero padding of the last 32-symbol group
  __syncthreads();
  for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
    if (gid == 0) sh_err = 0;
    if (!HEXT && gid < taps) {
      const int idx = P.tap_idx[f * taps + gid];
      sh_tidx[gid] = idx;
      sh_tx[gid] = P.tap_x[f * taps + gid];
      if (index_out) index_out[f * taps + gid] = idx + 1;
    }
    __syncthreads();
    if constexpr (HEXT) {
      for (int k = gid; k < nc; k += 256) {
        const cx<T> H = P.h_in[f * nc + k];
        if (h_out) h_out[f * nc + k] = H;
        geq[k] = cdiv(mk<T>(1, 0), H);
      }
    } else {
      // H(k) = sum_q x_q W^(idx_q k) for the thread's carriers k = base + gid + 256 u.  A table lookup per (carrier, tap)
      // is a 64-address gather (stride idx_q) and made this stage gather-bound; instead ONE lookup per tap gives
      // x_q W^(idx_q (base + gid)) and the eight wavefront-uniform factors W^(256 u idx_q) come through the scalar cache:
      // four FMAs per (carrier, tap), exact table values at both ends (an eight-step recurrence cost eight).
      for (int base = 0; base < nc; base += 8 * 256) {
        T hr[8], hi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) hr[u] = hi[u] = 0;
        for (int q = 0; q < taps; ++q) {
          const int idx = __builtin_amdgcn_readfirstlane(sh_tidx[q]);
          if (idx < 0) continue;                                       // wavefront-uniform
          const cx<T> x = mk<T>((T)sh_tx[q].x, (T)sh_tx[q].y);
          const cx<T> xw = x * P.tw[(int)(((int64_t)idx * (base + gid)) & (nfft - 1))];
          cx<T> st[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) st[u] = uniform_load(P.tw + ((idx * 256 * u) & (nfft - 1)));
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            hr[u] = fma(xw.x, st[u].x, hr[u]); hr[u] = fma(-xw.y, st[u].y, hr[u]);
            hi[u] = fma(xw.x, st[u].y, hi[u]); hi[u] = fma(xw.y, st[u].x, hi[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = base + gid + 256 * u;
          if (k < nc) {
            const cx<T> H = mk<T>((T)hr[u], (T)hi[u]);
            if (h_out) h_out[f * nc + k] = H;
            geq[k] = cdiv(mk<T>(1, 0), H);
          }
        }
      }
    }
    if (fine_est) {
      // the fine-sync rotation exp(j (2 pi tau k + phase)) of carrier k (fine_sync.m:23-29, :39-43) is the same for every
      // symbol of the frame: folded into the equaliser coefficient (formed in double, rounded once) -- one complex multiply
      // per sample instead of a double-precision rotation and a multiply
      __syncthreads();                                                // every geq[k] of this frame is in place
      const double tau = fine_est[2 * f], ph = fine_est[2 * f + 1];
      double psn = 0.0, pcs = 1.0;
      if (freq_desync) sincos(ph, &psn, &pcs);
      for (int k = gid; k < nc; k += 256) {
        double cs = 1.0, sn = 0.0;
        if (time_desync) {
          const double t = tau * (double)k;
          sincospi(2.0 * (t - floor(t)), &sn, &cs);
        }
        const double rr = cs * pcs - sn * psn, ri = sn * pcs + cs * psn;
        const cx<T> g = geq[k];
        geq[k] = mk<T>((T)((double)g.x * rr - (double)g.y * ri), (T)((double)g.x * ri + (double)g.y * rr));
      }
    }
    __syncthreads();
    const cx<T>* xf = xk + f * P.n_symb * (int64_t)x_stride;
    if constexpr (VEC) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      const int dump = (n_codes + 31) & ~31;                           // one byte behind the zero padding
      for (int k0 = 2 * gid; k0 < nc; k0 += 2 * 512) {
        int ci[2][2], cinc[2][2];
        cx<T> g[2][2];
        const cx<T>* xp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = k0 + 512 * u;
          const bool act = k < nc;                                     // k even, nc even: k + 1 < nc as well
          const int kc = act ? k : 0;
          xp[u] = xf + kc;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int d = act ? (int)P.drole[kc + h] : -1;
            ci[u][h] = d >= 0 ? d : dump;
            cinc[u][h] = d >= 0 ? nd : 0;
            g[u][h] = geq[kc + h];
          }
        }
        for (int s0 = 0; s0 < P.n_symb; s0 += 4) {
          v4f xv[4][2];
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (s0 + v < P.n_symb) {                                   // workgroup-uniform
#pragma unroll
              for (int u = 0; u < 2; ++u)
                xv[v][u] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(xp[u] + (int64_t)(s0 + v) * x_stride));
            }
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (s0 + v < P.n_symb) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const cx<T> y0 = mk<T>(xv[v][u].x, xv[v][u].y) * g[u][0], y1 = mk<T>(xv[v][u].z, xv[v][u].w) * g[u][1];
                codes[ci[u][0] + (s0 + v) * cinc[u][0]] = (uint8_t)slice_symbol<T, BA>(tab, y0);
                codes[ci[u][1] + (s0 + v) * cinc[u][1]] = (uint8_t)slice_symbol<T, BA>(tab, y1);
              }
            }
        }
      }
    } else {
    // 2 symbols x 8 carriers per thread requested together: this stage streams X(1..N_carrier, :) and is bound by the bytes it
      // keeps in flight (a register double buffer of the next step was measured: 170 VGPRs, half the wavefronts, slower)
      for (int k0 = gid; k0 < nc; k0 += 8 * 256) {
        int dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 256 * u;
          dv[u] = k < nc ? (int)P.drole[k] : -1;
        }
        for (int s0 = 0; s0 < P.n_symb; s0 += 2) {
          cx<T> xv[2][8];
#pragma unroll
          for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int u = 0; u < 8; ++u)
              xv[v][u] = (dv[u] >= 0 && s0 + v < P.n_symb) ? nt_load(xf + (int64_t)(s0 + v) * x_stride + k0 + 256 * u) : mk<T>(0, 0);
#pragma unroll
          for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (dv[u] >= 0 && s0 + v < P.n_symb) {
                const cx<T> xs = xv[v][u];
                const cx<T> ye = xs * geq[k0 + 256 * u];
                int code;
                if constexpr (BA >= 2 && sizeof(T) == 8) code = demap_square_lut<T, BA>(tab, sh_lut, ye);   // parity mode: exact, 9 ops / axis
                else code = slice_symbol<T, BA>(tab, ye);                                                  // fp32: arithmetic rank
                codes[(s0 + v) * nd + dv[u]] = (uint8_t)code;
              }
        }
      }
    }
    __syncthreads();
    const unsigned int err = pack_frame<2 * BA>(codes, n_codes, bps, P.frame_words,
                                                bits_out ? bits_out + f * P.frame_words : nullptr,
                                                ref_bits ? ref_bits + f * P.frame_words : nullptr, gid, 256);
    if (ref_bits && errors_out) {
      if (err) atomicAdd(&sh_err, err);
      __syncthreads();
      if (gid == 0) errors_out[f] = sh_err;
    }
    __syncthreads();
  }
}

// OFDM_demodulator for Nfft = 8192 on the wave-local machinery of the fast path: one radix-2 decimation-in-frequency
// step in registers,  even bins <- FFT4096(x[m] + x[m+4096]),  odd bins <- FFT4096((x[m] - x[m+4096]) W_8192^m),
// then each half is an NW = 8 transform (radix-8 DIF exchange + one 512-point transform per wavefront, the last pass
// pruned when only bins < 2048 are kept).  Persistent 512-thread workgroups; the kept rows are collected in LDS and
// written out contiguously.
template <typename T, bool PRUNE2>
__global__ __launch_bounds__(512, sizeof(T) == 4 ? 4 : 2) void demod_keep8192_kernel(const cx<T>* __restrict__ y, cx<T>* __restrict__ x,
                                                             const cx<T>* __restrict__ tw4096, const cx<T>* __restrict__ tw8192,
                                                             int64_t n_symb, int t_guard, int n_keep,
                                                             cx<T>* __restrict__ ypil /* or null */, const int32_t* __restrict__ pc0,
                                                             const cx<T>* __restrict__ pilots, int np, int n_symb_frame,
                                                             const int16_t* __restrict__ drole /* or null: every row is wanted */,
                                                             int64_t sym_stride /* samples between processed symbols */) {
  constexpr int NW = 8;
  constexpr int psh = 4;                                       // staging pad: one element every 16
  constexpr int NOUT = PRUNE2 ? 2 : 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* lwv = (cx<T>*)smem;                                   // [NW][WAVE_LDS_ELEMS]; also the output staging
  cx<T>* const ex = lwv;
  cx<T>* twl = lwv + NW * WAVE_LDS_ELEMS;                      // [WAVE_TW_ELEMS]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gid = threadIdx.x;
  DifTw<T, NW> dt;
  wave_tw_fill<T, NW>(twl, tw4096);
  dif_tw_init<T, NW>(dt, gid, tw4096);
  cx<T> twb[7], w2[8];
#pragma unroll
  for (int t = 1; t < 8; ++t) twb[t - 1] = tw4096[(t * (lane & 7) * 8) * NW];
#pragma unroll
  for (int t = 0; t < 8; ++t) w2[t] = tw8192[gid + 512 * t];
  // rows of this wavefront: even half 2 k', odd half 2 k' + 1, k' = NW (lane + 64 t) + wave.  A half whose rows hold no data
  // carrier (comb pilots 1:4:end: the even half of every even wavefront) is only needed on a frame's first symbol (pilot LS):
  // on the other symbols the wavefront skips that 512-point transform (the rows it would have produced are never read).
  bool even_data = true, odd_data = true;
  if (drole) {
    bool e = false, o = false;
#pragma unroll
    for (int t = 0; t < NOUT; ++t) {
      const int k = 2 * (NW * (lane + 64 * t) + wave);
      e = e || (k < n_keep && drole[k] >= 0);
      o = o || (k + 1 < n_keep && drole[k + 1] >= 0);
    }
    even_data = __any(e) != 0;
    odd_data = __any(o) != 0;
  }
  __syncthreads();
  for (int64_t s = blockIdx.x; s < n_symb; s += gridDim.x) {
    const cx<T>* src = y + s * sym_stride + t_guard;
    const bool first_of_frame = (unsigned)s % (unsigned)n_symb_frame == 0u;
    cx<T> a[8], b[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) { a[t] = nt_load(src + gid + 512 * t); b[t] = nt_load(src + gid + 512 * t + 4096); }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const cx<T> d = a[t] - b[t];
      a[t] = a[t] + b[t];
      b[t] = d * w2[t];
    }
    cx<T> oe[NOUT], oo[NOUT];
    // even bins
    dif_stage<T, NW>(a, dt);
    __syncthreads();                                           // previous symbol's staged rows have been written out
    dif_scatter<T, NW>(a, gid, ex);
    __syncthreads();
    if (even_data || first_of_frame) {                         // wavefront-uniform
      dif_gather<T>(a, wave, lane, ex);
      wave_fft512<T, PRUNE2>(a, lane, twb, twl, lwv + wave * WAVE_LDS_ELEMS);
#pragma unroll
      for (int t = 0; t < NOUT; ++t) oe[t] = a[t];
    } else {
#pragma unroll
      for (int t = 0; t < NOUT; ++t) oe[t] = mk<T>(0, 0);
    }
    // odd bins
    dif_stage<T, NW>(b, dt);
    __syncthreads();
    dif_scatter<T, NW>(b, gid, ex);
    __syncthreads();
    if (odd_data || first_of_frame) {
      dif_gather<T>(b, wave, lane, ex);
      wave_fft512<T, PRUNE2>(b, lane, twb, twl, lwv + wave * WAVE_LDS_ELEMS);
#pragma unroll
      for (int t = 0; t < NOUT; ++t) oo[t] = b[t];
    } else {
#pragma unroll
      for (int t = 0; t < NOUT; ++t) oo[t] = mk<T>(0, 0);
    }
    __syncthreads();                                           // every wavefront is done with its private region
    // bin k' = NW (lane + 64 t) + wave of each half  ->  rows 2k' (even) and 2k'+1 (odd); staged in pieces of STG rows.
    // Neighbouring lanes hold rows 16 apart: the staging index is padded by one element every 16 (row i at i + i/16), which
    // makes the scatter conflict-free (it was 16-way: SQ_LDS_BANK_CONFLICT 40 % of the kernel's LDS cycles) and keeps the
    // contiguous read-out at most 2-way.  Optionally the pilot LS values Y = X(pilotCarriers, 1) ./ pilotValues(:, 1) of a
    // frame's first symbol are taken from the staged rows (Task5_part2.m:190): no separate pass over X.
    constexpr int STG = (NW * WAVE_LDS_ELEMS) * 16 / 17;         // rows per piece
    for (int base = 0; base < n_keep; base += STG) {
#pragma unroll
      for (int t = 0; t < NOUT; ++t) {
        const int k = 2 * (NW * (lane + 64 * t) + wave) - base;
        if (k >= 0 && k < STG && k + base < n_keep) lwv[k + (k >> psh)] = oe[t];
        if (k + 1 >= 0 && k + 1 < STG && k + 1 + base < n_keep) lwv[k + 1 + ((k + 1) >> psh)] = oo[t];
      }
      __syncthreads();
      cx<T>* dst = x + s * (int64_t)n_keep + base;
      const int cnt = n_keep - base < STG ? n_keep - base : STG;
      for (int i = gid; i < cnt; i += 512) nt_store(dst + i, lwv[i + (i >> psh)]);
      const unsigned su = (unsigned)s, fr = su / (unsigned)n_symb_frame;    // symbol counts fit 32 bits (checked on the host)
      if (ypil && su - fr * (unsigned)n_symb_frame == 0u) {
        for (int p = gid; p < np; p += 512) {
          const int k = pc0[p] - base;
          if (k >= 0 && k < cnt) ypil[(size_t)fr * np + p] = cdiv(lwv[k + (k >> psh)], pilots[p]);
        }
      }
      if (base + STG < n_keep) __syncthreads();
    }
  }
}

template <typename T>
static int demod_keep8192_run(const void* y, void* x, int64_t n_symb, int t_guard, int n_keep, void* ypil, const int32_t* pc0,
                              const void* pilots, int np, int n_symb_frame, const void* drole, int64_t sym_stride = 0) {
  if (sym_stride == 0) sym_stride = 8192 + t_guard;              // every symbol of the stream; a frame length = first symbols only
  const void *tw4 = nullptr, *tw8 = nullptr;
  OFDM_ARG(n_symb < (int64_t)1 << 31, "rx_chain_task5: more than 2^31 symbols in one call");
  OFDM_TRY(get_twiddles(4096, std::is_same<T, double>::value, &tw4));
  OFDM_TRY(get_twiddles(8192, std::is_same<T, double>::value, &tw8));
  const size_t dyn = sizeof(cx<T>) * ((size_t)8 * WAVE_LDS_ELEMS + WAVE_TW_ELEMS);
  auto launch = [&](auto kern) -> int {
    const int per_cu = resident_blocks_per_cu((const void*)kern, 512, dyn);
    const unsigned grid = (unsigned)std::min<int64_t>(n_symb, (int64_t)ctx().num_cu * per_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), dyn, ctx().stream, (const cx<T>*)y, (cx<T>*)x, (const cx<T>*)tw4,
                       (const cx<T>*)tw8, n_symb, t_guard, n_keep, (cx<T>*)ypil, pc0, (const cx<T>*)pilots, np, n_symb_frame,
                       (const int16_t*)drole, sym_stride);
    return check_launch("demod_keep8192_kernel");
  };
  if (n_keep <= 2048) return launch(demod_keep8192_kernel<T, true>);
  return launch(demod_keep8192_kernel<T, false>);
}

// ---- one pass for Nfft 8192 on the eight-wavefront transform above -----------------------------------------------------------
// The symbols 2..S of a frame on demod_keep8192_kernel's transform (radix-2 step in registers, two NW = 8 transforms, the last
// radix-8 pass pruned to the kept rows), but the rows never leave the registers they come out in: each thread holds four carriers
// (2k', 2k'+1 for two k'), multiplies by its 1 ./ H (synthesised once per frame from the OMP taps, or read in MMSE mode), slices,
// and the frame's decisions are packed and counted from LDS -- rx_symbols_kernel's frame loop around the 8192-point transform.
// Sixteen wavefronts per CU (two 512-thread workgroups, 128 VGPRs) against the eight of rx_symbols_coop4_kernel (250 VGPRs, 79 KB).
template <int BA, bool HEXT>
__global__ __launch_bounds__(512, 4) void rx_symbols_r2_kernel(FastParams<float> P, const cx<float>* __restrict__ rx,
                                                               const cx<float>* __restrict__ tw4096, const cx<float>* __restrict__ tw8192,
                                                               int64_t n_frames, uint32_t* __restrict__ bits_out,
                                                               const uint32_t* __restrict__ ref_bits, uint32_t* __restrict__ errors_out,
                                                               cx<float>* __restrict__ h_out, int32_t* __restrict__ index_out,
                                                               DemapTable<float> tab) {
  using T = float;
  constexpr int NW = 8, N = 8192, NOUT = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* lwv = (cx<T>*)smem;                                   // [NW][WAVE_LDS_ELEMS]
  cx<T>* const ex = lwv;
  cx<T>* twl = lwv + NW * WAVE_LDS_ELEMS;                      // [WAVE_TW_ELEMS]
  uint8_t* codes = (uint8_t*)(twl + WAVE_TW_ELEMS);            // [n_symb * nd] (+ padding to 32)
  __shared__ unsigned int sh_err;
  __shared__ int sh_tidx[FAST_MAXT];
  __shared__ c64 sh_tx[FAST_MAXT];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gid = threadIdx.x;
  DifTw<T, NW> dt;
  wave_tw_fill<T, NW>(twl, tw4096);
  dif_tw_init<T, NW>(dt, gid, tw4096);
  cx<T> twb[7], w2[8];
#pragma unroll
  for (int t = 1; t < 8; ++t) twb[t - 1] = tw4096[(t * (lane & 7) * 8) * NW];
#pragma unroll
  for (int t = 0; t < 8; ++t) w2[t] = tw8192[gid + 512 * t];
  // this thread's carriers: kc[2 t] = 2 k' (even half), kc[2 t + 1] = 2 k' + 1 (odd half), k' = NW (lane + 64 t) + wave
  int kc[2 * NOUT], dd[2 * NOUT];
  bool e = false, o = false;
#pragma unroll
  for (int t = 0; t < NOUT; ++t) {
    const int k = 2 * (NW * (lane + 64 * t) + wave);
    kc[2 * t] = k; kc[2 * t + 1] = k + 1;
    dd[2 * t] = k < P.n_carrier ? (int)P.drole[k] : -1;
    dd[2 * t + 1] = k + 1 < P.n_carrier ? (int)P.drole[k + 1] : -1;
    e = e || dd[2 * t] >= 0;
    o = o || dd[2 * t + 1] >= 0;
  }
  // a half whose rows hold no data carrier (comb pilots 1:4:end: the even half of every even wavefront) is skipped
  const bool even_data = __any(e) != 0, odd_data = __any(o) != 0;
  const int Lsym = N + P.t_guard;
  const int taps = P.taps, bps = BA > 0 ? 2 * BA : P.bps, nd = P.nd;
  const int n_codes = nd * P.n_symb;
  if (gid < 32) codes[((n_codes + 31) & ~31) - 32 + gid] = 0;      // zero padding of the last 32-symbol group
  __syncthreads();
  for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
    const cx<T>* frx = rx + f * (int64_t)Lsym * P.n_symb;
    if (gid == 0) sh_err = 0;
    if (!HEXT && gid < taps) {
      const int idx = P.tap_idx[f * taps + gid];
      sh_tidx[gid] = idx;
      sh_tx[gid] = P.tap_x[f * taps + gid];
      if (index_out) index_out[f * taps + gid] = idx + 1;
    }
    __syncthreads();
    // ---- G = 1 ./ H on this thread's carriers: H = fft(h)(1..N_carrier) from the taps (OMP_estimate.m:36), or the MMSE estimate
    cx<T> geq[2 * NOUT];
#pragma unroll
    for (int t = 0; t < 2 * NOUT; ++t) {
      geq[t] = mk<T>(0, 0);
      if (kc[t] < P.n_carrier) {
        cx<T> H;
        if constexpr (HEXT) {
          H = P.h_in[f * P.n_carrier + kc[t]];
        } else {
          float hr = 0, hi = 0;                           // (fp32 sums: 32 terms of magnitude <= 1, the estimate itself is fp32)
          for (int q = 0; q < taps; ++q) {
            const int idx = sh_tidx[q];
            const float xr = (float)sh_tx[q].x, xi = (float)sh_tx[q].y;      // zero for unused / overwritten slots
            const int ee = ((idx < 0 ? 0 : idx) * kc[t]) & (N - 1);           // exact exponent (idx, k < 8192), v_sin / v_cos in turns
            const float turns = (float)ee * (1.0f / (float)N);
            const float ws = __builtin_amdgcn_sinf(turns), wc = __builtin_amdgcn_cosf(turns);
            hr = fmaf(xr, wc, fmaf(xi, ws, hr));             // w = (wc, -ws)
            hi = fmaf(xi, wc, fmaf(-xr, ws, hi));
          }
          H = mk<T>(hr, hi);
        }
        if (h_out) h_out[f * P.n_carrier + kc[t]] = H;
        geq[t] = cdiv(mk<T>(1, 0), H);
      }
    }
    // ---- symbol 1 from the stash
#pragma unroll
    for (int t = 0; t < 2 * NOUT; ++t)
      if (dd[t] >= 0) codes[dd[t]] = (uint8_t)slice_symbol<T, BA>(tab, P.stash[f * P.n_carrier + kc[t]] * geq[t]);
    // ---- symbols 2..S
    for (int s = 1; s < P.n_symb; ++s) {
      const cx<T>* src = frx + (int64_t)s * Lsym + P.t_guard;
      cx<T> a[8], b[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) { a[t] = nt_load(src + gid + 512 * t); b[t] = nt_load(src + gid + 512 * t + 4096); }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const cx<T> d = a[t] - b[t];
        a[t] = a[t] + b[t];
        b[t] = d * w2[t];
      }
      // even bins
      dif_stage<T, NW>(a, dt);
      __syncthreads();                                           // every wavefront has left its private region
      dif_scatter<T, NW>(a, gid, ex);
      __syncthreads();
      if (even_data) {                                           // wavefront-uniform
        dif_gather<T>(a, wave, lane, ex);
        wave_fft512<T, true>(a, lane, twb, twl, lwv + wave * WAVE_LDS_ELEMS);
#pragma unroll
        for (int t = 0; t < NOUT; ++t)
          if (dd[2 * t] >= 0) codes[s * nd + dd[2 * t]] = (uint8_t)slice_symbol<T, BA>(tab, a[t] * geq[2 * t]);
      }
      // odd bins
      dif_stage<T, NW>(b, dt);
      __syncthreads();
      dif_scatter<T, NW>(b, gid, ex);
      __syncthreads();
      if (odd_data) {
        dif_gather<T>(b, wave, lane, ex);
        wave_fft512<T, true>(b, lane, twb, twl, lwv + wave * WAVE_LDS_ELEMS);
#pragma unroll
        for (int t = 0; t < NOUT; ++t)
          if (dd[2 * t + 1] >= 0) codes[s * nd + dd[2 * t + 1]] = (uint8_t)slice_symbol<T, BA>(tab, b[t] * geq[2 * t + 1]);
      }
    }
    __syncthreads();
    const unsigned int err = pack_frame<2 * BA>(codes, n_codes, bps, P.frame_words,
                                                bits_out ? bits_out + f * P.frame_words : nullptr,
                                                ref_bits ? ref_bits + f * P.frame_words : nullptr, gid, 512);
    if (ref_bits && errors_out) {
      if (err) atomicAdd(&sh_err, err);
      __syncthreads();
      if (gid == 0) errors_out[f] = sh_err;
    }
    __syncthreads();                     // codes / sh_err are reused by the next frame
  }
}

static size_t r2_lds_bytes(const FastPlanView& pv) {
  return sizeof(cx<float>) * ((size_t)8 * WAVE_LDS_ELEMS + WAVE_TW_ELEMS) + (((size_t)pv.nd * pv.n_symb + 31) & ~size_t(31));
}
static bool chain_r2_supported(const FastPlanView& pv) {
  if (getenv("OFDM_SPLIT_NO_R2")) return false;
  if (pv.f64 || pv.nfft != 8192 || pv.n_carrier > 2048 || pv.taps > FAST_MAXT || pv.n_symb < 1) return false;
  return r2_lds_bytes(pv) <= 78u * 1024;                       // two workgroups per CU
}
static int chain_r2_symbols_run(const FastPlanView& pv, const FastParams<float>& P, const void* rx, int64_t n_frames, void* bits,
                                const void* ref, void* errs, void* h_out, void* idx_out) {
  const size_t lds = r2_lds_bytes(pv);
  DemapTable<float> tab;
  fill_demap_table<float>(*pv.dict, *pv.cinfo, tab);
  const void *tw4 = nullptr, *tw8 = nullptr;
  OFDM_TRY(get_twiddles(4096, false, &tw4));
  OFDM_TRY(get_twiddles(8192, false, &tw8));
  const bool mmse = pv.d_wt != nullptr;
  auto launch = [&](auto kern) -> int {
    int per_cu = resident_blocks_per_cu((const void*)kern, 512, lds);
    if (const char* e = getenv("OFDM_R2_WG_PER_CU")) per_cu = std::max(1, atoi(e));
    const unsigned grid = (unsigned)std::min<int64_t>(n_frames, (int64_t)ctx().num_cu * per_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, ctx().stream, P, (const cx<float>*)rx, (const cx<float>*)tw4, (const cx<float>*)tw8,
                       n_frames, (uint32_t*)bits, (const uint32_t*)ref, (uint32_t*)errs, (cx<float>*)h_out, (int32_t*)idx_out, tab);
    return OFDM_OK;
  };
  const int ba = pv.cinfo->kind == 1 ? pv.cinfo->bits_per_axis : 0;
#define R2_CASE(BAV)                                                   \
  if (mmse) OFDM_TRY(launch(rx_symbols_r2_kernel<BAV, true>));         \
  else OFDM_TRY(launch(rx_symbols_r2_kernel<BAV, false>))
  switch (ba) {
    case 2: R2_CASE(2); break;
    case 3: R2_CASE(3); break;
    case 4: R2_CASE(4); break;
    default: R2_CASE(0); break;
  }
#undef R2_CASE
  return check_launch("rx_symbols_r2_kernel");
}

bool chain_split_supported(int nfft, int n_carrier, int taps, int bps, int64_t nd_nsymb, bool f64) {
  if (getenv("OFDM_CHAIN_GENERIC")) return false;
  if (taps > FAST_MAXT || bps > 8) return false;
  const size_t lds = (size_t)(f64 ? 16 : 8) * n_carrier + (((size_t)nd_nsymb + 31) & ~size_t(31));
  (void)nfft;
  return lds <= 120 * 1024;
}

// equalise + demap + pack + BER of n_frames frames from X columns of x_stride rows (rows 1..N_carrier are read)
template <typename T>
int eq_demap_run(const FastPlanView& pv, const FastParams<T>& P, const cx<T>* xk, int x_stride, bool hext, int64_t n_frames,
                 void* bits, const void* ref, void* errs, void* h_out, void* idx_out, const double* fine_est, int time_desync,
                 int freq_desync) {
  hipStream_t st = ctx().stream;
  DemapTable<T> tab;
  fill_demap_table<T>(*pv.dict, *pv.cinfo, tab);
  const size_t dyn = sizeof(cx<T>) * (size_t)pv.n_carrier + (((size_t)pv.nd * pv.n_symb + 31) & ~size_t(31)) + 16;
  const bool vec = std::is_same<T, float>::value && (pv.n_carrier & 1) == 0 && (x_stride & 1) == 0 && pv.n_carrier <= 2048 &&
                   !getenv("OFDM_EQD_SCALAR");
  OFDM_ARG(dyn <= 150 * 1024, "rx_chain: equalise / demap stage needs %zu bytes of LDS", dyn);
  auto launch = [&](auto kern) -> int {
    int per_cu = resident_blocks_per_cu((const void*)kern, 256, dyn);
    if (const char* e = getenv("OFDM_EQD_WG_PER_CU")) per_cu = std::max(1, atoi(e));
    const unsigned grid = (unsigned)std::min<int64_t>(n_frames, (int64_t)ctx().num_cu * per_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), dyn, st, P, pv.nfft, xk, x_stride, n_frames, (uint32_t*)bits,
                       (const uint32_t*)ref, (uint32_t*)errs, (cx<T>*)h_out, (int32_t*)idx_out, tab, fine_est, time_desync,
                       freq_desync);
    return OFDM_OK;
  };
  const int ba = pv.cinfo->kind == 1 ? pv.cinfo->bits_per_axis : 0;
#define SPLIT_CASE(BAV)                                                                                    \
  if constexpr (std::is_same<T, float>::value) {                                                           \
    if (vec) {                                                                                             \
      if (hext) OFDM_TRY(launch(eq_demap_kernel<T, BAV, true, true>)); else OFDM_TRY(launch(eq_demap_kernel<T, BAV, false, true>)); \
      break;                                                                                               \
    }                                                                                                      \
  }                                                                                                        \
  if (hext) OFDM_TRY(launch(eq_demap_kernel<T, BAV, true>)); else OFDM_TRY(launch(eq_demap_kernel<T, BAV, false>))
  switch (ba) {
    case 2: SPLIT_CASE(2); break;
    case 3: SPLIT_CASE(3); break;
    case 4: SPLIT_CASE(4); break;
    default: SPLIT_CASE(0); break;
  }
#undef SPLIT_CASE
  return check_launch("eq_demap_kernel");
}
template int eq_demap_run<float>(const FastPlanView&, const FastParams<float>&, const cx<float>*, int, bool, int64_t, void*,
                                 const void*, void*, void*, void*, const double*, int, int);
template int eq_demap_run<double>(const FastPlanView&, const FastParams<double>&, const cx<double>*, int, bool, int64_t, void*,
                                  const void*, void*, void*, void*, const double*, int, int);

template <typename T>
static int split_run(const FastPlanView& pv, const void* tw, const void* rx, int64_t n_frames, void* bits,
                     const void* ref, void* errs, void* h_out, void* idx_out, const int32_t* d_pc0) {
  FastParams<T> P;
  OFDM_TRY(fast_params_prepare<T>(pv, tw, n_frames, P));
  const bool mmse = pv.d_wt != nullptr;
  if constexpr (std::is_same<T, float>::value) {
    const bool coop = chain_coop_supported(pv);                        // (the faster one where its layout conditions hold)
    const bool r2 = !coop && chain_r2_supported(pv);
    if (r2 || coop) {
      // one pass over the samples (ofdm_chain_coop.hip): the first symbol of every frame -> stash + pilot LS values, the
      // estimator, then every other symbol transformed, equalised, sliced, packed and counted without an X round trip
      hipStream_t st = ctx().stream;
      if (pv.fused_out) *pv.fused_out = 0;
      if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[0], st));
      OFDM_TRY(demod_keep8192_run<T>(rx, P.stash, n_frames, pv.t_guard, pv.n_carrier, (void*)P.ypil, d_pc0, P.pilots, pv.np, 1, nullptr,
                                     (int64_t)(8192 + pv.t_guard) * pv.n_symb));
      if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[1], st));
      if (mmse) OFDM_TRY(mmse_stage_run<T>(pv, P, n_frames));
      else OFDM_TRY(omp_batch_run<T>(P, n_frames));
      if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[2], st));
      if (r2) OFDM_TRY(chain_r2_symbols_run(pv, P, rx, n_frames, bits, ref, errs, h_out, idx_out));
      else OFDM_TRY(chain_coop_symbols_run(pv, P, rx, n_frames, bits, ref, errs, h_out, idx_out));
      if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[3], st));
      return OFDM_OK;
    }
  }
  const int64_t need = n_frames * pv.n_symb * (int64_t)pv.n_carrier;
  if (*pv.ws_x_elems < need) {
    OFDM_HIP(hipStreamSynchronize(ctx().stream));
    if (*pv.ws_x) { (void)hipFree(*pv.ws_x); *pv.ws_x = nullptr; }
    OFDM_HIP(hipMalloc(pv.ws_x, sizeof(cx<T>) * (size_t)need));
    *pv.ws_x_elems = need;
  }
  cx<T>* xk = (cx<T>*)*pv.ws_x;
  hipStream_t st = ctx().stream;
  if (pv.fused_out) *pv.fused_out = 0;
  if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[0], st));
  if (pv.nfft == 8192 && !getenv("OFDM_SPLIT_GENERIC_FFT")) {
    // the pilot LS values come out of the transform of each frame's first symbol (no pilot_ls pass)
    const bool fuse = !getenv("OFDM_SPLIT_NO_PLS_FUSE");
    // the rows of pilot-only sub-transforms are skipped on data symbols -- only when the pilot LS values come out of this
    // kernel (the separate pilot_ls pass and h_out-less consumers read the data rows and the first symbol's pilots only)
    OFDM_TRY(demod_keep8192_run<T>(rx, xk, n_frames * pv.n_symb, pv.t_guard, pv.n_carrier, fuse ? (void*)P.ypil : nullptr, d_pc0,
                                   P.pilots, pv.np, pv.n_symb, fuse && !getenv("OFDM_SPLIT_ALL_ROWS") ? pv.d_drole : nullptr));
    if (!fuse) {
      hipLaunchKernelGGL(pilot_ls_kernel<T>, dim3(cdiv_u(n_frames * pv.np, 256)), dim3(256), 0, st, P, (const cx<T>*)xk, d_pc0,
                         n_frames);
      OFDM_TRY(check_launch("pilot_ls_kernel"));
    }
  } else {
    OFDM_TRY(demod_keep_device(rx, xk, pv.nfft, n_frames * pv.n_symb, pv.t_guard, pv.n_carrier, pv.f64 != 0));
    hipLaunchKernelGGL(pilot_ls_kernel<T>, dim3(cdiv_u(n_frames * pv.np, 256)), dim3(256), 0, st, P, (const cx<T>*)xk, d_pc0,
                       n_frames);
    OFDM_TRY(check_launch("pilot_ls_kernel"));
  }
  if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[1], st));
  if (mmse) OFDM_TRY(mmse_stage_run<T>(pv, P, n_frames));
  else OFDM_TRY(omp_batch_run<T>(P, n_frames));
  if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[2], st));
  OFDM_TRY(eq_demap_run<T>(pv, P, xk, pv.n_carrier, mmse, n_frames, bits, ref, errs, h_out, idx_out, nullptr, 0, 0));
  if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[3], st));
  return OFDM_OK;
}

int chain_split_run(const FastPlanView& pv, const void* tw, const void* rx, int64_t n_frames, void* bits,
                    const void* ref, void* errs, void* h_out, void* idx_out, const int32_t* d_pc0) {
  if (pv.f64) return split_run<double>(pv, tw, rx, n_frames, bits, ref, errs, h_out, idx_out, d_pc0);
  return split_run<float>(pv, tw, rx, n_frames, bits, ref, errs, h_out, idx_out, d_pc0);
}

}  // namespace ofdm
