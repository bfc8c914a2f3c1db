// Fused Task-5 RX chain: one launch takes frames of n_symb guarded OFDM symbols from HBM to packed
// decided bits + per-frame bit-error counts.  Nothing but the input samples, the packed bits and
// the reference bits crosses HBM (SURVEY.md section 8d "algorithmic bytes").
//
// Call order restated (T5/Task5_part2.m:169-193,:272,:279-303; sensing matrix
// T5/Main_model_Task_5.m:182-190):
//   X  = OFDM_demodulator(rx, Tg)                                  all symbols of the frame
//   Y  = X(pilotCarriers,1) ./ pilotValues(:,1)                    symbol 1 only
//   [H, h, index] = OMP_estimate(Y, S, Nfft, taps)                 S = P*F(:,1:K)
//   Xeq = equalize_signal(X, H, N_carrier)
//   bits = demapping(-1, get_payload(Xeq, dataCarriers)(:).', Constellation)
//   errors = sum(bits ~= ref_bits)                                 BER_func numerator
//
// One group of Nfft/8 threads owns one frame (one group per workgroup for Nfft >= 2048).
// OMP is evaluated in its "batch" form: the dense dictionary correlation c0 = S^H y is formed
// ONCE per frame; afterwards S^H r_i = c0 - G(:,index) x_i with G = S^H S a function of the atom
// index difference only (closed-form table, double).  The LS refit solves the (<= taps)^2 Gram
// system by Cholesky in double, b_i = a_i^H y is a double dot product, and the stopping rule uses
// ||r_{i-1}-r_i||^2 = ||r_{i-1}||^2 - ||r_i||^2 (nested projections).
#include <cstring>

#include "chain_fast_core.hpp"
#include "rx_plan.hpp"

namespace ofdm {

constexpr int CH_MAXT = 32;      // dominant taps supported by the fused chain

template <typename T>
struct ChainParams {
  int n_symb, t_guard, n_carrier, np, nd, k_atoms, taps;
  const int16_t* prole;      // [nfft] pilot position of a carrier or -1
  const int16_t* drole;      // [nfft] data position of a carrier or -1
  const cx<T>* pilots;       // [np] pilot column (symbol 1)
  const cx<T>* sct;          // [np][k_atoms] conj(S) with the atom index fastest
  const c64* gram;           // [k_atoms] g[d] = sum_p exp(-2 pi i pc0[p] d / nfft)
  const int32_t* pc0;        // [np] 0-based pilot carriers
  const cx<T>* tw;           // [nfft] exp(-2 pi i m / nfft)
  DemapTable<T> tab;
  int frame_words;           // packed 32-bit words per frame
  // byte offsets of the per-group LDS pieces (computed once on the host)
  unsigned off_y, off_c0, off_xs, off_picks, off_gram, off_codes, group_bytes;
};

// ---- group-wide reductions (group = TPX threads, power of two; several groups per workgroup when
//      TPX < 256).  All threads of the WORKGROUP must call them together.
template <int TPX>
__device__ __forceinline__ double group_sum(double v, double* sh /* [wg_waves] */) {
  constexpr int W = TPX < 64 ? TPX : 64;
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if constexpr (TPX > 64) {
    const int wave = threadIdx.x >> 6, gw0 = (threadIdx.x / TPX) * (TPX / 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < TPX / 64; ++w) t += sh[gw0 + w];
    v = t;
  }
  return v;
}

struct ScoreIdx { float s; int i; };
__device__ __forceinline__ ScoreIdx better(ScoreIdx a, ScoreIdx b) {
  return (b.s > a.s || (b.s == a.s && b.i < a.i)) ? b : a;     // first maximum wins
}
template <int TPX>
__device__ __forceinline__ ScoreIdx group_argmax(ScoreIdx v, ScoreIdx* sh /* [wg_waves] */) {
  constexpr int W = TPX < 64 ? TPX : 64;
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1) {
    ScoreIdx o{__shfl_xor(v.s, off, 64), __shfl_xor(v.i, off, 64)};
    v = better(v, o);
  }
  if constexpr (TPX > 64) {
    const int wave = threadIdx.x >> 6, gw0 = (threadIdx.x / TPX) * (TPX / 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    ScoreIdx t = sh[gw0];
#pragma unroll
    for (int w = 1; w < TPX / 64; ++w) t = better(t, sh[gw0 + w]);
    v = t;
  }
  return v;
}

__host__ __device__ inline unsigned align16(size_t v) { return (unsigned)((v + 15) & ~size_t(15)); }

template <typename T>
static void chain_layout(ChainParams<T>& P, int nfft, int bps) {
  unsigned b = 0;
  b += align16(sizeof(cx<T>) * (size_t)fft_lds_elems(nfft));           // fft exchange (offset 0)
  P.off_y = b;      b += align16(sizeof(cx<T>) * (size_t)P.np);
  P.off_c0 = b;     b += align16(sizeof(cx<T>) * (size_t)P.k_atoms);
  P.off_xs = b;     b += align16(sizeof(c64) * CH_MAXT * 2);           // x, b
  P.off_picks = b;  b += align16(sizeof(int) * (CH_MAXT + 4));         // picks, n, stop, rho
  P.off_gram = b;   b += align16(sizeof(c64) * (size_t)P.taps * P.taps);
  P.off_codes = b;  b += align16((size_t)P.nd * P.n_symb);             // decided symbol codes (bps <= 8)
  P.group_bytes = b;
  (void)bps;
}

template <typename T, int N>
__global__ __launch_bounds__(fft_wg_threads(N)) void rx_chain_kernel(ChainParams<T> P, const cx<T>* __restrict__ rx,
                                                                     int64_t n_frames,
                                                                     uint32_t* __restrict__ bits_out,
                                                                     const uint32_t* __restrict__ ref_bits,
                                                                     uint32_t* __restrict__ errors_out,
                                                                     cx<T>* __restrict__ h_out,
                                                                     int32_t* __restrict__ index_out) {
  constexpr int TPX = N / 8;
  constexpr int FPW = fft_xforms_per_wg(N);
  constexpr int WGW = fft_wg_threads(N) / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double sh_sum[WGW];
  __shared__ ScoreIdx sh_arg[WGW];
  const int g = threadIdx.x / TPX;
  const int j = threadIdx.x % TPX;
  const int64_t frame = (int64_t)blockIdx.x * FPW + g;
  const bool live = frame < n_frames;
  // ---- carve the group's LDS
  unsigned char* base = smem + (size_t)g * P.group_bytes;
  cx<T>* lfft = (cx<T>*)base;
  cx<T>* Y = (cx<T>*)(base + P.off_y);
  cx<T>* c0 = (cx<T>*)(base + P.off_c0);
  c64* xs = (c64*)(base + P.off_xs);
  c64* bs = xs + CH_MAXT;
  int* picks = (int*)(base + P.off_picks);
  int* ctl = picks + CH_MAXT;
  c64* Gm = (c64*)(base + P.off_gram);
  uint8_t* codes = base + P.off_codes;
  const int ldg = P.taps;
  const int L = N + P.t_guard;
  const cx<T>* frx = rx + (live ? frame : 0) * (int64_t)L * P.n_symb;

  // ================= symbol 1: demodulate, pilots, OMP ==========================================
  cx<T> v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = live ? frx[P.t_guard + j + e * TPX] : mk<T>(0, 0);
  wg_fft<T, N, false>(v, j, P.tw, lfft);
  // Y = X(pilotCarriers,1) ./ pilotValues(:,1)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = j + e * TPX;
    const int p = P.prole[k];
    if (p >= 0) Y[p] = cdiv(v[e], P.pilots[p]);
  }
  if (j == 0) { ctl[0] = 0; ctl[1] = 0; }      // ctl[0] = picks made, ctl[1] = stopped
  __syncthreads();
  // c0 = S^H y ; ||y||^2
  double ynorm = 0;
  for (int p = j; p < P.np; p += TPX) ynorm += (double)Y[p].x * Y[p].x + (double)Y[p].y * Y[p].y;
  ynorm = group_sum<TPX>(ynorm, sh_sum);
  for (int k = j; k < P.k_atoms; k += TPX) {
    cx<T> acc = mk<T>(0, 0);
    for (int p = 0; p < P.np; ++p) acc = acc + P.sct[(size_t)p * P.k_atoms + k] * Y[p];
    c0[k] = acc;
  }
  __syncthreads();
  double rho_prev = ynorm;
  for (int it = 0; it < P.taps; ++it) {
    const int n_prev = ctl[0];
    const bool active = (ctl[1] == 0);
    // residual correlation c = c0 - G(:,index) x  and its arg-max (OMP_estimate.m:7,:14)
    ScoreIdx best{-1.0f, 0x7fffffff};
    for (int k = j; k < P.k_atoms; k += TPX) {
      cx<T> c = c0[k];
      for (int q = 0; q < n_prev; ++q) {
        const int d = picks[q] - k;                       // G[k][index_q] = gram[index_q - k]
        const c64 gq = d >= 0 ? P.gram[d] : conj(P.gram[-d]);
        const c64 t = gq * xs[q];
        c = c - mk<T>((T)t.x, (T)t.y);
      }
      const float sc = (float)((double)c.x * c.x + (double)c.y * c.y);    // |.|^2 is monotone in |.|
      best = better(best, ScoreIdx{sc, k});
    }
    best = group_argmax<TPX>(best, sh_arg);
    const int kp = best.i < P.k_atoms ? best.i : 0;      // all-NaN scores: MATLAB max returns index 1
    // b_it = a_kp^H y in double
    double br = 0, bi = 0;
    for (int p = j; p < P.np; p += TPX) {
      const cx<T> a = P.sct[(size_t)p * P.k_atoms + kp];    // conj(S[p,kp])
      br += (double)a.x * Y[p].x - (double)a.y * Y[p].y;
      bi += (double)a.x * Y[p].y + (double)a.y * Y[p].x;
    }
    br = group_sum<TPX>(br, sh_sum);
    bi = group_sum<TPX>(bi, sh_sum);
    if (j == 0 && active) {
      int dup = -1;
      for (int q = 0; q < n_prev; ++q) if (picks[q] == kp) dup = q;
      if (dup >= 0) {
        // pinv of a matrix with a repeated column: coefficient split equally, residual unchanged -> break
        const c64 half{xs[dup].x * 0.5, xs[dup].y * 0.5};
        xs[dup] = half; xs[n_prev] = half; picks[n_prev] = kp;
        ctl[0] = n_prev + 1; ctl[1] = 1;
      } else {
        picks[n_prev] = kp;
        bs[n_prev] = c64{br, bi};
        // Gram of the selected atoms: G[a][b] = a_a^H a_b = gram[idx_b - idx_a]; lower triangle,
        // factored in place (Cholesky), then the two triangular solves.
        const int n = n_prev + 1;
        for (int a = 0; a < n; ++a)
          for (int b2 = 0; b2 <= a; ++b2) {
            const int d = picks[b2] - picks[a];             // G[a][b2] = a_a^H a_b2 = gram[idx_b2 - idx_a]
            Gm[a * ldg + b2] = d >= 0 ? P.gram[d] : conj(P.gram[-d]);
          }
        // in-place Cholesky (lower), then the two triangular solves
        for (int c = 0; c < n; ++c) {
          double dd = Gm[c * ldg + c].x;
          for (int k2 = 0; k2 < c; ++k2) dd -= norm2(Gm[c * ldg + k2]);
          const double ljj = sqrt(dd);
          Gm[c * ldg + c] = c64{ljj, 0};
          for (int r = c + 1; r < n; ++r) {
            c64 s = Gm[r * ldg + c];
            for (int k2 = 0; k2 < c; ++k2) s = s - mulc(Gm[r * ldg + k2], Gm[c * ldg + k2]);
            Gm[r * ldg + c] = c64{s.x / ljj, s.y / ljj};
          }
        }
        for (int r = 0; r < n; ++r) {
          c64 s = bs[r];
          for (int k2 = 0; k2 < r; ++k2) s = s - Gm[r * ldg + k2] * xs[k2];
          const double l = Gm[r * ldg + r].x;
          xs[r] = c64{s.x / l, s.y / l};
        }
        for (int r = n - 1; r >= 0; --r) {
          c64 s = xs[r];
          for (int k2 = r + 1; k2 < n; ++k2) s = s - mulc(xs[k2], Gm[k2 * ldg + r]);
          const double l = Gm[r * ldg + r].x;
          xs[r] = c64{s.x / l, s.y / l};
        }
        // ||r_i||^2 = ||y||^2 - Re(b^H x)
        double proj = 0;
        for (int r = 0; r < n; ++r) proj += bs[r].x * xs[r].x + bs[r].y * xs[r].y;
        const double rho = ynorm - proj;
        ctl[0] = n;
        // OMP_estimate.m:20: norm(r_i - r_{i-1})/norm(r_{i-1}) < 1e-2, only inside the i1 >= 2 loop
        if (it >= 1) {
          const double num = rho_prev - rho;
          if (!(num > 0.0) || sqrt(num / rho_prev) < 1e-2) ctl[1] = 1;
        }
        ((double*)(ctl + 2))[0] = rho;
      }
    }
    __syncthreads();
    rho_prev = ((double*)(ctl + 2))[0];
  }
  const int n_picks = ctl[0];
  if (live && index_out && j < P.taps) index_out[frame * P.taps + j] = j < n_picks ? picks[j] + 1 : 0;

  // ================= equaliser taps: G[k] = 1 / H[k], H = fft(h) restricted to 1..N_carrier ======
  cx<T> geq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = j + e * TPX;
    geq[e] = mk<T>(0, 0);
    if (k < P.n_carrier) {
      double hr = 0, hi = 0;
      // est_fade_chan(index(i1)) = x(i1): a later duplicate overwrites an earlier one (:31-33)
      for (int q = 0; q < n_picks; ++q) {
        bool overwritten = false;
        for (int q2 = q + 1; q2 < n_picks; ++q2) overwritten |= (picks[q2] == picks[q]);
        if (overwritten) continue;
        const cx<T> w = P.tw[(int)(((int64_t)picks[q] * k) & (N - 1))];
        hr += xs[q].x * (double)w.x - xs[q].y * (double)w.y;
        hi += xs[q].x * (double)w.y + xs[q].y * (double)w.x;
      }
      const cx<T> H = mk<T>((T)hr, (T)hi);
      if (live && h_out) h_out[frame * P.n_carrier + k] = H;
      geq[e] = cdiv(mk<T>(1, 0), H);
    }
  }

  // ================= all symbols: equalise, payload, demap =======================================
  const int bps = P.tab.bps;
  for (int s = 0; s < P.n_symb; ++s) {
    if (s > 0) {
      const cx<T>* src = frx + (int64_t)s * L + P.t_guard;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = live ? src[j + e * TPX] : mk<T>(0, 0);
      wg_fft<T, N, false>(v, j, P.tw, lfft);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = j + e * TPX;
      if (k < P.n_carrier) {
        const int d = P.drole[k];
        if (d >= 0) codes[s * P.nd + d] = (uint8_t)demap_decide(P.tab, v[e] * geq[e]);
      }
    }
  }
  __syncthreads();

  // ================= pack bits (MSB-first inside each byte) + BER numerator =======================
  const int64_t frame_bits = (int64_t)P.nd * P.n_symb * bps;
  unsigned int err = 0;
  for (int w = j; w < P.frame_words; w += TPX) {
    uint32_t word = 0;
    const int64_t b0 = (int64_t)w * 32;
    for (int b = 0; b < 32; ++b) {
      const int64_t i = b0 + b;
      if (i < frame_bits) {
        const int q = (int)(i / bps), r = (int)(i % bps);
        const uint32_t bit = (codes[q] >> (bps - 1 - r)) & 1u;
        word |= bit << ((b & ~7) + 7 - (b & 7));
      }
    }
    if (live) {
      if (bits_out) bits_out[frame * P.frame_words + w] = word;
      if (ref_bits) err += __popc(word ^ ref_bits[frame * P.frame_words + w]);   // padding bits are 0 in both
    }
  }
  if (ref_bits && errors_out) {
    const double tot = group_sum<TPX>((double)err, sh_sum);
    if (live && j == 0) errors_out[frame] = (uint32_t)tot;
  }
}


// ---------------------------------------------------------------------------------------------
// Per-frame DeScrambler as a pass over the packed decisions (T5/DeScrambler.m:8-13, register reset per frame,
// T5/Main_model_Task_5.m:257-274) for every path whose pack stage does not descramble itself -- all but the wave-per-frame
// symbol kernel: the chain writes its raw decisions to a plan workspace, this pass writes the descrambled words to the
// caller's bits_out and counts the errors against the TX's input bits.  One workgroup per frame, nothing atomic.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void descr_pass_kernel(const uint32_t* __restrict__ raw, uint32_t* __restrict__ out,
                                                         const uint32_t* __restrict__ ref, uint32_t* __restrict__ errs,
                                                         int frame_words, int64_t frame_bits, uint32_t descr) {
  const int64_t f = blockIdx.x;
  const uint32_t* r = raw + f * frame_words;
  unsigned err = 0;
  for (int w = threadIdx.x; w < frame_words; w += 256) {
    const uint32_t cur = __builtin_bswap32(r[w]);                     // stream order: MSB first
    const uint32_t prev = w ? __builtin_bswap32(r[w - 1]) : descr;
    const int64_t valid = frame_bits - (int64_t)w * 32;
    const uint32_t d = descr_word(cur, prev) & (valid >= 32 ? 0xffffffffu : ~(0xffffffffu >> (int)valid));
    const uint32_t o = __builtin_bswap32(d);
    if (out) out[f * frame_words + w] = o;
    if (ref) err += __popc(o ^ ref[f * frame_words + w]);
  }
  if (ref && errs) {
    for (int off = 32; off > 0; off >>= 1) err += __shfl_xor(err, off, 64);
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = err;
    __syncthreads();
    if (threadIdx.x == 0) errs[f] = part[0] + part[1] + part[2] + part[3];
  }
}

// workspace for the raw decisions of `n_frames` frames (plan-owned, grown on demand)
int descr_raw_workspace(ofdm_rx_plan* pl, int64_t n_frames, void** raw) {
  const size_t need = (size_t)pl->frame_words * 4 * (size_t)n_frames;
  if (pl->ws_raw_bytes < need) {
    OFDM_HIP(hipStreamSynchronize(ctx().stream));
    if (pl->ws_raw) { (void)hipFree(pl->ws_raw); pl->ws_raw = nullptr; pl->ws_raw_bytes = 0; }
    OFDM_HIP(hipMalloc(&pl->ws_raw, need));
    pl->ws_raw_bytes = need;
  }
  *raw = pl->ws_raw;
  return OFDM_OK;
}

int descr_pass_run(ofdm_rx_plan* pl, const void* raw, void* bits, const void* ref, void* errs, int64_t n_frames) {
  if (n_frames == 0) return OFDM_OK;
  hipLaunchKernelGGL(descr_pass_kernel, dim3((unsigned)n_frames), dim3(256), 0, ctx().stream, (const uint32_t*)raw,
                     (uint32_t*)bits, (const uint32_t*)ref, (uint32_t*)errs, pl->frame_words,
                     (int64_t)pl->nd * pl->n_symb * pl->bps, pl->descr);
  return check_launch("descr_pass_kernel");
}

}  // namespace ofdm

using namespace ofdm;

namespace ofdm {
bool chain_fast_supported(int nfft, int n_carrier, int taps, int bps, int64_t nd_nsymb);     // ofdm_chain_fast.hip
int mmse_build_operator(const c64* h, int64_t n_h, double snr_db, const int32_t* pilot_loc, int np, int n_carrier,
                        int m_pad, std::vector<c64>& wt);                                  // ofdm_chain_mmse.hip
bool chain_split_supported(int nfft, int n_carrier, int taps, int bps, int64_t nd_nsymb, bool f64);   // ofdm_chain_split.hip
int chain_split_run(const FastPlanView& pv, const void* tw, const void* rx, int64_t n_frames, void* bits,
                    const void* ref, void* errs, void* h_out, void* idx_out, const int32_t* d_pc0);
int chain_fast_run(const FastPlanView& pv, const void* tw, const void* rx, int64_t n_frames, void* bits,
                   const void* ref, void* errs, void* h_out, void* idx_out);
}  // namespace ofdm

constexpr size_t GENERIC_LDS_LIMIT = 158 * 1024;
// dynamic LDS the generic single kernel would ask for (mirrors chain_layout + launch_chain)
static size_t generic_lds_bytes(const ofdm_rx_plan* pl) {
  const size_t cs = pl->f64 ? sizeof(c64) : sizeof(c32);
  size_t b = 0;
  b += align16(cs * (size_t)fft_lds_elems(pl->nfft));
  b += align16(cs * (size_t)pl->np);
  b += align16(cs * (size_t)pl->k_atoms);
  b += align16(sizeof(c64) * CH_MAXT * 2);
  b += align16(sizeof(int) * (CH_MAXT + 4));
  b += align16(sizeof(c64) * (size_t)pl->taps * pl->taps);
  b += align16((size_t)pl->nd * pl->n_symb);
  return b * (size_t)fft_xforms_per_wg(pl->nfft);
}

template <typename T, int N>
static int launch_chain(const ofdm_rx_plan* pl, const void* tw, const void* rx, int64_t n_frames, void* bits,
                        const void* ref, void* errs, void* h_out, void* idx_out) {
  ChainParams<T> P;
  P.n_symb = pl->n_symb; P.t_guard = pl->t_guard; P.n_carrier = pl->n_carrier; P.np = pl->np; P.nd = pl->nd;
  P.k_atoms = pl->k_atoms; P.taps = pl->taps;
  P.prole = (const int16_t*)pl->d_prole; P.drole = (const int16_t*)pl->d_drole;
  P.pilots = (const cx<T>*)pl->d_pilots; P.sct = (const cx<T>*)pl->d_sct; P.gram = (const c64*)pl->d_gram;
  P.pc0 = (const int32_t*)pl->d_pc0; P.tw = (const cx<T>*)tw;
  fill_demap_table<T>(pl->dict, pl->cinfo, P.tab);
  P.frame_words = pl->frame_words;
  constexpr int FPW = fft_xforms_per_wg(N);
  chain_layout<T>(P, N, pl->bps);
  const size_t dyn = (size_t)P.group_bytes * FPW;
  OFDM_ARG(dyn <= GENERIC_LDS_LIMIT, "rx_chain_task5: configuration needs %zu bytes of LDS (limit 158 KiB; use fp32 or a smaller frame)", dyn);
  OFDM_HIP(hipFuncSetAttribute((const void*)rx_chain_kernel<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  hipLaunchKernelGGL((rx_chain_kernel<T, N>), dim3(cdiv_u(n_frames, FPW)), dim3(fft_wg_threads(N)), dyn, ctx().stream,
                     P, (const cx<T>*)rx, n_frames, (uint32_t*)bits, (const uint32_t*)ref, (uint32_t*)errs,
                     (cx<T>*)h_out, (int32_t*)idx_out);
  return check_launch("rx_chain_kernel");
}

extern "C" {

int ofdm_rx_plan_create(ofdm_rx_plan** plan_out, int nfft, int t_guard, int n_symb, int n_carrier,
                        const int32_t* pilot_carriers, int n_pilots, const int32_t* data_carriers, int n_data,
                        const void* pilot_values_col, int k_atoms, int dominant_taps, const char* constellation,
                        int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(plan_out, "rx_plan_create: null output");
  OFDM_ARG(fft_size_supported(nfft), "rx_plan_create: unsupported Nfft %d", nfft);
  OFDM_ARG(t_guard >= 0 && n_symb >= 1 && n_carrier >= 1 && n_carrier <= nfft, "rx_plan_create: bad frame geometry");
  // n_data = 0: a pilots-only plan (T5/Main_model_Task_5.m as committed, comb = 1) -- for ofdm_task5_mse_tile; the decoding
  // entries refuse it
  OFDM_ARG(n_pilots >= 1 && n_pilots < 32768 && n_data >= 0 && n_data < 32768, "rx_plan_create: bad carrier counts");
  OFDM_ARG(k_atoms >= 1 && k_atoms <= nfft && dominant_taps >= 1 && dominant_taps <= CH_MAXT && dominant_taps <= k_atoms,
           "rx_plan_create: needs 1 <= taps <= %d, taps <= K <= Nfft", CH_MAXT);
  OFDM_ARG(dominant_taps <= nfft / 8, "rx_plan_create: taps exceed the group size");
  ofdm_rx_plan* pl = new ofdm_rx_plan();
  pl->device = ctx().device;
  pl->nfft = nfft; pl->t_guard = t_guard; pl->n_symb = n_symb; pl->n_carrier = n_carrier; pl->np = n_pilots;
  pl->nd = n_data; pl->k_atoms = k_atoms; pl->taps = dominant_taps; pl->f64 = is_f64(flags) ? 1 : 0;
  pl->d_prole = pl->d_drole = pl->d_pilots = pl->d_sct = pl->d_gram = pl->d_pc0 = nullptr;
  pl->pilots_in_band = 1;
  if (!constellation_info(constellation, pl->cinfo)) {
    delete pl;
    set_error("rx_plan_create: unknown constellation '%s'", constellation ? constellation : "(null)");
    return OFDM_ERR_ARG;
  }
  constellation_table(constellation, pl->dict);
  pl->bps = pl->cinfo.bps;
  const int64_t frame_bits = (int64_t)n_data * n_symb * pl->bps;
  pl->frame_words = (int)((frame_bits + 31) / 32);
  std::vector<int16_t> prole(nfft, -1), drole(nfft, -1);
  std::vector<int32_t> pc0(n_pilots);
  int rc = OFDM_OK, mod4 = 0;
  for (int d = 0; d < n_data && rc == OFDM_OK; ++d) {
    if (data_carriers[d] < 1 || data_carriers[d] > n_carrier) { set_error("rx_plan_create: data carrier outside 1..N_carrier"); rc = OFDM_ERR_ARG; break; }
    drole[data_carriers[d] - 1] = (int16_t)d;
    mod4 |= 1 << ((data_carriers[d] - 1) & 3);
  }
  pl->data_mod4 = mod4;
  for (int p = 0; p < n_pilots && rc == OFDM_OK; ++p) {
    if (pilot_carriers[p] < 1 || pilot_carriers[p] > nfft) { set_error("rx_plan_create: pilot carrier outside 1..Nfft"); rc = OFDM_ERR_ARG; break; }
    pc0[p] = pilot_carriers[p] - 1;
    pl->pilot_loc.push_back(pilot_carriers[p]);
    prole[pc0[p]] = (int16_t)p;
    if (pilot_carriers[p] > n_carrier) pl->pilots_in_band = 0;
  }
  if (rc != OFDM_OK) { delete pl; return rc; }
  // comb layout (T5/Main_model_Task_5.m:18-22: pilotCarriers = 1 : comb : N_carrier)?  Then S^H Y is an inverse
  // transform of size Nfft/comb and the fast path needs no dictionary correlation.
  if (n_pilots >= 2 && pc0[0] == 0) {
    const int comb = pc0[1] - pc0[0];
    bool is_comb = comb >= 1 && nfft % comb == 0;
    for (int p = 0; p < n_pilots && is_comb; ++p) is_comb = pc0[p] == comb * p;
    const int m = is_comb ? nfft / comb : 0;
    if (is_comb && n_pilots <= m) pl->comb_m = m;
    if (is_comb && m <= 512 && 512 % m == 0 && n_pilots <= m) {
      int lg = 0;
      while ((m << lg) < 512) ++lg;
      pl->comb_lg_up = lg;
    }
  }
  // conj(S) transposed: sct[p][k] = exp(+2 pi i pc0[p] k / nfft); Gram table g[d] = sum_p exp(-2 pi i pc0[p] d / nfft)
  const size_t cs = pl->f64 ? sizeof(c64) : sizeof(c32);
  std::vector<c64> sct((size_t)n_pilots * k_atoms), gram(k_atoms);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int p = 0; p < n_pilots; ++p)
    for (int k = 0; k < k_atoms; ++k) {
      const int64_t ph = ((int64_t)pc0[p] * k) % nfft;
      const long double a = two_pi * (long double)ph / (long double)nfft;
      sct[(size_t)p * k_atoms + k] = c64{(double)cosl(a), (double)sinl(a)};
    }
  for (int d = 0; d < k_atoms; ++d) {
    long double sr = 0, si = 0;
    for (int p = 0; p < n_pilots; ++p) {
      const int64_t ph = ((int64_t)pc0[p] * d) % nfft;
      const long double a = two_pi * (long double)ph / (long double)nfft;
      sr += cosl(a);
      si -= sinl(a);
    }
    gram[d] = c64{(double)sr, (double)si};
  }
  auto to_dev = [&](const void* src, size_t bytes, void** dst) -> int {
    OFDM_HIP(hipMalloc(dst, bytes ? bytes : 16));
    OFDM_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return OFDM_OK;
  };
  rc = to_dev(prole.data(), sizeof(int16_t) * nfft, &pl->d_prole);
  if (rc == OFDM_OK) rc = to_dev(drole.data(), sizeof(int16_t) * nfft, &pl->d_drole);
  if (rc == OFDM_OK) rc = to_dev(pc0.data(), sizeof(int32_t) * n_pilots, &pl->d_pc0);
  if (rc == OFDM_OK) rc = to_dev(gram.data(), sizeof(c64) * k_atoms, &pl->d_gram);
  if (rc == OFDM_OK) rc = to_dev(pilot_values_col, cs * n_pilots, &pl->d_pilots);
  if (rc == OFDM_OK) {
    if (pl->f64) rc = to_dev(sct.data(), sizeof(c64) * sct.size(), &pl->d_sct);
    else {
      std::vector<c32> s32(sct.size());
      for (size_t i = 0; i < sct.size(); ++i) s32[i] = c32{(float)sct[i].x, (float)sct[i].y};
      rc = to_dev(s32.data(), sizeof(c32) * s32.size(), &pl->d_sct);
    }
  }
  if (rc != OFDM_OK) { ofdm_rx_plan_destroy(pl); return rc; }
  ctx().live_plans += 1;
  *plan_out = pl;
  return OFDM_OK;
}

int ofdm_rx_plan_destroy(ofdm_rx_plan* pl) {
  if (!pl) return OFDM_OK;
  void* ptrs[] = {pl->d_prole, pl->d_drole, pl->d_pilots, pl->d_sct, pl->d_gram, pl->d_pc0,
                  pl->ws_stash, pl->ws_ypil, pl->ws_tapidx, pl->ws_tapx, pl->ws_h, pl->d_wt, pl->ws_x,
                  pl->ws_gen, pl->d_dict, pl->ws_t4, pl->d_t4_tx, pl->d_t4_w, pl->d_p2_sop, pl->ws_raw,
                  pl->d_mt, pl->d_sb_w, pl->d_sb_c0, pl->ws_v, pl->d_t4_bw, pl->d_t4_bc0};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (auto& e : pl->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : pl->ev_t4) if (e) (void)hipEventDestroy(e);
  if (ctx().ready && ctx().device == pl->device && ctx().live_plans > 0) ctx().live_plans -= 1;
  delete pl;
  return OFDM_OK;
}

// Per-frame DeScrambler of the fused receivers (T5/Main_model_Task_5.m:257-274, T4/Main_model_Task_4.m:354-364)
int ofdm_rx_plan_set_descrambler(ofdm_rx_plan* pl, const uint8_t* reg15) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(pl, "rx_plan_set_descrambler: null plan");
  uint32_t d = 0;
  if (reg15) {
    d = DESCR_ON;
    for (int m = 1; m <= 14; ++m) {            // Register(15) never reaches a tap before it is shifted out (array_xor: 13, 14)
      OFDM_ARG(reg15[m - 1] <= 1, "rx_plan_set_descrambler: register entries must be 0 or 1");
      d |= (uint32_t)reg15[m - 1] << (m - 1);
    }
    OFDM_ARG(reg15[14] <= 1, "rx_plan_set_descrambler: register entries must be 0 or 1");
  }
  pl->descr = d;
  return OFDM_OK;
}

int ofdm_rx_plan_set_mmse(ofdm_rx_plan* pl, const void* h, int64_t n_h, double snr_db, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(pl, "rx_plan_set_mmse: null plan");
  auto drop_factors = [&]() {
    for (void** q : {&pl->d_mt, &pl->d_sb_w, &pl->d_sb_c0}) if (*q) { (void)hipFree(*q); *q = nullptr; }
  };
  if (!h || n_h <= 0) {                                   // back to OMP mode
    OFDM_HIP(hipStreamSynchronize(ctx().stream));
    if (pl->d_wt) { (void)hipFree(pl->d_wt); pl->d_wt = nullptr; }
    drop_factors();
    return OFDM_OK;
  }
  OFDM_ARG((flags & OFDM_DEVICE) == 0, "rx_plan_set_mmse: h is a host array (the operator is built on the host)");
  OFDM_ARG(pl->np >= 2, "rx_plan_set_mmse: MMSE_CE needs at least two pilots (MMSE_CE.m:15)");
  std::vector<c64> hh((size_t)n_h);
  if (is_f64(flags)) std::memcpy(hh.data(), h, sizeof(c64) * (size_t)n_h);
  else for (int64_t i = 0; i < n_h; ++i) hh[i] = c64{(double)((const c32*)h)[i].x, (double)((const c32*)h)[i].y};
  const int m_pad = (pl->n_carrier + 15) & ~15;
  std::vector<c64> wt, mt;
  std::vector<double> sop;
  const int np_pad = (pl->np + 15) & ~15;
  const bool factored = !pl->f64 && mmse_factored_usable(pl->np, np_pad);
  OFDM_TRY(mmse_build_operator(hh.data(), n_h, snr_db, pl->pilot_loc.data(), pl->np, pl->n_carrier, m_pad, wt,
                               factored ? &mt : nullptr, np_pad, factored ? &sop : nullptr));
  OFDM_HIP(hipStreamSynchronize(ctx().stream));
  if (pl->d_wt) { (void)hipFree(pl->d_wt); pl->d_wt = nullptr; }
  drop_factors();
  if (factored) {                                        // fp32: H = Sop_banded * (M * Y), ofdm_chain_mmse.hip
    std::vector<c32> m32(mt.size());
    for (size_t i = 0; i < mt.size(); ++i) m32[i] = c32{(float)mt[i].x, (float)mt[i].y};
    std::vector<float> bw_w;
    std::vector<int32_t> bw_c0;
    int bw = 0, span = 0;
    mmse_band_spline(sop, pl->n_carrier, pl->np, bw_w, bw_c0, bw, span);
    OFDM_HIP(hipMalloc(&pl->d_mt, sizeof(c32) * m32.size()));
    OFDM_HIP(hipMemcpy(pl->d_mt, m32.data(), sizeof(c32) * m32.size(), hipMemcpyHostToDevice));
    OFDM_HIP(hipMalloc(&pl->d_sb_w, sizeof(float) * bw_w.size()));
    OFDM_HIP(hipMemcpy(pl->d_sb_w, bw_w.data(), sizeof(float) * bw_w.size(), hipMemcpyHostToDevice));
    OFDM_HIP(hipMalloc(&pl->d_sb_c0, sizeof(int32_t) * bw_c0.size()));
    OFDM_HIP(hipMemcpy(pl->d_sb_c0, bw_c0.data(), sizeof(int32_t) * bw_c0.size(), hipMemcpyHostToDevice));
    pl->np_pad = np_pad;
    pl->sb_bw = bw;
    pl->sb_span = span;
  }
  if (pl->f64) {
    OFDM_HIP(hipMalloc(&pl->d_wt, sizeof(c64) * wt.size()));
    OFDM_HIP(hipMemcpy(pl->d_wt, wt.data(), sizeof(c64) * wt.size(), hipMemcpyHostToDevice));
  } else {
    std::vector<c32> w32(wt.size());
    for (size_t i = 0; i < wt.size(); ++i) w32[i] = c32{(float)wt[i].x, (float)wt[i].y};
    OFDM_HIP(hipMalloc(&pl->d_wt, sizeof(c32) * w32.size()));
    OFDM_HIP(hipMemcpy(pl->d_wt, w32.data(), sizeof(c32) * w32.size(), hipMemcpyHostToDevice));
  }
  pl->m_pad = m_pad;
  return OFDM_OK;
}

int ofdm_rx_plan_set_timing(ofdm_rx_plan* pl, int enable) {
  OFDM_ARG(pl, "rx_plan_set_timing: null plan");
  if (enable && !pl->ev[0]) {
    for (int i = 0; i < 4; ++i) OFDM_HIP(hipEventCreate(&pl->ev[i]));
    for (int i = 0; i < 6; ++i) OFDM_HIP(hipEventCreate(&pl->ev_t4[i]));
  }
  pl->timing = enable ? 1 : 0;
  return OFDM_OK;
}

int ofdm_rx_plan_last_kernel_ms(ofdm_rx_plan* pl, float* ms3) {
  OFDM_ARG(pl && ms3 && pl->timing && pl->ev[0], "rx_plan_last_kernel_ms: timing is not enabled");
  OFDM_HIP(hipEventSynchronize(pl->ev[3]));
  if (pl->last_fast && pl->last_fused) {
    // rx_pilot_omp_kernel | (no separate OMP launch) | rx_symbols_kernel
    OFDM_HIP(hipEventElapsedTime(&ms3[0], pl->ev[0], pl->ev[1]));
    ms3[1] = 0.f;
    OFDM_HIP(hipEventElapsedTime(&ms3[2], pl->ev[1], pl->ev[3]));
  } else if (pl->last_fast) {
    for (int i = 0; i < 3; ++i) OFDM_HIP(hipEventElapsedTime(&ms3[i], pl->ev[i], pl->ev[i + 1]));
  } else {
    ms3[0] = ms3[1] = 0.f;
    OFDM_HIP(hipEventElapsedTime(&ms3[2], pl->ev[0], pl->ev[3]));
  }
  return OFDM_OK;
}

int ofdm_rx_plan_last_task4_ms(ofdm_rx_plan* pl, float* ms5) {
  OFDM_ARG(pl && ms5 && pl->timing && pl->ev_t4[0] && pl->t4_timed, "rx_plan_last_task4_ms: no timed ofdm_rx_chain_task4 call");
  OFDM_HIP(hipEventSynchronize(pl->ev_t4[5]));
  for (int i = 0; i < 5; ++i) OFDM_HIP(hipEventElapsedTime(&ms5[i], pl->ev_t4[i], pl->ev_t4[i + 1]));
  return OFDM_OK;
}

int64_t ofdm_rx_plan_frame_bytes(const ofdm_rx_plan* pl) { return pl ? (int64_t)pl->frame_words * 4 : 0; }

int ofdm_rx_chain_task5(ofdm_rx_plan* pl, const void* rx, int64_t n_frames, uint8_t* bits_out,
                        const uint8_t* ref_bits, uint32_t* errors_out, void* h_out, int32_t* index_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(pl && rx && n_frames >= 0, "rx_chain_task5: bad arguments");
  OFDM_PLAN_DEVICE(pl);
  OFDM_ARG((is_f64(flags) ? 1 : 0) == pl->f64, "rx_chain_task5: precision flag differs from the plan's");
  OFDM_ARG(!errors_out || ref_bits, "rx_chain_task5: errors_out needs ref_bits");
  OFDM_ARG(pl->nd >= 1, "rx_chain_task5: the plan has no data carriers");
  if (n_frames == 0) return OFDM_OK;
  const size_t cs = csize(flags);
  const size_t frame_samples = (size_t)(pl->nfft + pl->t_guard) * pl->n_symb;
  const size_t fb = (size_t)pl->frame_words * 4;
  Stage st(flags);
  const void *drx, *dref; void *dbits, *derr, *dh, *didx;
  OFDM_TRY(st.in(rx, cs * frame_samples * n_frames, &drx));
  OFDM_TRY(st.in(ref_bits, fb * n_frames, &dref));
  OFDM_TRY(st.out(bits_out, fb * n_frames, &dbits));
  OFDM_TRY(st.out(errors_out, sizeof(uint32_t) * n_frames, &derr));
  OFDM_TRY(st.out(h_out, cs * (size_t)pl->n_carrier * n_frames, &dh));
  OFDM_TRY(st.out(index_out, sizeof(int32_t) * (size_t)pl->taps * n_frames, &didx));
  const void* tw = nullptr;
  OFDM_TRY(get_twiddles(pl->nfft, pl->f64 != 0, &tw));
  const bool fast = pl->pilots_in_band &&
                    chain_fast_supported(pl->nfft, pl->n_carrier, pl->taps, pl->bps, (int64_t)pl->nd * pl->n_symb);
  // split form: Nfft beyond the wave-local fast path, a frame state that does not fit the generic kernel's LDS, or
  // MMSE mode outside the fast path
  bool split = false;
  if (!fast && pl->pilots_in_band &&
      chain_split_supported(pl->nfft, pl->n_carrier, pl->taps, pl->bps, (int64_t)pl->nd * pl->n_symb, pl->f64 != 0))
    split = pl->nfft > 4096 || pl->d_wt != nullptr || generic_lds_bytes(pl) > GENERIC_LDS_LIMIT;
  // DeScrambler of the plan: fused into the pack stage of the wave-per-frame symbol kernel; every other path hands its raw
  // decisions to descr_pass_kernel (the stages themselves then neither compare nor count)
  void* craw = nullptr;
  bool descr_pass = false;
  if (fast || split) {
    FastPlanView pv;
    make_plan_view(pl, pv);
    pl->last_fast = 1;
    descr_pass = (pl->descr & DESCR_ON) && !(fast && chain_wave_supported(pv));
    if (descr_pass) {
      OFDM_TRY(descr_raw_workspace(pl, n_frames, &craw));
      pv.descr = 0;
    }
    void* cb = descr_pass ? craw : dbits;
    const void* cr = descr_pass ? nullptr : dref;
    void* ce = descr_pass ? nullptr : derr;
    if (fast) OFDM_TRY(chain_fast_run(pv, tw, drx, n_frames, cb, cr, ce, dh, didx));
    else OFDM_TRY(chain_split_run(pv, tw, drx, n_frames, cb, cr, ce, dh, didx, (const int32_t*)pl->d_pc0));
    if (descr_pass) OFDM_TRY(descr_pass_run(pl, craw, dbits, dref, derr, n_frames));
    return st.finish();
  }
  descr_pass = (pl->descr & DESCR_ON) != 0;
  if (descr_pass) OFDM_TRY(descr_raw_workspace(pl, n_frames, &craw));
  OFDM_ARG(!pl->d_wt, "rx_chain_task5: the MMSE mode of a plan needs pilots inside 1..N_carrier, at most 32 taps and a frame "
                      "whose decisions fit the workgroup's LDS (ofdm_MMSE_CE covers every other case)");
  pl->last_fast = 0;
  if (pl->timing) OFDM_HIP(hipEventRecord(pl->ev[0], ctx().stream));
  {
    void* cb = descr_pass ? craw : dbits;
    const void* cr = descr_pass ? nullptr : dref;
    void* ce = descr_pass ? nullptr : derr;
#define CALL(NN)                                                                                        \
  if (pl->f64) OFDM_TRY((launch_chain<double, NN>(pl, tw, drx, n_frames, cb, cr, ce, dh, didx)));       \
  else OFDM_TRY((launch_chain<float, NN>(pl, tw, drx, n_frames, cb, cr, ce, dh, didx)));
    OFDM_FFT_DISPATCH(pl->nfft, CALL)
#undef CALL
  }
  if (descr_pass) OFDM_TRY(descr_pass_run(pl, craw, dbits, dref, derr, n_frames));
  if (pl->timing) OFDM_HIP(hipEventRecord(pl->ev[3], ctx().stream));
  return st.finish();
}

}  // extern "C"
