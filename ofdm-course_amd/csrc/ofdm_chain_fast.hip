// Fast path of the fused Task-5 RX chain for Nfft = 512 * NW (NW = 1, 2, 4, 8 wavefronts per frame),
// i.e. the metric configuration Nfft = 2048 and its neighbours.  Two or three launches per batch:
//
//   rx_pilot_omp_kernel (ofdm_chain_pilot.hip; comb pilots whose Nfft/comb divides 512, <= 8 taps)
//                      symbol 1 of every frame: FFT -> X1 stash + Y = X(pilotCarriers,1)./pilotValues(:,1)
//                      (Task5_part2.m:190), c0 = S^H Y as one wave-local inverse transform, OMP iterations
//   -- or, for any other pilot layout / tap count --
//   rx_pilot_kernel    symbol 1 of every frame: FFT -> X1 stash and Y
//   omp_batch_kernel   16 frames per workgroup: c0 = S^H Y as ONE real GEMM on the matrix cores
//                      (v_mfma_f32_16x16x4_f32, exact f32), then 4 frames per wavefront run the OMP
//                      iterations (OMP_estimate.m:7-23) side by side in batch form
//   -- then --
//   rx_symbols_kernel  per frame: H = fft(h) on carriers 1..N_carrier from the taps, then every symbol:
//                      FFT -> equalize_signal -> get_payload -> demapping -> packed bits -> BER numerator
//
// FFT structure (CDNA4-specific, chain_fast_core.hpp): one frame = NW wavefronts.  A radix-NW
// decimation-in-frequency stage (registers + one LDS exchange, two workgroup barriers) splits the transform
// into NW independent 512-point transforms, and each of those is done by ONE wavefront: 64 lanes x 8 points
// in registers, three radix-8 passes whose two transposes go through a wave-private LDS region with no
// workgroup barrier at all (a wavefront's LDS operations are processed in order).  Only the bins
// k < N_carrier are ever needed, so the last pass computes 2 of its 8 outputs when N_carrier <= Nfft/4.
// Global loads are 16 bytes per lane (NW = 4); DIF and second-pass twiddles live in registers, third-pass
// twiddles in an LDS table.
#include "chain_fast_core.hpp"
#include "omp_wave_core.hpp"

namespace ofdm {

// ---------------------------------------------------------------------------------------------
// kernel 1: symbol 1 -> stash + Y
// ---------------------------------------------------------------------------------------------
template <typename T, int NW, bool PRUNE2>
__global__ __launch_bounds__(64 * NW, sizeof(T) == 4 ? (PRUNE2 ? 4 : 3) : 2) void rx_pilot_kernel(FastParams<T> P, const cx<T>* __restrict__ rx,
                                                           int64_t n_frames) {
  constexpr int N = 512 * NW;
  constexpr int NOUT = PRUNE2 ? 2 : 8;
  __shared__ cx<T> lwv[NW * WAVE_LDS_ELEMS];
  __shared__ cx<T> twl[WAVE_TW_ELEMS];
  cx<T>* const ex = lwv;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gid = threadIdx.x;
  DifTw<T, NW> dt;
  wave_tw_fill<T, NW>(twl, P.tw);
  dif_tw_init<T, NW>(dt, gid, P.tw);
  cx<T> twb[7];
#pragma unroll
  for (int t = 1; t < 8; ++t) twb[t - 1] = P.tw[(t * (lane & 7) * 8) * NW];
  __syncthreads();
  int kk[NOUT], pp[NOUT];
#pragma unroll
  for (int t = 0; t < NOUT; ++t) {
    kk[t] = NW * (lane + 64 * t) + wave;
    pp[t] = kk[t] < P.n_carrier ? (int)P.prole[kk[t]] : -1;
  }
  const int64_t L = (int64_t)(N + P.t_guard) * P.n_symb;
  cx<T> v[8], nx[8];
  if ((int64_t)blockIdx.x < n_frames) frame_load<T, NW>(nx, rx + (int64_t)blockIdx.x * L + P.t_guard, gid, lane);
  for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = nx[e];
    if (f + gridDim.x < n_frames) frame_load<T, NW>(nx, rx + (f + gridDim.x) * L + P.t_guard, gid, lane);   // next frame in flight
    if constexpr (NW > 1) {
      dif_stage<T, NW>(v, dt);
      __syncthreads();
      dif_scatter<T, NW>(v, gid, ex);
      __syncthreads();
      dif_gather<T>(v, wave, lane, ex);
    }
    wave_fft512<T, PRUNE2>(v, lane, twb, twl, lwv + wave * WAVE_LDS_ELEMS);
#pragma unroll
    for (int t = 0; t < NOUT; ++t) {
      if (kk[t] < P.n_carrier) {
        P.stash[f * P.n_carrier + kk[t]] = v[t];
        if (pp[t] >= 0) P.ypil[f * P.np + pp[t]] = cdiv(v[t], P.pilots[pp[t]]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// kernel 2: batched OMP.  A workgroup of 4 wavefronts owns FB = 4 * FPW frames (FPW = 8, 4, 2 or 1
// frames per wavefront, chosen on the host from the LDS the per-frame state needs).
//   stage 1: c0 = S^H Y for all FB frames as one real GEMM on the matrix cores
//   stage 2: each wavefront runs the OMP iterations of FPW frames SIDE BY SIDE: a frame is a group
//            of LPF = 64/FPW lanes (arg-max by DPP butterflies inside the group).  Up to 8 taps the refit
//            is register-resident and redundant per lane (omp_frame_reg); beyond, the state lives in LDS
//            and the refit is spread over the group's lanes with R = L^-1 updated in place.
//            Solve arithmetic is in the data precision T (double in parity mode).
// ---------------------------------------------------------------------------------------------
struct OmpLayout {          // byte offsets into dynamic LDS
  unsigned off_y, off_c0, off_gram, off_state, off_fft, state_bytes, total;
  int fpw;                  // frames per wavefront
  int c0_stride;            // row stride of c0 (k_atoms, or np + 1 when c0 overwrites Y in place)
  int reg_c0;               // wave form with c0 by transform: c0 lands on Y's rows, moves to registers, and the R state takes the bytes
};

template <typename T>
static OmpLayout omp_layout(int np, int k_atoms, int taps, int fft_elems = 0) {
  OmpLayout o;
  const bool wave = taps > OMP_RT;            // more than OMP_RT taps: one frame per wavefront (omp_wave_core.hpp)
  // per frame: up to OMP_RT taps nothing (registers); beyond, R = L^-1 [taps][odd stride] complex T
  o.state_bytes = wave ? (unsigned)((sizeof(cx<T>) * (size_t)taps * omp_wave_rs(taps) + 15) & ~15u) : 16u;
  const size_t gram_elems = wave ? (size_t)((k_atoms + 511) & ~511) + k_atoms : (size_t)k_atoms;   // two-sided table for the wave form
  const size_t per_frame = sizeof(cx<T>) * (np + 1) + sizeof(cx<T>) * k_atoms + o.state_bytes;
  int fpw = wave ? 1 : 4;     // 16 frames per workgroup: 2 workgroups per CU keep 8 wavefronts in flight
  if (const char* e = getenv("OFDM_OMP_FPW")) { const int v = atoi(e); if (!wave && (v == 1 || v == 2 || v == 4 || v == 8)) fpw = v; }
  while (fpw > 1 && 4 * fpw * per_frame + sizeof(cx<T>) * gram_elems > 96 * 1024) fpw >>= 1;
  o.fpw = fpw;
  const int fb = 4 * fpw;
  unsigned b = 0;
  o.c0_stride = k_atoms;
  o.reg_c0 = (wave && fft_elems > 0 && k_atoms <= 512 && k_atoms <= np + 1 && !getenv("OFDM_OMP_C0_LDS")) ? 1 : 0;
  const unsigned fft_bytes = (unsigned)((sizeof(cx<T>) * (size_t)fft_elems + 15) & ~15u);
  if (o.reg_c0) {
    // Gram table | { Y rows (c0 written over them) + transform scratch }  ==  { R state of the four frames }: 41.8 KB instead of
    // 74.8 KB at 32 taps, K = Np = 512 -- three resident workgroups per CU instead of two for a pursuit that is bound by the
    // instruction issue of its (few) wavefronts
    o.off_gram = b;  b += (unsigned)((sizeof(cx<T>) * gram_elems + 15) & ~15u);
    o.off_y = b; o.off_c0 = b; o.c0_stride = np + 1;
    const unsigned ybytes = (unsigned)((sizeof(cx<T>) * fb * (np + 1) + 15) & ~15u);
    o.off_fft = b + ybytes;
    o.off_state = b;
    b += std::max<unsigned>(fb * o.state_bytes, ybytes + fft_bytes);
    o.total = b;
    return o;
  }
  o.off_y = b;     b += (unsigned)((sizeof(cx<T>) * fb * (np + 1) + 15) & ~15u);   // rows padded by one element
  o.off_c0 = b;    b += (unsigned)((sizeof(cx<T>) * fb * k_atoms + 15) & ~15u);
  o.off_gram = b;  b += (unsigned)((sizeof(cx<T>) * gram_elems + 15) & ~15u);
  // the transform scratch of the c0 stage is dead before the per-frame OMP state is first written: same bytes
  o.off_state = b; o.off_fft = b;
  b += std::max<unsigned>(fb * o.state_bytes, fft_bytes);
  o.total = b;
  return o;
}

// CM = 2048: comb pilots with Nfft/comb = 2048 -- c0 = S^H Y is the first K outputs of a 2048-point inverse transform of
// Y (one workgroup transform per frame, tw_m = its twiddle table) instead of the K x Np dictionary correlation.
template <typename T, bool MFMA, int CM = 0>
__global__ __launch_bounds__(256) void omp_batch_kernel(FastParams<T> P, OmpLayout lay, int64_t n_frames,
                                                        const cx<T>* __restrict__ tw_m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* Yl = (cx<T>*)(smem + lay.off_y);       // [FB][np + 1]
  cx<T>* c0 = (cx<T>*)(smem + lay.off_c0);      // [FB][k_atoms]
  cx<T>* gl = (cx<T>*)(smem + lay.off_gram);    // [k_atoms] Gram table in the working precision
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int FPW = lay.fpw, FB = 4 * FPW;
  const int64_t f0 = (int64_t)blockIdx.x * FB;
  const int np = P.np, K = P.k_atoms, taps = P.taps;
  const int YS = np + 1;                        // LDS row stride of Y (bank-conflict padding)
  // ---- stage Y (zeros for frames past the end) and the Gram table
  for (int i = tid; i < FB * np; i += 256) {
    const int f = i / np, p = i - f * np;
    Yl[f * YS + p] = (f0 + f < n_frames) ? P.ypil[(f0 + f) * np + p] : mk<T>(0, 0);
  }
  if (taps > OMP_RT) {                          // two-sided: gl[KP + d] = a_k^H a_{k+d}, d in (-K, K)
    const int KP = (K + 511) & ~511;
    for (int i = tid; i < KP + K; i += 256) {
      const int d = i - KP;
      const c64 gq = P.gram[d >= 0 ? d : (d > -K ? -d : 0)];
      gl[i] = mk<T>((T)gq.x, (T)(d >= 0 ? gq.y : -gq.y));
    }
  } else {
    for (int i = tid; i < K; i += 256) gl[i] = mk<T>((T)P.gram[i].x, (T)P.gram[i].y);
  }
  __syncthreads();
  const int CS = lay.c0_stride;
  // ||Y||^2 of this wavefront's frame(s) -- before c0 may overwrite Y (reg_c0 layout)
  const int LPF = 64 / FPW;                      // lanes per frame
  const int grp = lane / LPF, sl = lane - grp * LPF;
  const int fi = wave * FPW + grp;               // frame slot inside the workgroup
  double ynorm = 0;
  {
    const cx<T>* yf = Yl + fi * YS;
    for (int p = sl; p < np; p += LPF) ynorm += (double)yf[p].x * yf[p].x + (double)yf[p].y * yf[p].y;
    for (int off = LPF >> 1; off > 0; off >>= 1) ynorm += __shfl_xor(ynorm, off, 64);
  }
  // ---- c0 = S^H Y
  if constexpr (CM > 0) {
    static_assert(CM == 2048, "one 256-thread workgroup = one 2048-point transform");
    cx<T>* fl = (cx<T>*)(smem + lay.off_fft);
    if (lay.reg_c0) __syncthreads();             // every wavefront has its ||Y||^2: the rows may be overwritten
    for (int fi2 = 0; fi2 < FB; ++fi2) {
      cx<T> v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = tid + e * (CM / 8);
        v[e] = i < np ? conj(Yl[fi2 * YS + i]) : mk<T>(0, 0);
      }
      wg_fft<T, CM, false>(v, tid, tw_m, fl);                 // c0 = conj(FFT(conj(Y)))  (its first barrier follows the loads above)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = tid + e * (CM / 8);
        if (k < K) c0[fi2 * CS + k] = conj(v[e]);
      }
    }
  } else if constexpr (MFMA) {
    corr_mfma_f32(P.sct, K, K / 16, Yl, YS, np, FB, (float*)c0, CS, wave, lane);
  } else {
    for (int i = tid; i < FB * K; i += 256) {
      const int f = i / K, k = i - f * K;
      cx<T> acc = mk<T>(0, 0);
      for (int p = 0; p < np; ++p) acc = acc + P.sct[(size_t)p * K + k] * Yl[f * YS + p];
      c0[f * CS + k] = acc;
    }
  }
  __syncthreads();
  // ---- stage 2: FPW frames per wavefront side by side (no workgroup barrier below, but for the hand-over of the reg_c0 layout)
  const int64_t f = f0 + fi;
  const bool live = f < n_frames;
  const cx<T>* cf = c0 + fi * CS;
  if (taps <= OMP_RT) {
    omp_frame_reg<T>(P, cf, gl, K, taps, LPF, sl, live, ynorm, f);
  } else {
    // one frame per wavefront (lay.fpw == 1): picks / coefficients in lanes, R = L^-1 in LDS (omp_wave_core.hpp)
    cx<T>* Rm = (cx<T>*)(smem + lay.off_state + (size_t)fi * lay.state_bytes);
    cx<T> c0r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c0r[u] = mk<T>(0, 0);
    if (lay.reg_c0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c0r[u] = lane + 64 * u < K ? cf[lane + 64 * u] : mk<T>(0, 0);
      __syncthreads();                           // every wavefront holds its c0: the R states may take the bytes
      omp_frame_wave<T, true>(P, cf, c0r, gl, Rm, K, taps, live, ynorm, f);
    } else {
      omp_frame_wave<T, false>(P, cf, c0r, gl, Rm, K, taps, live, ynorm, f);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// kernel 3: all symbols of a frame
// ---------------------------------------------------------------------------------------------
// HEXT: the channel estimate of every frame comes from P.h_in (MMSE mode) instead of the OMP taps.
// (second launch bound = wavefronts per SIMD the register allocation must allow: 5 <-> 96 VGPRs, i.e. five resident
// 256-thread workgroups per CU for the pruned Nfft <= 2048 instantiations, which is where the benchmark runs)
template <typename T, int NW, bool PRUNE2, int BA, bool HEXT = false>
__global__ __launch_bounds__(64 * NW, sizeof(T) == 4 ? (PRUNE2 ? (NW <= 4 ? 5 : 4) : 3) : 2) void rx_symbols_kernel(FastParams<T> P, const cx<T>* __restrict__ rx,
                                                             int64_t n_frames, uint32_t* __restrict__ bits_out,
                                                             const uint32_t* __restrict__ ref_bits,
                                                             uint32_t* __restrict__ errors_out,
                                                             cx<T>* __restrict__ h_out, int32_t* __restrict__ index_out,
                                                             DemapTable<T> tab) {
  constexpr int N = 512 * NW;
  constexpr int NOUT = PRUNE2 ? 2 : 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* lwv = (cx<T>*)smem;                                           // [NW][WAVE_LDS_ELEMS]
  cx<T>* const ex = lwv;                                               // exchange regions alias the private ones
  cx<T>* twl = lwv + NW * WAVE_LDS_ELEMS;                              // [WAVE_TW_ELEMS]
  uint8_t* codes = (uint8_t*)(twl + WAVE_TW_ELEMS);                    // [n_symb * nd]
  __shared__ unsigned int sh_err;
  __shared__ int sh_tidx[FAST_MAXT];
  __shared__ c64 sh_tx[FAST_MAXT];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gid = threadIdx.x;
  DifTw<T, NW> dt;
  wave_tw_fill<T, NW>(twl, P.tw);
  dif_tw_init<T, NW>(dt, gid, P.tw);
  cx<T> twb[7];
#pragma unroll
  for (int t = 1; t < 8; ++t) twb[t - 1] = P.tw[(t * (lane & 7) * 8) * NW];
  int kk[NOUT], dd[NOUT];
#pragma unroll
  for (int t = 0; t < NOUT; ++t) {
    kk[t] = NW * (lane + 64 * t) + wave;
    dd[t] = kk[t] < P.n_carrier ? (int)P.drole[kk[t]] : -1;
  }
  // Sub-transform `wave` yields the carriers = wave (mod NW).  With comb pilots whose period is a multiple of NW's divisor
  // (the benchmark layouts: pilots 1:4:end, NW = 4 or 8) some wavefronts own pilot carriers only: on data symbols they keep
  // the barriers company and skip their 512-point transform.
  bool mine = false;
#pragma unroll
  for (int t = 0; t < NOUT; ++t) mine = mine || dd[t] >= 0;
  const bool wave_has_data = __any(mine) != 0;
  const int Lsym = N + P.t_guard;
  const int taps = P.taps, bps = P.bps, nd = P.nd;
  const int64_t frame_bits = (int64_t)nd * P.n_symb * bps;
  const int n_codes = nd * P.n_symb;
  if (gid < 32) codes[((n_codes + 31) & ~31) - 32 + gid] = 0;      // zero padding of the last 32-symbol group
  __syncthreads();
  cx<T> v[8], nx[8];
  for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
    const cx<T>* frx = rx + f * (int64_t)Lsym * P.n_symb;
    if (P.n_symb > 1) frame_load<T, NW>(nx, frx + Lsym + P.t_guard, gid, lane);
    if (gid == 0) sh_err = 0;
    // ---- H = fft(h)(1..N_carrier) from the taps; G = 1 ./ H           (OMP_estimate.m:36, equalize_signal.m:6)
    //      MMSE mode: H was written by mmse_apply_kernel (ofdm_chain_mmse.hip)
    if (!HEXT && gid < taps) {
      const int idx = P.tap_idx[f * taps + gid];
      sh_tidx[gid] = idx;
      sh_tx[gid] = P.tap_x[f * taps + gid];
      if (index_out) index_out[f * taps + gid] = idx + 1;
    }
    __syncthreads();
    cx<T> geq[NOUT];
#pragma unroll
    for (int t = 0; t < NOUT; ++t) {
      geq[t] = mk<T>(0, 0);
      if constexpr (HEXT) {
        if (kk[t] < P.n_carrier) {
          const cx<T> H = P.h_in[f * P.n_carrier + kk[t]];
          if (h_out) h_out[f * P.n_carrier + kk[t]] = H;
          geq[t] = cdiv(mk<T>(1, 0), H);
        }
      } else if (kk[t] < P.n_carrier) {
        double hr = 0, hi = 0;
        for (int q = 0; q < taps; ++q) {
          const int idx = sh_tidx[q];
          const c64 x = sh_tx[q];                       // zero for unused / overwritten slots
          // W_N^(idx k) computed, not looked up: the table gather (64 addresses of stride 4 idx per wave-instruction,
          // one dependent round trip per tap) made this prologue 12 % of a frame's latency (in-kernel stamps); the
          // exponent is reduced exactly in integers, so the sine / cosine see an exact argument
          const int e = (int)(((int64_t)(idx < 0 ? 0 : idx) * kk[t]) & (N - 1));
          T ws, wc;
          if constexpr (sizeof(T) == 4) {                     // v_sin_f32 / v_cos_f32 take the angle in turns; |error| ~ 1e-6
            const float turns = (float)e * (1.0f / (float)N);
            ws = __builtin_amdgcn_sinf(turns);
            wc = __builtin_amdgcn_cosf(turns);
          } else {
            sincospi((double)(2 * e) * (1.0 / (double)N), &ws, &wc);
          }
          hr += x.x * (double)wc + x.y * (double)ws;       // w = (wc, -ws)
          hi += x.y * (double)wc - x.x * (double)ws;
        }
        const cx<T> H = mk<T>((T)hr, (T)hi);
        if (h_out) h_out[f * P.n_carrier + kk[t]] = H;
        geq[t] = cdiv(mk<T>(1, 0), H);
      }
    }
    // ---- symbol 1 from the stash
#pragma unroll
    for (int t = 0; t < NOUT; ++t)
      if (dd[t] >= 0) codes[dd[t]] = (uint8_t)slice_symbol<T, BA>(tab, P.stash[f * P.n_carrier + kk[t]] * geq[t]);
    // ---- symbols 2..S: the next symbol's samples are in flight while this one is transformed.  Measured and
    //      rejected: alternating two register sets instead of copying (-1 workgroup of occupancy), two symbols in
    //      flight (119 VGPRs, 11 % slower) -- the kernel is not HBM-latency bound.
    for (int s = 1; s < P.n_symb; ++s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = nx[e];
      if (s + 1 < P.n_symb) frame_load<T, NW>(nx, frx + (int64_t)(s + 1) * Lsym + P.t_guard, gid, lane);
      if constexpr (NW > 1) {
        dif_stage<T, NW>(v, dt);
        __syncthreads();                 // every wavefront has finished its previous transform (region reuse)
        dif_scatter<T, NW>(v, gid, ex);
        __syncthreads();
        if (wave_has_data) dif_gather<T, true>(v, wave, lane, ex);
      }
      if (wave_has_data) {                                             // wavefront-uniform
        wave_fft512<T, PRUNE2, true, true>(v, lane, twb, twl, lwv + wave * WAVE_LDS_ELEMS);
#pragma unroll
        for (int t = 0; t < NOUT; ++t)
          if (dd[t] >= 0) codes[s * nd + dd[t]] = (uint8_t)slice_symbol<T, BA>(tab, v[t] * geq[t]);
      }
    }
    __syncthreads();
    // ---- pack (bit i of the frame -> byte i/8, bit 7-i%8) + BER numerator.  A group of 32 decided
    //      symbols is exactly `bps` 32-bit words: two 16-byte LDS reads, then registers only.
    const unsigned int err = pack_frame<2 * BA>(codes, n_codes, bps, P.frame_words,
                                                bits_out ? bits_out + f * P.frame_words : nullptr,
                                                ref_bits ? ref_bits + f * P.frame_words : nullptr, gid, 64 * NW);
    (void)frame_bits;
    if (ref_bits && errors_out) {
      if (err) atomicAdd(&sh_err, err);
      __syncthreads();
      if (gid == 0) errors_out[f] = sh_err;
    }
    __syncthreads();                     // codes / sh_err are reused by the next frame
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

bool chain_fast_supported(int nfft, int n_carrier, int taps, int bps, int64_t nd_nsymb) {
  if (getenv("OFDM_CHAIN_GENERIC")) return false;
  if (!(nfft == 512 || nfft == 1024 || nfft == 2048 || nfft == 4096)) return false;
  if (taps > FAST_MAXT || bps > 8) return false;
  if (nd_nsymb > 48 * 1024) return false;
  (void)n_carrier;
  return true;
}

// Fills the kernel parameter block of a plan and grows the plan's workspace to n_frames (shared by the fast
// path here and the split path of ofdm_chain_split.hip).
template <typename T>
int fast_params_prepare(const FastPlanView& pv, const void* tw, int64_t n_frames, FastParams<T>& P) {
  P.n_symb = pv.n_symb; P.t_guard = pv.t_guard; P.n_carrier = pv.n_carrier; P.np = pv.np; P.nd = pv.nd;
  P.k_atoms = pv.k_atoms; P.taps = pv.taps; P.frame_words = pv.frame_words; P.bps = pv.bps;
  P.prole = (const int16_t*)pv.d_prole; P.drole = (const int16_t*)pv.d_drole;
  P.pilots = (const cx<T>*)pv.d_pilots; P.sct = (const cx<T>*)pv.d_sct; P.gram = (const c64*)pv.d_gram;
  P.tw = (const cx<T>*)tw;
  P.comb_m = pv.comb_m;
  // workspace (grown on demand, kept by the plan)
  if (*pv.ws_frames < n_frames) {
    void** ptrs[] = {pv.ws_stash, pv.ws_ypil, pv.ws_tapidx, pv.ws_tapx, pv.ws_h};
    OFDM_HIP(hipStreamSynchronize(ctx().stream));
    for (void** p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
    OFDM_HIP(hipMalloc(pv.ws_stash, sizeof(cx<T>) * (size_t)pv.n_carrier * n_frames));
    OFDM_HIP(hipMalloc(pv.ws_ypil, sizeof(cx<T>) * (size_t)pv.np * n_frames));
    OFDM_HIP(hipMalloc(pv.ws_tapidx, sizeof(int32_t) * (size_t)pv.taps * n_frames));
    OFDM_HIP(hipMalloc(pv.ws_tapx, sizeof(c64) * (size_t)pv.taps * n_frames));
    if (pv.d_wt) OFDM_HIP(hipMalloc(pv.ws_h, sizeof(cx<T>) * (size_t)pv.n_carrier * n_frames));
    *pv.ws_frames = n_frames;
  }
  P.stash = (cx<T>*)*pv.ws_stash; P.ypil = (cx<T>*)*pv.ws_ypil;
  P.tap_idx = (int32_t*)*pv.ws_tapidx; P.tap_x = (c64*)*pv.ws_tapx;
  const bool mmse = pv.d_wt != nullptr;
  if (mmse && !*pv.ws_h) {                       // the plan switched to MMSE mode after the workspace was made
    OFDM_HIP(hipStreamSynchronize(ctx().stream));
    OFDM_HIP(hipMalloc(pv.ws_h, sizeof(cx<T>) * (size_t)pv.n_carrier * *pv.ws_frames));
  }
  P.h_in = mmse ? (const cx<T>*)*pv.ws_h : nullptr;
  P.descr = pv.descr;
  return OFDM_OK;
}
template int fast_params_prepare<float>(const FastPlanView&, const void*, int64_t, FastParams<float>&);
template int fast_params_prepare<double>(const FastPlanView&, const void*, int64_t, FastParams<double>&);

// omp_batch_kernel over n_frames frames: pilot LS values P.ypil -> taps P.tap_idx / P.tap_x
template <typename T>
int omp_batch_run(const FastParams<T>& P, int64_t n_frames) {
  hipStream_t st = ctx().stream;
  const bool by_fft = P.comb_m == 2048 && P.np <= 2048 && P.k_atoms <= 2048 && !getenv("OFDM_OMP_NO_FFT");
  const OmpLayout lay = omp_layout<T>(P.np, P.k_atoms, P.taps, by_fft ? fft_lds_elems(2048) : 0);
  OFDM_ARG(lay.total <= 150 * 1024, "rx_chain_task5: OMP stage needs %u bytes of LDS", lay.total);
  const unsigned grid = cdiv_u(n_frames, 4 * lay.fpw);
  const bool mfma = std::is_same<T, float>::value && (P.k_atoms % 16 == 0) && (P.np % 4 == 0) && !getenv("OFDM_OMP_NO_MFMA");
  auto launch = [&](auto kern, const void* twm) -> int {
    OFDM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lay.total, st, P, lay, n_frames, (const cx<T>*)twm);
    return OFDM_OK;
  };
  if (by_fft) {
    const void* twm = nullptr;
    OFDM_TRY(get_twiddles(2048, std::is_same<T, double>::value, &twm));
    OFDM_TRY(launch(omp_batch_kernel<T, false, 2048>, twm));
  } else if (mfma) {
    if constexpr (std::is_same<T, float>::value) OFDM_TRY(launch(omp_batch_kernel<T, true>, nullptr));
  } else {
    OFDM_TRY(launch(omp_batch_kernel<T, false>, nullptr));
  }
  return check_launch("omp_batch_kernel");
}
template int omp_batch_run<float>(const FastParams<float>&, int64_t);
template int omp_batch_run<double>(const FastParams<double>&, int64_t);

template <typename T, int NW, bool PRUNE2>
static int launch_fast(const FastPlanView& pv, const void* tw, const void* rx, int64_t n_frames, void* bits,
                       const void* ref, void* errs, void* h_out, void* idx_out) {
  FastParams<T> P;
  OFDM_TRY(fast_params_prepare<T>(pv, tw, n_frames, P));
  const bool mmse = pv.d_wt != nullptr;
  DemapTable<T> tab;
  fill_demap_table<T>(*pv.dict, *pv.cinfo, tab);
  const int ncu = ctx().num_cu;
  hipStream_t st = ctx().stream;
  if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[0], st));
  const bool fused = !mmse && pv.comb_lg_up >= 0 && pv.taps <= OMP_RT && pv.k_atoms <= 512 && !getenv("OFDM_FAST_UNFUSED");
  if (pv.fused_out) *pv.fused_out = fused ? 1 : 0;
  if (fused) {
    // kernels 1+2 in one launch (comb pilots): c0 by a wave-local inverse transform (ofdm_chain_pilot.hip)
    OFDM_TRY(pilot_omp_run<T>(P, 512 * NW, PRUNE2, pv.comb_lg_up, rx, n_frames));
    if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[1], st));
  } else {
    // kernel 1
    {
      const unsigned grid = (unsigned)std::min<int64_t>(n_frames, (int64_t)ncu * std::max(1, 16 / NW));
      hipLaunchKernelGGL((rx_pilot_kernel<T, NW, PRUNE2>), dim3(grid), dim3(64 * NW), 0, st, P, (const cx<T>*)rx, n_frames);
      OFDM_TRY(check_launch("rx_pilot_kernel"));
    }
    if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[1], st));
    // kernel 2
    if (mmse) {
      OFDM_TRY(mmse_stage_run<T>(pv, P, n_frames));
    } else {
      OFDM_TRY(omp_batch_run<T>(P, n_frames));
    }
  }
  if (pv.ev && !fused) OFDM_HIP(hipEventRecord(pv.ev[2], st));
  // kernel 3
  if constexpr (std::is_same<T, float>::value && NW == 4 && PRUNE2) {
    if (chain_wave_supported(pv)) {                  // one wavefront per frame, no workgroup barrier (ofdm_chain_wave.hip)
      OFDM_TRY(chain_wave_symbols_run(pv, P, rx, n_frames, bits, ref, errs, h_out, idx_out));
      if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[3], st));
      return OFDM_OK;
    }
  }
  {
    constexpr int N = 512 * NW;
    const size_t dyn = sizeof(cx<T>) * ((size_t)NW * WAVE_LDS_ELEMS + WAVE_TW_ELEMS) +
                       (((size_t)pv.nd * pv.n_symb + 31) & ~size_t(31));
    OFDM_ARG(dyn <= 150 * 1024, "rx_chain_task5: symbol stage needs %zu bytes of LDS", dyn);
    // persistent grid = CUs x resident workgroups per CU (from the occupancy API: registers + LDS);
    // workgroups are independent, so an optimistic answer only queues a few of them.
    auto launch = [&](auto kern) -> int {
      int per_cu = resident_blocks_per_cu((const void*)kern, 64 * NW, dyn);
      if (const char* e = getenv("OFDM_FAST_WG_PER_CU")) per_cu = std::max(1, atoi(e));
      const unsigned grid = (unsigned)std::min<int64_t>(n_frames, (int64_t)ncu * per_cu);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), dyn, st, P, (const cx<T>*)rx, n_frames, (uint32_t*)bits,
                         (const uint32_t*)ref, (uint32_t*)errs, (cx<T>*)h_out, (int32_t*)idx_out, tab);
      return OFDM_OK;
    };
    const int ba = pv.cinfo->kind == 1 ? pv.cinfo->bits_per_axis : 0;
    if (mmse) {
      switch (ba) {
        case 2: OFDM_TRY(launch(rx_symbols_kernel<T, NW, PRUNE2, 2, true>)); break;
        case 3: OFDM_TRY(launch(rx_symbols_kernel<T, NW, PRUNE2, 3, true>)); break;
        case 4: OFDM_TRY(launch(rx_symbols_kernel<T, NW, PRUNE2, 4, true>)); break;
        default: OFDM_TRY(launch(rx_symbols_kernel<T, NW, PRUNE2, 0, true>)); break;
      }
    } else {
      switch (ba) {
        case 2: OFDM_TRY(launch(rx_symbols_kernel<T, NW, PRUNE2, 2>)); break;
        case 3: OFDM_TRY(launch(rx_symbols_kernel<T, NW, PRUNE2, 3>)); break;
        case 4: OFDM_TRY(launch(rx_symbols_kernel<T, NW, PRUNE2, 4>)); break;
        default: OFDM_TRY(launch(rx_symbols_kernel<T, NW, PRUNE2, 0>)); break;
      }
    }
    OFDM_TRY(check_launch("rx_symbols_kernel"));
  }
  if (pv.ev) OFDM_HIP(hipEventRecord(pv.ev[3], st));
  return OFDM_OK;
}

template <typename T>
static int dispatch_fast(const FastPlanView& pv, const void* tw, const void* rx, int64_t n_frames, void* bits,
                         const void* ref, void* errs, void* h_out, void* idx_out) {
  const int nw = pv.nfft / 512;
  const bool prune = pv.n_carrier <= 128 * nw;
#define FAST_CALL(NWV)                                                                                       \
  return prune ? launch_fast<T, NWV, true>(pv, tw, rx, n_frames, bits, ref, errs, h_out, idx_out)            \
               : launch_fast<T, NWV, false>(pv, tw, rx, n_frames, bits, ref, errs, h_out, idx_out)
  switch (nw) {
    case 1: FAST_CALL(1);
    case 2: FAST_CALL(2);
    case 4: FAST_CALL(4);
    case 8: FAST_CALL(8);
  }
#undef FAST_CALL
  set_error("rx_chain_task5(fast): unsupported Nfft %d", pv.nfft);
  return OFDM_ERR_UNSUPPORTED;
}

int chain_fast_run(const FastPlanView& pv, const void* tw, const void* rx, int64_t n_frames, void* bits,
                   const void* ref, void* errs, void* h_out, void* idx_out) {
  if (pv.f64) return dispatch_fast<double>(pv, tw, rx, n_frames, bits, ref, errs, h_out, idx_out);
  return dispatch_fast<float>(pv, tw, rx, n_frames, bits, ref, errs, h_out, idx_out);
}

}  // namespace ofdm
