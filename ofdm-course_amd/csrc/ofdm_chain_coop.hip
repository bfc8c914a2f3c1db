// Symbol stage of the fused Task-5 RX chain for Nfft = 8192, fp32, N_carrier <= 2048 (BASELINE config 5) in ONE pass over the
// received samples: OFDM_demodulator.m:5-8 -> equalize_signal.m:3-7 -> get_payload.m:3 -> demapping.m:7-18 -> packed bits ->
// BER_func.m:3-6 numerator, with H from the OMP taps (OMP_estimate.m:25-36) or from the MMSE stage -- no X round trip.
//
// Only carriers 1..N_carrier <= Nfft/4 are ever used, so the 8192-point transform is a radix-4 decimation-in-frequency step
// onto FOUR 2048-point transforms each pruned to its first quarter -- exactly the transform of the metric kernel:
//     y_s[m] = (sum_r x[m + 2048 r] W_4^(r s)) W_8192^(m s),      X[4 k' + s] = FFT_2048(y_s)[k'],   k' < 512
// A workgroup = four wavefronts = one symbol at a time:
//   A. wavefront w loads x[m + 2048 r] for ITS quarter of m (32 coalesced 8-byte nontemporal loads per lane), does the radix-4
//      butterflies in registers and hands y_s[m] to wavefront s through LDS -- the ONLY exchange between wavefronts, two
//      workgroup barriers per symbol (the split form's demodulator took seven)
//   B. wavefront s runs the 2048-point transform of y_s alone (ofdm_chain_wave.hip's decomposition: 32-point DFT in registers,
//      four rounds of eight kj across the lanes, last radix-8 pruned to 2 outputs), equalises, slices and writes code bytes
//   C. the raw samples of the NEXT symbol are requested right after step A into registers of their own, so they travel during B
//   D. wavefront 0 packs the previous symbol's codes and counts errors while the others are in A / B of the next symbol
// Measured (3072 frames of 14 symbols, C5): 0.75 ms for this stage at two workgroups per CU against 1.31 ms at one -- it is bound
// by latency (VALU issue 45 %, LDS 24 %, 61 % of the wavefront-cycles waiting), and the 48 KB hand-over buffer is what keeps a
// third workgroup off the CU.  Handing the residues over one at a time through 16 KB (44.7 KB per workgroup, 168 VGPRs) was
// built and measured: 220 spilled registers (the three y_s of a quarter live beside the sample array) and six barriers per
// symbol -- 1.63 ms.  A barrier right behind the hand-over reads instead of behind the transforms: 0.79 ms.
// With comb-4 pilots (1 : 4 : end) the sub-transform s = 0 holds nothing but pilot carriers: its wavefront skips step B on data
// symbols (and is the one that packs).  H = fft(h_est) on the carriers is ONE MORE transform of the same kind per frame (the
// taps scattered into a 512-sample vector), computed in place of a symbol; the first symbol's rows come from the stash of the
// pilot pass (demod_keep8192_kernel on first symbols only + pilot LS + omp_batch_kernel, ofdm_chain_split.hip).
#include <algorithm>

#include "chain_fast_core.hpp"

namespace ofdm {

constexpr int CP_N = 8192, CP_SUB = 2048, CP_R = 4;
constexpr int CP_TR_ELEMS = 576;
// dynamic LDS: W_2048^(l ka) [7][64] | W_64^(t (l & 7)) [7][64] | 4 wave regions | exchange y_s [3][2048] (s = 1..3; its first
//              512 entries hold h_est during a frame's H transform) | codes [2][nd_pad]      -- 76.1 KB + codes: two workgroups per CU
constexpr unsigned CP_OFF_TWA = 0, CP_OFF_TWB = 8 * 64 * 7;
constexpr unsigned CP_OFF_WAVE = CP_OFF_TWB + 8 * 64 * 7;
constexpr unsigned CP_OFF_EX = CP_OFF_WAVE + CP_R * 8 * CP_TR_ELEMS;
constexpr unsigned CP_OFF_CODES = CP_OFF_EX + 3 * CP_SUB * 8;

__device__ __forceinline__ cx<float> cp_w32(int m) {
  constexpr float C[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                          0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                          0.19509032201612826785f, 0.0f};
  m &= 31;
  const int q = m >> 3, r = m & 7;
  const float c = C[r], s = C[8 - r];
  switch (q) {
    case 0: return mk<float>(c, -s);
    case 1: return mk<float>(-s, -c);
    case 2: return mk<float>(-c, s);
    default: return mk<float>(s, c);
  }
}

template <int BA, bool HEXT>
__global__ __launch_bounds__(256, 2) void rx_symbols_coop4_kernel(FastParams<float> P, unsigned codes_bytes,
                                                                  const cx<float>* __restrict__ rx, int64_t n_frames,
                                                                  uint32_t* __restrict__ bits_out, const uint32_t* __restrict__ ref_bits,
                                                                  uint32_t* __restrict__ errors_out, cx<float>* __restrict__ h_out,
                                                                  int32_t* __restrict__ index_out, DemapTable<float> tab) {
  using T = float;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int tap_i[FAST_MAXT];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = the residue s of this wavefront's carriers
  const unsigned lane8 = 8u * lane;
  const int nd = P.nd, taps = P.taps, n_symb = P.n_symb, nc = P.n_carrier;
  const int Lsym = CP_N + P.t_guard;
  const int64_t Lframe = (int64_t)Lsym * n_symb;
  auto twa_at = [&](int row) { return *(const cx<T>*)(smem + CP_OFF_TWA + 512 * row + lane8); };     // W_2048^(lane (row+1))
  auto twb_at = [&](int row) { return *(const cx<T>*)(smem + CP_OFF_TWB + 512 * row + lane8); };     // W_64^((row+1) (lane&7))
  // carrier of output t = 2 kb + q of this lane (lane = 8 c + ka): k' = c + 8 kb + 32 (ka + 8 q), k = 4 k' + wave
  auto kk_of = [&](int t) { return 4 * ((lane >> 3) + 8 * (t >> 1) + 32 * ((lane & 7) + 8 * (t & 1))) + wave; };
  {
    cx<T>* const ta = (cx<T>*)(smem + CP_OFF_TWA);
    cx<T>* const tb = (cx<T>*)(smem + CP_OFF_TWB);
    for (int i = threadIdx.x; i < 7 * 64; i += 256) {
      const int t = i / 64 + 1, l = i & 63;
      ta[i] = P.tw[(4 * t * l) & (CP_N - 1)];                          // W_2048^x = W_8192^(4x)
      tb[i] = P.tw[(128 * t * (l & 7)) & (CP_N - 1)];                  // W_64^x = W_8192^(128 x)
    }
  }
  // data position of the lane's output t (or -1), two to a register
  int ddp[4];
#pragma unroll
  for (int t2 = 0; t2 < 4; ++t2) {
    const int k0 = kk_of(2 * t2), k1 = kk_of(2 * t2 + 1);
    const unsigned d0 = k0 < nc ? (unsigned)(unsigned short)P.drole[k0] : 0xffffu;
    const unsigned d1 = k1 < nc ? (unsigned)(unsigned short)P.drole[k1] : 0xffffu;
    ddp[t2] = (int)(d0 | (d1 << 16));
  }
  auto dd_of = [&](int t) { return (int)(short)((unsigned)ddp[t >> 1] >> (16 * (t & 1))); };
  // per-lane constants: the radix-4 twiddles W_8192^m of this wavefront's samples m = 512 w + l + 64 j', and the part
  // W_2048^(8 l kb) of the transform's twiddle W_2048^(l (ka + 8 kb)) that the 7-row table leaves out
  cx<T> wq[8], wkb[3];
#pragma unroll
  for (int j = 0; j < 8; ++j) wq[j] = P.tw[512 * wave + lane + 64 * j];
#pragma unroll
  for (int kb = 1; kb < 4; ++kb) wkb[kb - 1] = P.tw[(32 * lane * kb) & (CP_N - 1)];
  __syncthreads();
  bool any_data = false;
#pragma unroll
  for (int t = 0; t < 8; ++t) any_data = any_data || dd_of(t) >= 0;
  const bool wave_has_data = __any(any_data) != 0;                     // wavefront-uniform
  const unsigned wbase = CP_OFF_WAVE + (unsigned)wave * 8 * CP_TR_ELEMS;
  cx<T>* const t1w = (cx<T>*)(smem + wbase + lane8);
  cx<T>* const t1r = (cx<T>*)(smem + wbase) + 72 * (lane >> 3) + (lane & 7);
  cx<T>* const t2w = (cx<T>*)(smem + wbase) + 65 * (lane & 7) + 8 * (lane >> 3);
  cx<T>* const t2r = (cx<T>*)(smem + wbase + lane8);
  cx<T>* const ex = (cx<T>*)(smem + CP_OFF_EX);                        // y_s at ex[(s - 1) * 2048 + m]; hbuf = ex[0 .. 511]
  uint8_t* const codes0 = smem + CP_OFF_CODES;
  const int frame_words = P.frame_words, bps = P.bps;
  const int words_per_symbol = (nd >> 5) * bps;                        // nd is a multiple of 32 (checked on the host)

  // the pruned 2048-point transform of the 32 values in v (lane l holds y[l + 64 j]); emit(t, value) for the lane's 8 outputs
  // refill(kb): called when the registers v[kb + 4 m] of round kb are in LDS and free
  auto transform = [&](cx<T> (&v)[32], auto emit, auto refill) __attribute__((always_inline)) {
#pragma unroll
    for (int j0 = 0; j0 < 4; ++j0) {
      dft8<T, false>(v[j0], v[j0 + 4], v[j0 + 8], v[j0 + 12], v[j0 + 16], v[j0 + 20], v[j0 + 24], v[j0 + 28]);
      if (j0 > 0) {
#pragma unroll
        for (int ka = 1; ka < 8; ++ka) v[j0 + 4 * ka] = v[j0 + 4 * ka] * cp_w32(j0 * ka);
      }
    }
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) dft4<T, false>(v[4 * ka], v[4 * ka + 1], v[4 * ka + 2], v[4 * ka + 3]);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int ka = 0; ka < 8; ++ka) {
        cx<T> z = v[kb + 4 * ka];
        if (ka > 0) z = z * twa_at(ka - 1);
        if (kb > 0) z = z * wkb[kb - 1];
        t1w[72 * ka] = z;
      }
      refill(kb);
      wave_sync();
      cx<T> u[8];
      lds_read8<8, true>(u, t1r);
      wave_sync();
      dft8<T, false>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]);
#pragma unroll
      for (int t = 1; t < 8; ++t) u[t] = u[t] * twb_at(t - 1);
#pragma unroll
      for (int t = 0; t < 8; ++t) t2w[t] = u[t];
      wave_sync();
      lds_read8<65, true>(u, t2r);
      wave_sync();
      dft8_first2<T>(u);
      emit(2 * kb, u[0]);
      emit(2 * kb + 1, u[1]);
    }
  };
  // where this wavefront's raw samples of (frame, symbol) start: x[m + 2048 r] at src[64 j' + 2048 r], m = 512 w + l + 64 j'.
  // ONE register array serves the raw samples and the transform: sample (j', r) lives in v[r + 4 j'], i.e. the quarter r is the
  // register class that round kb = r of the transform frees -- the next symbol's samples are requested round by round into the
  // registers the current transform has just left (the metric kernel's prefetch: no second register set).
  auto raw_src = [&](int64_t f, int sy) { return rx + f * Lframe + (int64_t)sy * Lsym + P.t_guard + 512 * wave + lane; };
  cx<T> v[32];
  auto load_quarter = [&](const cx<T>* src, int r) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[r + 4 * j] = nt_load(src + 64 * j + CP_SUB * r);
  };
  auto load_raw = [&](const cx<T>* src) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) load_quarter(src, r);
  };
  // a frame's first data symbol is requested by whatever runs last before it: the H transform's rounds (OMP mode, wavefronts
  // that run it), the previous frame's last transform (MMSE mode), or directly
  constexpr bool H_REFILLS = !HEXT;
  if ((int64_t)blockIdx.x < n_frames && n_symb > 1 && (HEXT || !(wave_has_data || h_out))) load_raw(raw_src(blockIdx.x, 1));

  for (int64_t f = blockIdx.x; f < n_frames; f += gridDim.x) {
    __syncthreads();                                                   // the previous frame's last pack / exchange reads are done
    cx<T> geq[8];
    // ---- G = 1 ./ H on this lane's carriers
    if constexpr (HEXT) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        geq[t] = mk<T>(0, 0);
        if (kk_of(t) < nc) {
          const cx<T> H = P.h_in[f * nc + kk_of(t)];
          if (h_out) h_out[f * nc + kk_of(t)] = H;
          geq[t] = cdiv(mk<T>(1, 0), H);
        }
      }
    } else {
      // h_est: the picked taps at their delays (a later pick of the same atom overwrites, OMP_estimate.m:31-33), then
      // H = fft(h_est)(4 k' + s) = FFT_2048(h[m] W_8192^(m s))[k'] -- h has no samples beyond 511 (K <= 512): no butterfly
      for (int i = threadIdx.x; i < 512; i += 256) ex[i] = mk<T>(0, 0);
      int my_idx = -1;
      if ((int)threadIdx.x < taps) {
        my_idx = P.tap_idx[f * taps + threadIdx.x];
        tap_i[threadIdx.x] = my_idx;
        if (index_out) index_out[f * taps + threadIdx.x] = my_idx + 1;
      }
      __syncthreads();
      if (my_idx >= 0) {
        bool later = false;
        for (int q2 = threadIdx.x + 1; q2 < taps; ++q2) later = later || tap_i[q2] == my_idx;
        if (!later) { const c64 xv = P.tap_x[f * taps + threadIdx.x]; ex[my_idx] = mk<T>((T)xv.x, (T)xv.y); }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 8; ++t) geq[t] = mk<T>(0, 0);
      if (wave_has_data || h_out) {
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = mk<T>(0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int m = lane + 64 * j;
          v[j] = ex[m] * P.tw[(m * wave) & (CP_N - 1)];
        }
        const cx<T>* nsrc = raw_src(f, 1);
        transform(v, [&](int t, cx<T> H) {
          if (kk_of(t) < nc) {
            if (h_out) h_out[f * nc + kk_of(t)] = H;
            geq[t] = cdiv(mk<T>(1, 0), H);
          }
        }, [&](int kb) { if (n_symb > 1) load_quarter(nsrc, kb); });
      } else if (f != (int64_t)blockIdx.x && n_symb > 1) {
        load_raw(raw_src(f, 1));                                       // (the first frame's were requested before the loop)
      }
      __syncthreads();                                                 // hbuf (= exchange slot 0) is free again
    }
    // ---- symbol 1 from the stash of the pilot pass -> codes slot 0
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int d = dd_of(t);
      if (d >= 0) codes0[d] = (uint8_t)slice_symbol<T, BA>(tab, P.stash[f * nc + kk_of(t)] * geq[t]);
    }
    unsigned err = 0;
    auto pack = [&](int sy) __attribute__((always_inline)) {          // wavefront 0 only
      const int woff = sy * words_per_symbol;
      err += pack_frame_t<2 * BA, false>(codes0 + (sy & 1) * codes_bytes, nd, bps, frame_words - woff,
                                         bits_out ? bits_out + f * frame_words + woff : nullptr,
                                         ref_bits ? ref_bits + f * frame_words + woff : nullptr, lane, 64, 0u);
    };
    for (int sy = 1; sy < n_symb; ++sy) {
      // ---- A. radix-4 butterflies on this wavefront's quarter of m, y_s[m] -> exchange (s = 1..3; s = 0 is pilot carriers only
      //         on a data symbol when its wavefront holds no data -- then nobody reads it)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const cx<T> a = v[4 * j] + v[4 * j + 2], b = v[4 * j] - v[4 * j + 2];
        const cx<T> c = v[4 * j + 1] + v[4 * j + 3], d = mul_mi<T, false>(v[4 * j + 1] - v[4 * j + 3]);
        const cx<T> w1 = wq[j], w2 = w1 * w1, w3 = w2 * w1;
        const int m = 512 * wave + lane + 64 * j;
        ex[m] = (b + d) * w1;
        ex[CP_SUB + m] = (a - c) * w2;
        ex[2 * CP_SUB + m] = (b - d) * w3;
        (void)c;                                                       // y_0 = a + c: pilot carriers only, nobody reads it
      }
      // ---- C. what the freed registers are refilled with: this frame's next symbol; past its last one the next frame's first
      //         data symbol in MMSE mode (in OMP mode the next frame's H transform needs the registers first and requests it)
      const bool more = sy + 1 < n_symb;
      const int64_t fn = more ? f : f + gridDim.x;
      const bool fetch = more || (!H_REFILLS && fn < n_frames);
      const cx<T>* nsrc = raw_src(fetch ? fn : f, more ? sy + 1 : 1);
      if (!wave_has_data && fetch) load_raw(nsrc);                     // no transform on this wavefront: its registers are free now
      __syncthreads();                                                 // B: y_s complete
      if (sy == 1 && wave == 0) pack(0);
      if (wave_has_data) {
        const cx<T>* src = ex + (wave - 1) * CP_SUB + lane;            // (wavefront 0 never gets here: the host takes this kernel
#pragma unroll                                                         //  only when residue 0 holds no data carrier)
        for (int j = 0; j < 32; ++j) v[j] = src[64 * j];
        uint8_t* const cslot = codes0 + (sy & 1) * codes_bytes;
        transform(v, [&](int t, cx<T> X) {
          const int d = dd_of(t);
          if (d >= 0) cslot[d] = (uint8_t)slice_symbol<T, BA>(tab, X * geq[t]);
        }, [&](int kb) { if (fetch) load_quarter(nsrc, kb); });
      }
      __syncthreads();                                                 // E: codes of symbol sy complete, exchange free
      if (wave == 0) pack(sy);
      // (a second barrier right behind the exchange reads instead of E -- nobody waits for anybody's transform, the pack one
      //  symbol late -- measured 4 % slower: 0.794 against 0.765 ms per 3072 frames)
    }
    if (n_symb == 1) { __syncthreads(); if (wave == 0) pack(0); }
    if (wave == 0 && ref_bits && errors_out) {
      for (int off = 32; off > 0; off >>= 1) err += __shfl_xor(err, off, 64);
      if (lane == 0) errors_out[f] = err;
    }
  }
}

// the kernel's geometry: Nfft 8192, fp32, N_carrier <= 2048, K <= 512, no data carrier on the residue class 0 (mod 4), whole
// 32-code groups per symbol
bool chain_coop_supported(const FastPlanView& pv) {
  if (getenv("OFDM_SPLIT_NO_COOP")) return false;
  if (pv.f64 || pv.nfft != CP_N || pv.n_carrier > CP_N / 4 || pv.k_atoms > 512 || pv.taps > FAST_MAXT) return false;
  if ((pv.data_mod4 & 1) != 0 || (pv.nd & 31) != 0 || pv.n_symb < 1) return false;
  if (pv.descr & DESCR_ON) return false;
  return CP_OFF_CODES + 2u * (unsigned)((pv.nd + 63) & ~31) <= 78u * 1024;
}

int chain_coop_symbols_run(const FastPlanView& pv, const FastParams<float>& P, const void* rx, int64_t n_frames, void* bits,
                           const void* ref, void* errs, void* h_out, void* idx_out) {
  const unsigned codes_bytes = (unsigned)((pv.nd + 63) & ~31);
  const unsigned lds = CP_OFF_CODES + 2 * codes_bytes;
  DemapTable<float> tab;
  fill_demap_table<float>(*pv.dict, *pv.cinfo, tab);
  const bool mmse = pv.d_wt != nullptr;
  auto launch = [&](auto kern) -> int {
    int per_cu = resident_blocks_per_cu((const void*)kern, 256, lds);
    if (const char* e = getenv("OFDM_COOP_WG_PER_CU")) per_cu = std::max(1, atoi(e));
    const unsigned grid = (unsigned)std::min<int64_t>(n_frames, (int64_t)ctx().num_cu * per_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx().stream, P, codes_bytes, (const cx<float>*)rx, n_frames, (uint32_t*)bits,
                       (const uint32_t*)ref, (uint32_t*)errs, (cx<float>*)h_out, (int32_t*)idx_out, tab);
    return OFDM_OK;
  };
  const int ba = pv.cinfo->kind == 1 ? pv.cinfo->bits_per_axis : 0;
#define COOP_CASE(BAV)                                                     \
  if (mmse) OFDM_TRY(launch(rx_symbols_coop4_kernel<BAV, true>));          \
  else OFDM_TRY(launch(rx_symbols_coop4_kernel<BAV, false>))
  switch (ba) {
    case 2: COOP_CASE(2); break;
    case 3: COOP_CASE(3); break;
    case 4: COOP_CASE(4); break;
    default: COOP_CASE(0); break;
  }
#undef COOP_CASE
  return check_launch("rx_symbols_coop4_kernel");
}

}  // namespace ofdm
