// Symbol stage of the fused Task-5 RX chain with ONE WAVEFRONT PER FRAME (Nfft = 2048, fp32): the metric kernel.
//
//   rx_symbols_wave_kernel   per frame: H = fft(h) on carriers 1..N_carrier from the OMP taps (OMP_estimate.m:25-36) or
//                            from the MMSE estimate, then every symbol: OFDM_demodulator.m:5-8 (CP dropped, 2048-point
//                            FFT) -> equalize_signal.m:3-7 -> get_payload.m:3 -> demapping.m:7-18 -> packed bits ->
//                            BER_func.m:3-6 numerator
//
// Why one wavefront per frame: the four-wavefronts-per-symbol form (ofdm_chain_fast.hip) pays two workgroup barriers per
// symbol for its radix-4 exchange, one more barrier + an LDS atomic per frame, and four copies of every per-lane constant.
// Here a frame never leaves its wavefront: no s_barrier, no atomic, no exchange between wavefronts anywhere in the kernel.
//
// Transform: 2048 = 32 x 64.  Lane l holds x[l + 64 j], j < 32 (32 coalesced 8-byte nontemporal loads per symbol; the
// wave-per-frame access shape streams at 6.87 TB/s, tools/ubench/hbm_read.hip).
//   1. 32-point DFT over j in registers (8 x 4: four dft8, constant twiddles W_32, eight dft4)  ->  Z[kj], kj < 32
//   2. Z[kj] *= W_2048^(l kj)                                (LDS table, lane-contiguous)
//   3. for every kj a 64-point DFT ACROSS the lanes, l = l0 + 8 l1, kl = ka + 8 kb:
//        radix-8 over l1 (-> ka), twiddle W_64^(l0 ka), radix-8 over l0 (-> kb) of which only kb < 2 is computed:
//        carrier k = kj + 32 kl < 512 <=> kl < 16  (only carriers 1..N_carrier <= Nfft/4 are ever used)
//      done in four rounds of eight kj through a wave-private 4.6 KB LDS region; both transposes are conflict-free
//      under the per-instruction banking of MI355X_MICROARCH.md (checked by tools/lds_bank_check.py):
//        T1: element 72 c + lane            (write, c = position of kj in its round) / 72 c' + 8 e + l0'   (read by lane 8 c' + l0', e = l1)
//        T2: element 65 l0' + 8 c' + ka     (write)             / 65 e + lane         (read by lane 8 c'' + ka'', e = l0)
// A round is one residue class of the carriers: round r = the kj = r + 4 c, c < 8, i.e. the carriers k = r (mod 4).  With the
// pilots on 1:4:end (the benchmark layout) round 0 holds nothing but pilot carriers, which only a frame's first symbol needs
// (it comes from the pilot stage's stash): data symbols skip that round and everything of the register transform that feeds it
// (template parameter SKIP) -- a fifth of the instructions.
// The registers of a round are dead once they are in LDS, so the NEXT symbol's samples are loaded straight into them:
// a register prefetch that costs no registers and no copies (the four-wavefront form copies 16 VGPRs per symbol).  With the
// residue-class rounds the freed registers are the next class r in a DIFFERENT register layout (wv_reg), so the transform
// exists in two layouts that alternate and the symbol loop runs in pairs.
// Two LDS round trips per sample instead of three.
#include <algorithm>
#include <type_traits>

#include "chain_fast_core.hpp"

namespace ofdm {

constexpr int WV_N = 2048;
constexpr int WV_TR_ELEMS = 576;          // wave-private transpose region (largest index 72*7+63 = 567, 65*7+63 = 518)
constexpr int WV_TW_ROWS = 31;            // W_2048^(lane * kj), kj = 1..31
constexpr int WV_WPB = 4;              // wavefronts (= frames in flight) per workgroup; three workgroups per CU

// dynamic LDS: W_2048 table [31][64] | W_64 table [7][64] | data positions [4][64] (8-byte stride) | per-wave regions
constexpr unsigned WV_OFF_TW = 0, WV_OFF_TWB = 8 * 64 * WV_TW_ROWS, WV_OFF_DD = WV_OFF_TWB + 8 * 64 * 7;
constexpr unsigned WV_OFF_LUT = WV_OFF_DD + 8 * 64 * 4;              // decision thresholds of demap_square_lut
constexpr unsigned WV_OFF_WAVE = WV_OFF_LUT + 256;
constexpr unsigned WV_TRASH_OFF = 8 * WV_TR_ELEMS, WV_DESCR_OFF = WV_TRASH_OFF + 8 * 64, WV_CODES_OFF = WV_DESCR_OFF + 16;
// (WV_DESCR_OFF: one word per wavefront, the DeScrambler state between the pack batches of a frame -- in LDS so that it
//  costs no register across the symbol loop)
struct WaveLayout {
  unsigned wave_bytes, total;
  int cb;                                 // symbols per pack batch
};

// W_32^m = exp(-2 pi i m / 32), m compile-time after unrolling
__device__ __forceinline__ cx<float> w32(int m) {
  constexpr float C[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                          0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                          0.19509032201612826785f, 0.0f};
  m &= 31;
  // cos / sin of 2 pi m / 32 by quadrant symmetry
  const int q = m >> 3, r = m & 7;
  const float c = C[r], s = C[8 - r];
  switch (q) {
    case 0: return mk<float>(c, -s);
    case 1: return mk<float>(-s, -c);
    case 2: return mk<float>(-c, s);
    default: return mk<float>(s, c);
  }
}

// Register that holds element (class j0, position m) of the 32-point register transform -- sample j0 + 4 m before the dft8s,
// (j0, ka = m) after them, Z[ka + 8 kb] = (kb = j0, ka = m) after the dft4s:
//   layout 0: j0 + 4 m                          (a class is a stride-4 set, a position a quad)
//   layout 1: 4 j0 + 16 (m >> 2) + (m & 3)      (a class is two quads, a position a stride-4 set)
// Round r reads the positions ka = r, r + 4 of every class: in layout 0 the quads r and r + 4 = class r of layout 1, in layout 1
// the registers = r (mod 4) = class r of layout 0.
__host__ __device__ constexpr int wv_reg(int L, int j0, int m) { return L == 0 ? j0 + 4 * m : 4 * j0 + 16 * (m >> 2) + (m & 3); }

// LUT: level rank by arithmetic guess + the two neighbouring thresholds from LDS (demap_square_lut) instead of counting all
// 2^BA - 1 thresholds
// SL = 0: count all 2^BA - 1 thresholds (demap_square); 1: LUT; 2: arithmetic rank around the exactly compared centre
template <int BA, int SL>
__device__ __forceinline__ int slice_wave(const DemapTable<float>& tab, const float* lut, cx<float> z) {
  if constexpr (SL == 1 && BA >= 2) return demap_square_lut<float, BA>(tab, lut, z);
  else if constexpr (SL == 2 && BA >= 2) return demap_square_arith<BA>(tab, z);
  else return slice_symbol<float, BA, true>(tab, z);
}

template <int BA, bool HEXT, int WPB = WV_WPB, int ABL = 0, bool WBUF = true, int LUT = 2, int SKIP = 0, bool DESCR = false>
__global__ __launch_bounds__(64 * WPB, WPB == 8 ? 4 : 3) void rx_symbols_wave_kernel(FastParams<float> P, WaveLayout lay,
                                                                  const cx<float>* __restrict__ rx, int64_t n_frames,
                                                                  uint32_t* __restrict__ bits_out,
                                                                  const uint32_t* __restrict__ ref_bits,
                                                                  uint32_t* __restrict__ errors_out,
                                                                  cx<float>* __restrict__ h_out,
                                                                  int32_t* __restrict__ index_out, DemapTable<float> tab) {
  using T = float;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Compile-time LDS layout (WV_OFF_*): every address below is one of four per-lane registers (lane8, trl, t1r, t2w) plus
  // an immediate offset of the DS instruction -- run-time table offsets cost a VGPR each, and the kernel has none to spare.
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves_wg = blockDim.x >> 6;
  const unsigned lane8 = 8u * lane;
  auto tw32_at = [&](int row) { return *(const cx<T>*)(smem + WV_OFF_TW + 512 * row + lane8); };     // W_2048^(lane (row+1))
  auto twbl_at = [&](int row) { return *(const cx<T>*)(smem + WV_OFF_TWB + 512 * row + lane8); };    // W_64^((row+1) (lane&7))
  {
    cx<T>* const tw32 = (cx<T>*)(smem + WV_OFF_TW);
    cx<T>* const twbl = (cx<T>*)(smem + WV_OFF_TWB);
    for (int i = threadIdx.x; i < WV_TW_ROWS * 64; i += blockDim.x) {
      const int kj = i / 64 + 1, l = i & 63;
      tw32[i] = P.tw[(kj * l) & (WV_N - 1)];
    }
    for (int i = threadIdx.x; i < 7 * 64; i += blockDim.x) {
      const int t = i / 64 + 1, l = i & 63;
      twbl[i] = P.tw[t * (l & 7) * 32];
    }
  }
  const int nd = P.nd, taps = P.taps, n_symb = P.n_symb, n_carrier = P.n_carrier;
  const int Lsym = WV_N + P.t_guard;
  const int64_t Lframe = (int64_t)Lsym * n_symb;
  const int CB = lay.cb;
  // carriers of this lane: round r, kb -> k = 32 (ka + 8 kb) + r + 4 c,  lane = 8 c + ka  (a round = the kj = r + 4 c, c < 8:
  // one residue class mod 4 of the carriers -- a comb-4 pilot class is a whole round, which data symbols then skip)
  const int c_out = lane >> 3, ka_out = lane & 7;
  auto kk_of = [&](int t) { return 32 * (ka_out + 8 * (t & 1)) + (t >> 1) + 4 * c_out; };
  // data position of output t (or 0xffff): table [4 rounds][64 lanes], two to a word, the same for every wavefront
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k0 = kk_of(2 * r), k1 = kk_of(2 * r + 1);
      const unsigned d0 = k0 < n_carrier ? (unsigned)(unsigned short)P.drole[k0] : 0xffffu;
      const unsigned d1 = k1 < n_carrier ? (unsigned)(unsigned short)P.drole[k1] : 0xffffu;
      *(unsigned*)(smem + WV_OFF_DD + 512 * r + lane8) = d0 | (d1 << 16);
    }
  }
  const T* const lut = (const T*)(smem + WV_OFF_LUT);
  if constexpr (BA >= 2) demap_lut_fill<T, BA>(tab, (T*)(smem + WV_OFF_LUT), threadIdx.x);
  __syncthreads();                                                     // the only workgroup barrier of the kernel
  // wave-private region: transposes [576 complex] | 64 spare 8-byte slots | codes
  const unsigned wbase = WV_OFF_WAVE + (unsigned)wave * lay.wave_bytes;
  unsigned char* const trl = smem + wbase + lane8;
  uint8_t* const codes = smem + wbase + WV_CODES_OFF;
  // a decided symbol of a carrier that holds no data goes to the lane's spare slot: no branch around the slicer
  auto code_ptr = [&](uint8_t* base, int t) {
    const unsigned d = (*(const unsigned*)(smem + WV_OFF_DD + 512 * (t >> 1) + lane8) >> (16 * (t & 1))) & 0xffffu;
    return d == 0xffffu ? (uint8_t*)(trl + WV_TRASH_OFF) : base + d;
  };
  // LDS addresses of the two transposes
  cx<T>* const t1w = (cx<T>*)trl;                                                              // + 72 c
  cx<T>* const t1r = (cx<T>*)(smem + wbase) + 72 * (lane >> 3) + (lane & 7);                   // + 8 e
  cx<T>* const t2w = (cx<T>*)(smem + wbase) + 65 * (lane & 7) + 8 * (lane >> 3);               // + ka
  cx<T>* const t2r = (cx<T>*)trl;                                                              // + 65 e

  const int64_t wave_id = (int64_t)blockIdx.x * n_waves_wg + wave, n_waves = (int64_t)gridDim.x * n_waves_wg;
  cx<T> v[32];
  // WBUF: the samples j = 4 c + 3 -- those whose registers only come free in the last round -- travel through eight
  // registers of their own and are requested a whole symbol ahead (right after the previous copy has been consumed):
  // every request then has at least ~45 % of a symbol period to land instead of ~30 %
  cx<T> w[8];
  // samples of (frame, symbol 2) of this wavefront's first frame
  if (wave_id < n_frames && n_symb > 1) {
    const cx<T>* src = rx + wave_id * Lframe + Lsym + P.t_guard + lane;
    // in the order the rounds below refill them (j = 4 c + r, round by round): the waits in front of the four dft8 of the
    // register transform then count the same outstanding loads on every path into the symbol loop
#pragma unroll
    for (int r = 0; r < (WBUF ? 3 : 4); ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c) v[4 * c + r] = nt_load(src + 64 * (4 * c + r));
    if constexpr (WBUF) {
#pragma unroll
      for (int c = 0; c < 8; ++c) w[c] = nt_load(src + 64 * (4 * c + 3));
    }
  }
  // One frame whose first transform finds its samples in register layout S0 (wv_reg).  The symbol loop takes the transforms in
  // pairs (layout S0, then the other), so its back-edge always carries layout S0: every register index is static and no value
  // ever changes registers.  (A run-time choice of the layout per symbol made the allocator permute 32 values at the join:
  // 2600 spill instructions, 1.6x slower.)
  auto do_frame = [&](auto SC, int64_t f) __attribute__((always_inline)) {
    constexpr int S0 = decltype(SC)::value;
    const cx<T>* frx = rx + f * Lframe;
    const int64_t fnext = f + n_waves;
    cx<T> geq[8];
    // ---- G = 1 ./ H on this lane's carriers (OMP_estimate.m:36, equalize_signal.m:6)
    if constexpr (HEXT) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        geq[t] = mk<T>(0, 0);
        if (kk_of(t) < n_carrier) {
          const cx<T> H = P.h_in[f * n_carrier + kk_of(t)];
          if (h_out) h_out[f * n_carrier + kk_of(t)] = H;
          geq[t] = cdiv(mk<T>(1, 0), H);
        }
      }
    } else {
      double hr[8], hi[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) hr[t] = hi[t] = 0.0;
      for (int q = 0; q < taps; ++q) {
        const int idx = P.tap_idx[f * taps + q];                       // wave-uniform address
        const c64 x = P.tap_x[f * taps + q];
        if (index_out && lane == 0) index_out[f * taps + q] = idx + 1;
        const int id0 = idx < 0 ? 0 : idx;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          // W_N^(idx k): exponent reduced exactly in integers, v_sin / v_cos take the angle in turns
          const int e = (id0 * kk_of(t)) & (WV_N - 1);
          const float turns = (float)e * (1.0f / (float)WV_N);
          const float ws = __builtin_amdgcn_sinf(turns), wc = __builtin_amdgcn_cosf(turns);
          hr[t] += x.x * (double)wc + x.y * (double)ws;                // w = (wc, -ws)
          hi[t] += x.y * (double)wc - x.x * (double)ws;
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        geq[t] = mk<T>(0, 0);
        if (kk_of(t) < n_carrier) {
          const cx<T> H = mk<T>((T)hr[t], (T)hi[t]);
          if (h_out) h_out[f * n_carrier + kk_of(t)] = H;
          geq[t] = cdiv(mk<T>(1, 0), H);
        }
      }
    }
    // ---- symbol 1 from the stash of the pilot stage
#pragma unroll
    for (int t = 0; t < 8; ++t)
      *code_ptr(codes, t) = (uint8_t)slice_wave<BA, LUT>(
          tab, lut, (kk_of(t) < n_carrier ? P.stash[f * n_carrier + kk_of(t)] : mk<T>(0, 0)) * geq[t]);
    unsigned err = 0;
    int slot = 1;                                                      // symbols in the codes buffer
    int64_t code0 = 0;                                                 // first code index of the buffer within the frame
    uint32_t* const dslot = (uint32_t*)(smem + wbase + WV_DESCR_OFF);  // DeScrambler state at the start of the pack batch
    if constexpr (DESCR) { if (lane == 0) *dslot = P.descr; }
    auto symbol = [&](auto LC, int s) __attribute__((always_inline)) {
      constexpr int L = decltype(LC)::value;
      // where the registers of a finished round are refilled from: the next symbol of this wavefront's stream
      // (past the last symbol of the last frame: the same symbol again, so that the loads are unconditional)
      const cx<T>* nsrc = frx + (int64_t)s * Lsym + P.t_guard + lane;
      if (s + 1 < n_symb) nsrc += Lsym;
      else if (fnext < n_frames) nsrc = rx + fnext * Lframe + Lsym + P.t_guard + lane;
      uint8_t* const cslot = codes + slot * nd;
      // One transform in register layout L (wv_reg): round r reads -- and thereby frees -- exactly the registers that hold
      // class r of the NEXT transform in the other layout, so the layouts alternate and every register index stays static.
      {
        // ---- 1. 32-point DFT over j = j0 + 4 j1 in registers: element (j0, ka) after the four dft8, Z[ka + 8 kb] after the dft4
#pragma unroll
        for (int j0 = 0; j0 < 4; ++j0) {
          if (WBUF && j0 == 3) {
            // transform the look-ahead registers where they are, hand the results to the (free) slots of class 3, THEN refill
            dft8<T, false>(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
#pragma unroll
            for (int ka = 0; ka < 8; ++ka) v[wv_reg(L, 3, ka)] = w[ka];
            if constexpr (ABL != 1) {
#pragma unroll
              for (int m = 0; m < 8; ++m) w[m] = nt_load(nsrc + 64 * (3 + 4 * m));
            }
          } else {
            dft8<T, false>(v[wv_reg(L, j0, 0)], v[wv_reg(L, j0, 1)], v[wv_reg(L, j0, 2)], v[wv_reg(L, j0, 3)], v[wv_reg(L, j0, 4)],
                           v[wv_reg(L, j0, 5)], v[wv_reg(L, j0, 6)], v[wv_reg(L, j0, 7)]);
          }
          if (j0 > 0) {
#pragma unroll
            for (int ka = 1; ka < 8; ++ka)
              if (!((SKIP >> (ka & 3)) & 1)) v[wv_reg(L, j0, ka)] = v[wv_reg(L, j0, ka)] * w32(j0 * ka);
          }
        }
#pragma unroll
        for (int ka = 0; ka < 8; ++ka)
          if (!((SKIP >> (ka & 3)) & 1))
            dft4<T, false>(v[wv_reg(L, 0, ka)], v[wv_reg(L, 1, ka)], v[wv_reg(L, 2, ka)], v[wv_reg(L, 3, ka)]);
        // ---- 2.-3. four rounds of eight kj = r + 4 c  (ka = r + 4 (c & 1), kb = c >> 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool skip = (SKIP >> r) & 1;                          // compile-time after unrolling
          cx<T> z_keep[8];
          if (!skip) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const int kj = r + 4 * c, ka = r + 4 * (c & 1), kb = c >> 1;
              const cx<T> z = kj == 0 ? v[wv_reg(L, kb, ka)] : v[wv_reg(L, kb, ka)] * tw32_at(kj - 1);
              t1w[72 * c] = z;
              z_keep[c] = z;
            }
          } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) z_keep[c] = v[wv_reg(L, c >> 1, r + 4 * (c & 1))];
          }
          // the round's registers are free: next symbol's samples of class r, in the other layout
          if constexpr (ABL == 1) {                                  // diagnostic build: issue time without the sample stream
#pragma unroll
            for (int m = 0; m < 8; ++m) v[wv_reg(1 - L, r, m)] = mk<T>(z_keep[m].y, z_keep[m].x);
          } else if (!(WBUF && r == 3)) {
#pragma unroll
            for (int m = 0; m < 8; ++m) v[wv_reg(1 - L, r, m)] = nt_load(nsrc + 64 * (r + 4 * m));
          }
          if (!skip) {
            wave_sync();
            cx<T> u[8];
            lds_read8<8, true>(u, t1r);
            wave_sync();
            dft8<T, false>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]);
#pragma unroll
            for (int t = 1; t < 8; ++t) u[t] = u[t] * twbl_at(t - 1);
#pragma unroll
            for (int t = 0; t < 8; ++t) t2w[t] = u[t];
            wave_sync();
            lds_read8<65, true>(u, t2r);
            wave_sync();
            dft8_first2<T>(u);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
              *code_ptr(cslot, 2 * r + kb) = (uint8_t)slice_wave<BA, LUT>(tab, lut, u[kb] * geq[2 * r + kb]);
          }
        }
      }
      ++slot;
      // ---- pack a full batch (or the frame's last one): bit i of the frame -> byte i/8, bit 7 - i%8; BER numerator
      if (slot == CB || s + 1 == n_symb) {
        const int n_codes = slot * nd;
        const int pad_to = (n_codes + 31) & ~31;
        if (lane < pad_to - n_codes) codes[n_codes + lane] = 0;
        wave_sync();
        const int woff = (int)(code0 >> 5) * P.bps;                   // code0 is a multiple of 32
        uint32_t dprev = 0;
        if constexpr (DESCR) dprev = *dslot;
        err += pack_frame_t<2 * BA, DESCR>(codes, n_codes, P.bps, P.frame_words - woff,
                                  bits_out ? bits_out + f * P.frame_words + woff : nullptr,
                                  ref_bits ? ref_bits + f * P.frame_words + woff : nullptr, lane, 64, dprev);
        // DeScrambler on: the next batch continues the frame's stream -- hand it the last 14 bits of this one (a batch that
        // is not the frame's last ends on a 32-code boundary)
        if (DESCR && s + 1 < n_symb) {
          if (lane == 0) *dslot = DESCR_ON | (descr_tail(codes, n_codes, P.bps) & 0x3fffu);
        }
        wave_sync();
        code0 += n_codes;
        slot = 0;
      }
    };
    for (int s = 1; s < n_symb; s += 2) {
      symbol(std::integral_constant<int, S0>{}, s);
      if (s + 1 < n_symb) symbol(std::integral_constant<int, 1 - S0>{}, s + 1);     // (an odd count leaves through here)
    }
    if (n_symb == 1) {                                                 // only the stash symbol
      const int n_codes = nd, pad_to = (n_codes + 31) & ~31;
      if (lane < pad_to - n_codes) codes[n_codes + lane] = 0;
      wave_sync();
      err += pack_frame_t<2 * BA, DESCR>(codes, n_codes, P.bps, P.frame_words, bits_out ? bits_out + f * P.frame_words : nullptr,
                                         ref_bits ? ref_bits + f * P.frame_words : nullptr, lane, 64, P.descr);
      wave_sync();
    }
    if (ref_bits && errors_out) {
      for (int off = 32; off > 0; off >>= 1) err += __shfl_xor(err, off, 64);
      if (lane == 0) errors_out[f] = err;
    }
  };
  // a frame of an odd number of transforms leaves the samples of the next frame in the other layout
  const bool odd = ((n_symb - 1) & 1) != 0;
  for (int64_t f = wave_id; f < n_frames;) {
    do_frame(std::integral_constant<int, 0>{}, f);
    f += n_waves;
    if (odd) {
      if (f >= n_frames) break;
      do_frame(std::integral_constant<int, 1>{}, f);
      f += n_waves;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool wave_layout(int nd, int n_symb, int wpb, WaveLayout& lay) {
  // symbols per pack batch: every batch but the last must end on a 32-code boundary
  int cb = 1;
  while ((cb * nd) % 32 != 0) ++cb;                                    // cb <= 32
  const int unit = cb;
  while (cb * nd < 1920 && cb < n_symb) cb += unit;
  if (cb >= n_symb) cb = n_symb;                                       // one batch: the whole frame
  const unsigned codes_bytes = (unsigned)(((size_t)cb * nd + 31 + 32) & ~size_t(31));
  if (codes_bytes > 6144) return false;
  lay.cb = cb;
  lay.wave_bytes = (WV_CODES_OFF + codes_bytes + 15) & ~15u;
  lay.total = WV_OFF_WAVE + (unsigned)wpb * lay.wave_bytes;
  return lay.total <= (wpb == 8 ? 78u : 52u) * 1024;                   // two / three workgroups per CU
}

bool chain_wave_supported(const FastPlanView& pv) {
  if (getenv("OFDM_FAST_NO_WAVE")) return false;
  if ((pv.descr & DESCR_ON) && pv.d_wt != nullptr) return false;      // DeScrambler + MMSE mode: the four-wavefront symbol stage
  if (pv.f64 || pv.nfft != WV_N || pv.n_carrier > WV_N / 4 || pv.taps > FAST_MAXT) return false;
  WaveLayout lay;
  return wave_layout(pv.nd, pv.n_symb, 4, lay) && wave_layout(pv.nd, pv.n_symb, 8, lay);
}

int chain_wave_symbols_run(const FastPlanView& pv, const FastParams<float>& P, const void* rx, int64_t n_frames, void* bits,
                           const void* ref, void* errs, void* h_out, void* idx_out) {
  // two builds of the kernel: 4 wavefronts per workgroup at <= 168 VGPRs (three workgroups = 12 wavefronts per CU) and
  // 8 per workgroup at <= 128 VGPRs (two workgroups = 16 wavefronts per CU)
  const int wpb = WV_WPB;       // (eight wavefronts per workgroup at <= 128 VGPRs, the LUT slicer and the form without look-ahead
                                //  registers were measured in round 2 -- 965 / 937 / +1.5 % against 911 us -- and are no longer built)
  WaveLayout lay;
  OFDM_ARG(wave_layout(pv.nd, pv.n_symb, wpb, lay), "rx_chain_task5(wave): frame does not fit the wave-per-frame stage");
  DemapTable<float> tab;
  fill_demap_table<float>(*pv.dict, *pv.cinfo, tab);
  const bool mmse = pv.d_wt != nullptr;
  auto launch = [&](auto kern) -> int {
    int per_cu = resident_blocks_per_cu((const void*)kern, 64 * wpb, lay.total);
    if (const char* e = getenv("OFDM_WAVE_WG_PER_CU")) per_cu = std::max(1, atoi(e));
    const int64_t want = (n_frames + wpb - 1) / wpb;
    const unsigned grid = (unsigned)std::min<int64_t>(want, (int64_t)ctx().num_cu * per_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), lay.total, ctx().stream, P, lay, (const cx<float>*)rx, n_frames,
                       (uint32_t*)bits, (const uint32_t*)ref, (uint32_t*)errs, (cx<float>*)h_out, (int32_t*)idx_out, tab);
    return OFDM_OK;
  };
  const int ba = pv.cinfo->kind == 1 ? pv.cinfo->bits_per_axis : 0;
  const bool exact = getenv("OFDM_WAVE_EXACT_SLICER") != nullptr;
  // no data carrier with index = 0 (mod 4) -- pilots "1:4:end", the benchmark layout: the round of that residue class is not
  // computed on data symbols (a quarter of the post-register work and of the register transform's second stage)
  const bool skip0 = (pv.data_mod4 & 15) == 14 && !getenv("OFDM_WAVE_NO_SKIP");
  const bool skip02 = (pv.data_mod4 & 15) == 10 && !getenv("OFDM_WAVE_NO_SKIP");      // pilots 1:2:end: two of the four rounds
#define WAVE_CASE(BAV, HX)                                                                \
  if (exact) OFDM_TRY(launch(rx_symbols_wave_kernel<BAV, HX, 4, 0, true, 0>));            \
  else if (skip0) OFDM_TRY(launch(rx_symbols_wave_kernel<BAV, HX, 4, 0, true, 2, 1>));    \
  else if (skip02) OFDM_TRY(launch(rx_symbols_wave_kernel<BAV, HX, 4, 0, true, 2, 5>));   \
  else OFDM_TRY(launch(rx_symbols_wave_kernel<BAV, HX, 4>))
  // per-frame DeScrambler in the pack stage (ofdm_rx_plan_set_descrambler): its own instantiations, OMP mode only
#define WAVE_CASE_D(BAV)                                                                            \
  if (exact) OFDM_TRY(launch(rx_symbols_wave_kernel<BAV, false, 4, 0, true, 0, 0, true>));          \
  else if (skip0) OFDM_TRY(launch(rx_symbols_wave_kernel<BAV, false, 4, 0, true, 2, 1, true>));     \
  else if (skip02) OFDM_TRY(launch(rx_symbols_wave_kernel<BAV, false, 4, 0, true, 2, 5, true>));    \
  else OFDM_TRY(launch(rx_symbols_wave_kernel<BAV, false, 4, 0, true, 2, 0, true>))
  if (P.descr & DESCR_ON) {
    OFDM_ARG(!mmse, "rx_chain_task5(wave): the DeScrambler variant exists in OMP mode only");
    switch (ba) {
      case 2: WAVE_CASE_D(2); break;
      case 3: WAVE_CASE_D(3); break;
      case 4: WAVE_CASE_D(4); break;
      default: WAVE_CASE_D(0); break;
    }
  } else if (mmse) {
    switch (ba) {
      case 2: WAVE_CASE(2, true); break;
      case 3: WAVE_CASE(3, true); break;
      case 4: WAVE_CASE(4, true); break;
      default: WAVE_CASE(0, true); break;
    }
  } else {
    switch (ba) {
      case 2: WAVE_CASE(2, false); break;
      case 3:
#ifdef OFDM_DIAG   // tools-only build (libofdm_mi355x_diag.so, build.py --diag): issue time without the sample stream
        if (getenv("OFDM_WAVE_ABL")) { OFDM_TRY(launch(rx_symbols_wave_kernel<3, false, 4, 1>)); break; }
#endif
        WAVE_CASE(3, false);
        break;
      case 4: WAVE_CASE(4, false); break;
      default: WAVE_CASE(0, false); break;
    }
  }
#undef WAVE_CASE
#undef WAVE_CASE_D
  return check_launch("rx_symbols_wave_kernel");
}

}  // namespace ofdm
