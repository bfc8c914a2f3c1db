// OFDM_demodulator of the batched Task-4 receiver (ofdm_rx_chain_task4, T4/Main_model_Task_4.m:292-310) with ONE WAVEFRONT PER
// SYMBOL RUN: Nfft = 2048, fp32, N_carrier <= 1024 (BASELINE config 3).
//
// The four-wavefronts-per-symbol form (t4_demod_kernel, ofdm_sync.hip) pays four workgroup barriers per symbol and transforms
// all 2048 bins; here a run of consecutive symbols of one frame never leaves its wavefront (the decomposition of the metric
// kernel, ofdm_chain_wave.hip): lane l holds x[l + 64 j], j < 32 -- 32 coalesced 8-byte nontemporal loads per symbol, taken
// from rx THROUGH the STO fix (add_STO(Rx, TgPosition); add_STO(., -(Nfft+T_guard)), :292-294) and multiplied by the merged
// rotor of add_CFO(., -FreqOffset) (:301) and remove_IFO's add_CFO(., -IFO) (remove_IFO.m:9) as they arrive --
//   1. 32-point DFT over j in registers (four dft8, constant W_32 twiddles, eight dft4)            -> Z[kj], kj = ka + 8 kb
//   2. Z[kj] *= W_2048^(l kj)                                                       (LDS table, lane-contiguous)
//   3. four rounds of eight CONSECUTIVE kj (round kb: the registers of sample class j = kb (mod 4), which are refilled with the
//      next symbol's samples of that class the moment they are in LDS): a 64-point DFT across the lanes as 8 x 8 through the
//      two conflict-free transposes of the metric kernel, the last radix-8 computing 4 of its 8 outputs (carriers < 1024)
// and writes rows 1..N_carrier of the column plus its pilot rows compactly (what fine_sync / estimate_channel sweep).
// A round's outputs are carriers c + 8 kb + 32 ka + 256 kb': eight 64-byte runs per store instruction.
// No s_barrier after the table fill, no atomics; 128 VGPRs -> four wavefronts per SIMD.
#include <algorithm>

#include "chain_fast_core.hpp"

namespace ofdm {

constexpr int T4W_N = 2048;
constexpr int T4W_TR_ELEMS = 576;                      // wave-private transpose region (largest index 72*7+63, 65*7+63)
constexpr int T4W_WPB = 4;
constexpr unsigned T4W_OFF_TW = 0, T4W_OFF_TWB = 8 * 64 * 31, T4W_OFF_PR = T4W_OFF_TWB + 8 * 64 * 7;
constexpr unsigned T4W_OFF_WAVE = T4W_OFF_PR + 2 * 64 * 16;
constexpr unsigned T4W_WAVE_BYTES = 8 * T4W_TR_ELEMS;
constexpr unsigned T4W_LDS = T4W_OFF_WAVE + T4W_WPB * T4W_WAVE_BYTES;

// W_32^m = exp(-2 pi i m / 32), m compile-time after unrolling
__device__ __forceinline__ cx<float> t4w_w32(int m) {
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

// X0 .. X3 of an 8-point DFT (forward), in v[0..3]
__device__ __forceinline__ void dft8_first4(cx<float> (&v)[8]) {
  using T = float;
  const T h = T(0.70710678118654752440084436210485);
  const cx<T> a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
  const cx<T> b0 = v[0] - v[4];
  cx<T> b1 = v[1] - v[5], b2 = v[2] - v[6], b3 = v[3] - v[7];
  b1 = mk<T>((b1.x + b1.y) * h, (b1.y - b1.x) * h);
  b2 = mk<T>(b2.y, -b2.x);
  b3 = mk<T>((b3.y - b3.x) * h, (-b3.x - b3.y) * h);
  v[0] = (a0 + a2) + (a1 + a3);                               // X0
  v[2] = (a0 - a2) + mul_mi<T, false>(a1 - a3);               // X2
  v[1] = (b0 + b2) + (b1 + b3);                               // X1
  v[3] = (b0 - b2) + mul_mi<T, false>(b1 - b3);               // X3
}

template <bool FD>
__global__ __launch_bounds__(64 * T4W_WPB, 3) void t4_demod_wave_kernel(const cx<float>* __restrict__ rx, cx<float>* __restrict__ X,
                                                                       const cx<float>* __restrict__ tw, int64_t len, int t_guard,
                                                                       int n_symb, int64_t n_frames, int time_desync,
                                                                       const int64_t* __restrict__ tg, const double* __restrict__ fo,
                                                                       const int32_t* __restrict__ ifo, int n_keep,
                                                                       cx<float>* __restrict__ xp, const int16_t* __restrict__ prole,
                                                                       int np, int spc) {
  using T = float;
  constexpr int N = T4W_N;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lane8 = 8u * lane;
  auto tw32_at = [&](int row) { return *(const cx<T>*)(smem + T4W_OFF_TW + 512 * row + lane8); };     // W_2048^(lane (row+1))
  auto twbl_at = [&](int row) { return *(const cx<T>*)(smem + T4W_OFF_TWB + 512 * row + lane8); };    // W_64^((row+1) (lane&7))
  {
    cx<T>* const t32 = (cx<T>*)(smem + T4W_OFF_TW);
    cx<T>* const tbl = (cx<T>*)(smem + T4W_OFF_TWB);
    for (int i = threadIdx.x; i < 31 * 64; i += blockDim.x) {
      const int kj = i / 64 + 1, l = i & 63;
      t32[i] = tw[(kj * l) & (N - 1)];
    }
    for (int i = threadIdx.x; i < 7 * 64; i += blockDim.x) {
      const int t = i / 64 + 1, l = i & 63;
      tbl[i] = tw[t * (l & 7) * 32];
    }
    // output t = 4 kb + kb' of lane 8 c + ka is carrier k = c + 8 kb + 32 ka + 256 kb':  its pilot row, -1 (kept, no pilot)
    // or -2 (beyond N_carrier: not stored)
    int16_t* const pr = (int16_t*)(smem + T4W_OFF_PR);
    for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) {
      const int t = i / 64, l = i & 63;
      const int k = (l >> 3) + 8 * (t >> 2) + 32 * (l & 7) + 256 * (t & 3);
      pr[i] = k < n_keep ? (xp ? prole[k] : (int16_t)-1) : (int16_t)-2;
    }
  }
  __syncthreads();                                                     // the only workgroup barrier of the kernel
  const int16_t* const prt = (const int16_t*)(smem + T4W_OFF_PR) + lane;
  const unsigned wbase = T4W_OFF_WAVE + (unsigned)wave * T4W_WAVE_BYTES;
  cx<T>* const t1w = (cx<T>*)(smem + wbase + lane8);                                          // + 72 c
  cx<T>* const t1r = (cx<T>*)(smem + wbase) + 72 * (lane >> 3) + (lane & 7);                   // + 8 e
  cx<T>* const t2w = (cx<T>*)(smem + wbase) + 65 * (lane & 7) + 8 * (lane >> 3);               // + ka
  cx<T>* const t2r = (cx<T>*)(smem + wbase + lane8);                                          // + 65 e
  const int k_lane = (lane >> 3) + 32 * (lane & 7);                    // carrier of output (kb = 0, kb' = 0)

  const int sym_len = N + t_guard;
  const int chunks = (n_symb + spc - 1) / spc;
  const int64_t n_items = n_frames * chunks;
  const int64_t n_waves = (int64_t)gridDim.x * T4W_WPB;
  const double inv = 1.0 / (double)N;

  // where symbol sy of frame f is read from and how many of its samples exist: with the STO fix stream index i reads
  // rx[i - sym_len + TgPosition] when that lies inside the frame, zero otherwise (t4_raw) -- the missing part is always a tail
  // (or the whole blanked first symbol)
  auto source = [&](int64_t f, int sy, const cx<T>*& src, int& mlim) {
    const cx<T>* xf = rx + f * len;
    if (!time_desync) { src = xf + (int64_t)sy * sym_len + t_guard + lane; mlim = N; return; }
    const int64_t pos = tg[f];
    const int64_t b = (int64_t)(sy - 1) * sym_len + t_guard + pos;     // source index of sample m = 0
    src = xf + b + lane;
    const int64_t room = sy >= 1 ? len - b : 0;                        // samples m < room exist (b >= 0 when sy >= 1)
    mlim = room >= N ? N : (room > 0 ? (int)room : 0);
  };
  cx<T> v[32];
  const cx<T>* src = rx;
  int mlim = 0;
  const int64_t it0 = (int64_t)blockIdx.x * T4W_WPB + wave;
  if (it0 < n_items) {                                                 // samples of the first symbol of the first run
    const int64_t f = it0 / chunks;
    source(f, (int)(it0 - f * chunks) * spc, src, mlim);
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = lane + 64 * j < mlim ? nt_load(src + 64 * j) : mk<T>(0, 0);
  }
  for (int64_t it = it0; it < n_items; it += n_waves) {
    const int64_t f = it / chunks;
    const int s0 = (int)(it - f * chunks) * spc;
    const int s1 = min(s0 + spc, n_symb);
    // rotor of the merged rotation at stream index i: turns = frac(-FreqOffset i / N) - (IFO i mod N) / N, exactly reduced
    // (turns_of); the rotors of the fixed distances 64, 512, 1024, 1536 are frame constants, kept in scalar registers
    double c1 = 0.0;
    int fi = 0;
    auto turns_of = [&](int64_t i) -> float {
      double t = c1 * (double)i * inv;
      t -= floor(t);
      if (fi > 0) { t -= (double)(((int64_t)fi * i) & (N - 1)) * inv; t += t < 0.0 ? 1.0 : 0.0; }
      return (float)t;
    };
    auto uniform_rotor = [&](int64_t d) {
      float sn, cs;
      sincospif(2.0f * turns_of(d), &sn, &cs);
      return mk<T>(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cs))),
                   __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sn))));
    };
    cx<T> step = mk<T>(1, 0), e8 = step, e16 = step, e24 = step;
    if constexpr (FD) {
      c1 = -fo[f];
      fi = ifo[f];
      step = uniform_rotor(64);
      e8 = uniform_rotor(512);
      e16 = uniform_rotor(1024);
      e24 = uniform_rotor(1536);
    }
    for (int sy = s0; sy < s1; ++sy) {
      // what the registers of a finished round are refilled with: the next symbol of the run, past its end the first symbol
      // of this wavefront's next run, past the last run nothing
      const cx<T>* nsrc = src;
      int nlim = 0;
      if (sy + 1 < s1) {
        source(f, sy + 1, nsrc, nlim);
      } else if (it + n_waves < n_items) {
        const int64_t f2 = (it + n_waves) / chunks;
        source(f2, (int)(it + n_waves - f2 * chunks) * spc, nsrc, nlim);
      }
      if constexpr (FD) {
        // exact phase of this lane's first sample; the anchors j = 8, 16, 24 by the frame constants, chains of eight steps
        float sn, cs;
        sincospif(2.0f * turns_of((int64_t)sy * sym_len + t_guard + lane), &sn, &cs);
        const cx<T> r0 = mk<T>(cs, sn);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          cx<T> cur = q == 0 ? r0 : r0 * (q == 1 ? e8 : (q == 2 ? e16 : e24));
#pragma unroll
          for (int j = 8 * q; j < 8 * q + 8; ++j) {
            v[j] = v[j] * cur;
            if (j + 1 < 8 * q + 8) cur = cur * step;
          }
        }
      }
      // ---- 1. 32-point DFT over j = j0 + 4 m in registers: Z[ka + 8 kb] ends in v[kb + 4 ka]
#pragma unroll
      for (int j0 = 0; j0 < 4; ++j0) {
        dft8<T, false>(v[j0], v[j0 + 4], v[j0 + 8], v[j0 + 12], v[j0 + 16], v[j0 + 20], v[j0 + 24], v[j0 + 28]);
        if (j0 > 0) {
#pragma unroll
          for (int ka = 1; ka < 8; ++ka) v[j0 + 4 * ka] = v[j0 + 4 * ka] * t4w_w32(j0 * ka);
        }
      }
#pragma unroll
      for (int ka = 0; ka < 8; ++ka) dft4<T, false>(v[4 * ka], v[4 * ka + 1], v[4 * ka + 2], v[4 * ka + 3]);
      cx<T>* const drow = X + ((int64_t)f * n_symb + sy) * n_keep;
      cx<T>* const dxp = xp ? xp + ((int64_t)f * n_symb + sy) * np : nullptr;
      // ---- 2.-3. four rounds of eight kj = ka + 8 kb
      cx<T> ev[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int ka = 0; ka < 8; ++ka) {
          const int kj = ka + 8 * kb;
          t1w[72 * ka] = kj == 0 ? v[kb] : v[kb + 4 * ka] * tw32_at(kj - 1);
        }
        // the round's registers are free: the next symbol's samples of class j = kb (mod 4)
        if (nlim == N) {
#pragma unroll
          for (int m = 0; m < 8; ++m) v[kb + 4 * m] = nt_load(nsrc + 64 * (kb + 4 * m));
        } else {
#pragma unroll
          for (int m = 0; m < 8; ++m) v[kb + 4 * m] = lane + 64 * (kb + 4 * m) < nlim ? nt_load(nsrc + 64 * (kb + 4 * m)) : mk<T>(0, 0);
        }
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
        dft8_first4(u);
        // pilot rows, compact (8-byte scattered stores; collecting a symbol's pilot rows in LDS and writing them as one
        // contiguous run was measured equal to 1 % slower in three A/B pairs)
        if (dxp) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int pr = prt[64 * (4 * kb + q)];
            if (pr >= 0) dxp[pr] = u[q];
          }
        }
        // kept rows as 16-byte stores that complete 128-byte lines: an even round's four outputs wait in registers for the odd
        // round's.  Carriers k and k + 1 sit in lanes 8 c + ka and 8 (c + 1) + ka, the two halves of a 16-lane row: a lane with
        // even c sends its odd-round values and receives the partner's even-round ones (row_ror:8), and then owns the pairs
        // (k, k + 1) of the even round; the odd one the pairs (k - 1, k) of the odd round, eight carriers further on.  One
        // store instruction = for every (ka, q) the 16 consecutive carriers 16 (kb / 2) + 32 ka + 256 q .. + 15: whole lines.
        // (Measured per 4096 frames of 50 symbols: 8-byte stores in 64-byte runs 0.44 ms, 16-byte stores in 64-byte runs the
        //  same, the same bytes as contiguous 1 KB per instruction 0.17 ms.)
        if ((kb & 1) == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) ev[q] = u[q];
        } else {
          const bool odd = (lane >> 3) & 1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const cx<T> snd = odd ? ev[q] : u[q];
            const cx<T> rcv = mk<T>(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(snd.x), 0x128, 0xF, 0xF, false)),
                                    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(snd.y), 0x128, 0xF, 0xF, false)));
            const int k0 = k_lane + (odd ? 7 : 0) + 8 * (kb - 1) + 256 * q;       // even lane: k, odd lane: (k - 1) + 8
            typedef float v4 __attribute__((ext_vector_type(4)));
            v4 o;
            if (odd) { o.x = rcv.x; o.y = rcv.y; o.z = u[q].x; o.w = u[q].y; }
            else { o.x = ev[q].x; o.y = ev[q].y; o.z = rcv.x; o.w = rcv.y; }
            if (k0 + 1 < n_keep) __builtin_nontemporal_store(o, reinterpret_cast<v4*>(drow + k0));
          }
        }
      }
      src = nsrc;
      mlim = nlim;
    }
  }
}

// remove_IFO.m:5-8 of the batched receiver in ONE launch, one wavefront per frame: the segment rx_signal(Nfft+1 : 2 Nfft) taken
// from rx through the STO fix (t4_raw) and add_CFO(., -FreqOffset) (one rotor per sample, the demodulator's form), its
// 2048-point spectrum on the transform above with all eight outputs of the last radix-8, and IFO = the first bin whose
// magnitude exceeds 0.77 (:6-8) -- a minimum over the lane's 32 bins, then across the wavefront.  ifo_out[f] = bin or -1
// (no line: status -1, the script would stop at inds(1)).  Replaces segment copy -> FFT launch -> search -> finalize.
__global__ __launch_bounds__(64 * T4W_WPB) void t4_ifo_wave_kernel(const cx<float>* __restrict__ rx, const cx<float>* __restrict__ tw,
                                                                   int64_t len, int t_guard, int64_t n_frames, int time_desync,
                                                                   const int64_t* __restrict__ tg, const double* __restrict__ fo,
                                                                   double thr, int32_t* __restrict__ ifo_out,
                                                                   int32_t* __restrict__ status) {
  using T = float;
  constexpr int N = T4W_N;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lane8 = 8u * lane;
  auto tw32_at = [&](int row) { return *(const cx<T>*)(smem + T4W_OFF_TW + 512 * row + lane8); };
  auto twbl_at = [&](int row) { return *(const cx<T>*)(smem + T4W_OFF_TWB + 512 * row + lane8); };
  {
    cx<T>* const t32 = (cx<T>*)(smem + T4W_OFF_TW);
    cx<T>* const tbl = (cx<T>*)(smem + T4W_OFF_TWB);
    for (int i = threadIdx.x; i < 31 * 64; i += blockDim.x) t32[i] = tw[((i / 64 + 1) * (i & 63)) & (N - 1)];
    for (int i = threadIdx.x; i < 7 * 64; i += blockDim.x) tbl[i] = tw[(i / 64 + 1) * (i & 7) * 32];
  }
  __syncthreads();
  const unsigned wbase = T4W_OFF_WAVE + (unsigned)wave * T4W_WAVE_BYTES;
  cx<T>* const t1w = (cx<T>*)(smem + wbase + lane8);
  cx<T>* const t1r = (cx<T>*)(smem + wbase) + 72 * (lane >> 3) + (lane & 7);
  cx<T>* const t2w = (cx<T>*)(smem + wbase) + 65 * (lane & 7) + 8 * (lane >> 3);
  cx<T>* const t2r = (cx<T>*)(smem + wbase + lane8);
  const int sym_len = N + t_guard;
  const double inv = 1.0 / (double)N;
  for (int64_t f = (int64_t)blockIdx.x * T4W_WPB + wave; f < n_frames; f += (int64_t)gridDim.x * T4W_WPB) {
    const cx<T>* xf = rx + f * len;
    const int64_t pos = time_desync ? tg[f] : 0;
    const double cfo = -fo[f];
    cx<T> v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int64_t i = (int64_t)N + lane + 64 * j;                   // stream index of rx_signal(Nfft + 1 + m)
      if (!time_desync) v[j] = nt_load(xf + i);
      else {
        const int64_t i1 = i - sym_len;
        v[j] = (i1 >= 0 && i1 + pos < len) ? nt_load(xf + i1 + pos) : mk<T>(0, 0);
      }
    }
    // add_CFO(., -FreqOffset): the exactly reduced phase of the lane's first sample and of the distances 64 j (frame constants),
    // sine / cosine in float -- the form of the demodulator above (a double sincospi per sample made this kernel 66 us per
    // 4096 frames)
    {
      auto rotor = [&](int64_t i) {
        double t = cfo * (double)i * inv;
        t -= floor(t);
        float sn, cs;
        sincospif(2.0f * (float)t, &sn, &cs);
        return mk<T>(cs, sn);
      };
      const cx<T> r0 = rotor((int64_t)N + lane), step = rotor(64);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        cx<T> cur = q == 0 ? r0 : r0 * rotor(512 * q);
#pragma unroll
        for (int j = 8 * q; j < 8 * q + 8; ++j) {
          v[j] = v[j] * cur;
          if (j + 1 < 8 * q + 8) cur = cur * step;
        }
      }
    }
#pragma unroll
    for (int j0 = 0; j0 < 4; ++j0) {
      dft8<T, false>(v[j0], v[j0 + 4], v[j0 + 8], v[j0 + 12], v[j0 + 16], v[j0 + 20], v[j0 + 24], v[j0 + 28]);
      if (j0 > 0) {
#pragma unroll
        for (int ka = 1; ka < 8; ++ka) v[j0 + 4 * ka] = v[j0 + 4 * ka] * t4w_w32(j0 * ka);
      }
    }
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) dft4<T, false>(v[4 * ka], v[4 * ka + 1], v[4 * ka + 2], v[4 * ka + 3]);
    int first = 0x7fffffff;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int ka = 0; ka < 8; ++ka) {
        const int kj = ka + 8 * kb;
        t1w[72 * ka] = kj == 0 ? v[kb] : v[kb + 4 * ka] * tw32_at(kj - 1);
      }
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
      dft8<T, false>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {                                   // bin (lane >> 3) + 8 kb + 32 (lane & 7) + 256 q
        const bool above = sqrt((double)u[q].x * u[q].x + (double)u[q].y * u[q].y) > thr;
        const int k = (lane >> 3) + 8 * kb + 32 * (lane & 7) + 256 * q;
        if (above && k < first) first = k;
      }
    }
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
    if (lane == 0) {
      const int32_t r = first == 0x7fffffff ? -1 : first;
      ifo_out[f] = r;
      if (r < 0 && status[f] >= 0) status[f] = -1;
    }
  }
}

int t4_ifo_wave_launch(const void* rx, const void* tw, int64_t len, int t_guard, int64_t F, int td, const int64_t* tg,
                       const double* fo, int32_t* ifo_out, int32_t* status) {
  const unsigned grid = (unsigned)std::min<int64_t>((F + T4W_WPB - 1) / T4W_WPB, (int64_t)ctx().num_cu * 8);
  hipLaunchKernelGGL(t4_ifo_wave_kernel, dim3(grid), dim3(64 * T4W_WPB), T4W_LDS, ctx().stream, (const cx<float>*)rx, (const cx<float>*)tw,
                     len, t_guard, F, td, tg, fo, 0.77, ifo_out, status);
  return check_launch("t4_ifo_wave_kernel");
}

bool t4_demod_wave_supported(int nfft, int n_keep, int np, bool f64) {
  return !f64 && nfft == T4W_N && n_keep <= 1024 && (n_keep & 1) == 0 && !getenv("OFDM_T4_NO_WAVE");   // even: 16-byte row alignment
}

int t4_demod_wave_launch(const void* rx, void* X, const void* tw, int64_t len, int t_guard, int n_symb, int64_t F, int td, int fd,
                         const int64_t* tg, const double* fo, const int32_t* ifo, int n_keep, void* xp, const void* prole, int np) {
  // symbol runs: enough work items to give every resident wavefront (16 per CU) about five
  const int64_t resident = (int64_t)ctx().num_cu * 16;
  int spc = (int)std::min<int64_t>(n_symb, std::max<int64_t>(1, F * n_symb / (5 * resident)));
  if (const char* e = getenv("OFDM_T4_WAVE_SPC")) spc = std::max(1, std::min(n_symb, atoi(e)));
  const int chunks = (n_symb + spc - 1) / spc;
  const int64_t items = F * chunks;
  const unsigned grid = (unsigned)std::min<int64_t>((items + T4W_WPB - 1) / T4W_WPB, (int64_t)ctx().num_cu * 4);
  if (fd)
    hipLaunchKernelGGL(t4_demod_wave_kernel<true>, dim3(grid), dim3(64 * T4W_WPB), T4W_LDS, ctx().stream, (const cx<float>*)rx,
                       (cx<float>*)X, (const cx<float>*)tw, len, t_guard, n_symb, F, td, tg, fo, ifo, n_keep, (cx<float>*)xp,
                       (const int16_t*)prole, np, spc);
  else
    hipLaunchKernelGGL(t4_demod_wave_kernel<false>, dim3(grid), dim3(64 * T4W_WPB), T4W_LDS, ctx().stream, (const cx<float>*)rx,
                       (cx<float>*)X, (const cx<float>*)tw, len, t_guard, n_symb, F, td, tg, fo, ifo, n_keep, (cx<float>*)xp,
                       (const int16_t*)prole, np, spc);
  return check_launch("t4_demod_wave_kernel");
}

}  // namespace ofdm
