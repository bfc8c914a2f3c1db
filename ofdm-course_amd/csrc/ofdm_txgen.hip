// ofdm_tx_frames / ofdm_tx_frames_ex: synthetic RX frames of an RX plan's geometry, generated entirely on the device
// (SURVEY.md 8f-1), so that Monte-Carlo sweeps never touch host memory.  Per frame:
//
//   payload   one Philox4x32-10 draw per QAM symbol j (counter (j_lo, j_hi, stream, 1), key = seed, stream = frame0 + f):
//             code = top bps bits of word 0; its bits MSB-first are bits j*bps .. j*bps+bps-1 of the frame (mapping.m:15-18)
//   Scrambler (optional) with the register reset for every frame (T5/Main_model_Task_5.m:55-69)
//   mapping + OFDM_map_carriers   X(dataCarriers, s) = dict(code + 1), X(pilotCarriers, s) = the plan's pilot column
//   OFDM_modulator, then the channel stages.  The REFERENCE order (T5/Main_model_Task_5.m:106-127, T4/Main_model_Task_4.m:
//   94-110,:257-267, T5/Task5_part2.m:134,:152) is  Noise -> add_STO -> add_CFO -> conv(h)  = `noise_first` != 0;
//   ofdm_tx_frames keeps the order of rounds 1-2, conv(h) -> Noise (noise_first = 0: add_STO -> add_CFO -> conv -> Noise).
//   STO / CFO per frame (T4:101-110): fixed, or drawn per frame from Philox counter (0, 0, stream, 2):
//             Time_Delay = word0 mod (Nfft + T_Guard + 1)            [randi([0, Nfft + T_Guard])]
//             Freq_Shift = (word1 mod 31) + ((word2 + 0.5) 2^-32 - 0.5)   [randi([0, 30]) + (rand - 0.5)]
//
// The payload / impairment draws are INPUTS of every parity test (restated in oracle/ofdm_oracle.py:payload_codes_philox,
// :sto_cfo_draw_philox); the stages after them are the library's own kernels (scramble_kernel, mod_kernel, fir_kernel,
// awgn_kernel, sto_cfo_frames_kernel), each tested against the oracle on its own and composed in tests/test_gpu_txgen.py.
#include <algorithm>

#include "rx_plan.hpp"

namespace ofdm {
int mod_device(const void* x, void* y, int nfft, int64_t n_symb, int t_guard, bool f64);     // ofdm_modem.hip
int sto_cfo_frames_device(const void* y, void* out, int64_t len, int64_t n_frames, const int64_t* d_sto, const double* d_cfo,
                          int nfft, bool f64);                                              // ofdm_channel.hip

__device__ __forceinline__ uint32_t payload_code(uint64_t j, uint32_t stream, uint32_t k0, uint32_t k1, int bps) {
  uint32_t c0 = (uint32_t)j, c1 = (uint32_t)(j >> 32), c2 = stream, c3 = 1u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0 >> (32 - bps);
}

__device__ __forceinline__ void philox_words(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                             uint32_t (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// per-frame impairment draws of T4/Main_model_Task_4.m:101-110 (mode 1: the fixed value, mode 2: drawn)
__global__ __launch_bounds__(256) void tx_draw_kernel(int64_t* __restrict__ sto, double* __restrict__ cfo, int sto_mode,
                                                      int64_t sto_value, int cfo_mode, double cfo_value, uint32_t sto_span,
                                                      uint32_t k0, uint32_t k1, uint32_t stream0, int64_t n_frames) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= n_frames) return;
  uint32_t r[4];
  philox_words(0u, 0u, stream0 + (uint32_t)f, 2u, k0, k1, r);
  if (sto) sto[f] = sto_mode == 2 ? (int64_t)(r[0] % sto_span) : sto_value;
  if (cfo) cfo[f] = cfo_mode == 2 ? (double)(r[1] % 31u) + (((double)r[2] + 0.5) * 2.3283064365386963e-10 - 0.5) : cfo_value;
}

// bytes (one per bit, [n_frames][frame_bits]) -> packed words of the chain's layout
__global__ __launch_bounds__(256) void tx_pack_bits_kernel(const uint8_t* __restrict__ bits, uint32_t* __restrict__ packed,
                                                           int64_t frame_bits, int frame_words, int64_t n_frames) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)frame_words * n_frames; i += (int64_t)gridDim.x * 256) {
    const int64_t f = i / frame_words;
    const int64_t b0 = (i - f * frame_words) * 32;
    uint32_t w = 0;
    for (int b = 0; b < 32; ++b) w = (w << 1) | ((b0 + b < frame_bits) ? (uint32_t)(bits[f * frame_bits + b0 + b] & 1u) : 0u);
    packed[i] = __builtin_bswap32(w);
  }
}

// X [nfft x n_symb * n_frames]: one thread per carrier of a symbol (coalesced), roles from the plan's tables
template <typename T>
__global__ __launch_bounds__(256) void tx_fill_kernel(cx<T>* __restrict__ X, const int16_t* __restrict__ prole,
                                                      const int16_t* __restrict__ drole, const cx<T>* __restrict__ pilots,
                                                      const cx<T>* __restrict__ dict, int nfft, int n_symb, int nd, int bps,
                                                      uint32_t k0, uint32_t k1, uint32_t stream0, int64_t n_frames,
                                                      const uint8_t* __restrict__ sc_bits) {
  const int64_t total = (int64_t)nfft * n_symb * n_frames;
  const int64_t frame_bits = (int64_t)nd * n_symb * bps;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % nfft);
    const int64_t col = i / nfft;
    const int s = (int)(col % n_symb);
    const int64_t f = col / n_symb;
    cx<T> v = mk<T>(0, 0);
    const int p = prole[k], d = drole[k];
    if (p >= 0) v = pilots[p];                                    // pilot rows written last in OFDM_map_carriers.m:8
    else if (d >= 0) {
      uint32_t code;
      if (sc_bits) {                                               // the frame's scrambled bits, bps per symbol, MSB first
        const uint8_t* b = sc_bits + f * frame_bits + ((int64_t)s * nd + d) * bps;
        code = 0;
        for (int q = 0; q < bps; ++q) code = (code << 1) | (uint32_t)(b[q] & 1u);
      } else {
        code = payload_code((uint64_t)s * nd + d, stream0 + (uint32_t)f, k0, k1, bps);
      }
      v = dict[code];
    }
    X[i] = v;
  }
}

// packed reference bits (layout of ofdm_rx_chain_task5) and, optionally, one byte per bit: 32 symbols -> bps words
__global__ __launch_bounds__(256) void tx_bits_kernel(uint32_t* __restrict__ packed, uint8_t* __restrict__ bits, int n_symb,
                                                      int nd, int bps, int frame_words, uint32_t k0, uint32_t k1,
                                                      uint32_t stream0, int64_t n_frames) {
  const int n_codes = nd * n_symb;
  const int n_groups = (n_codes + 31) >> 5;
  const int64_t frame_bits = (int64_t)n_codes * bps;
  for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < (int64_t)n_groups * n_frames; gi += (int64_t)gridDim.x * 256) {
    const int64_t f = gi / n_groups;
    const int grp = (int)(gi - f * n_groups);
    unsigned long long acc = 0;
    int nb = 0, w = grp * bps;
    for (int i = 0; i < 32; ++i) {
      const int j = 32 * grp + i;
      const uint32_t code = j < n_codes ? payload_code((uint64_t)j, stream0 + (uint32_t)f, k0, k1, bps) : 0u;
      if (bits && j < n_codes)
        for (int b = 0; b < bps; ++b) bits[f * frame_bits + (int64_t)j * bps + b] = (uint8_t)((code >> (bps - 1 - b)) & 1u);
      acc = (acc << bps) | code;
      nb += bps;
      if (nb >= 32) {
        nb -= 32;
        if (packed && w < frame_words) packed[f * frame_words + w] = __builtin_bswap32((uint32_t)(acc >> nb));
        ++w;
      }
    }
  }
}

}  // namespace ofdm

using namespace ofdm;

extern "C" int ofdm_tx_frames_ex(ofdm_rx_plan* pl, const void* h, int h_len, double snr_db, int noise_on, uint64_t seed,
                                 int64_t frame0, int64_t n_frames, const uint8_t* scr_reg15, int sto_mode, int64_t sto_value,
                                 int cfo_mode, double cfo_value, int noise_first, void* rx_out, uint8_t* ref_bits_out,
                                 uint8_t* bits_out, uint8_t* sc_ref_bits_out, int64_t* sto_out, double* cfo_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(pl && n_frames >= 0 && rx_out, "tx_frames: bad arguments");
  OFDM_ARG(pl->nd >= 1, "tx_frames: the plan has no data carriers");
  OFDM_PLAN_DEVICE(pl);
  OFDM_ARG((is_f64(flags) ? 1 : 0) == pl->f64, "tx_frames: precision flag differs from the plan's");
  OFDM_ARG(pl->pilots_in_band, "tx_frames: pilots outside 1..N_carrier are not supported");
  OFDM_ARG(frame0 >= 0 && frame0 + n_frames < ((int64_t)1 << 32), "tx_frames: frame index outside the 32-bit stream range");
  OFDM_ARG(sto_mode >= 0 && sto_mode <= 2 && cfo_mode >= 0 && cfo_mode <= 2, "tx_frames: sto_mode / cfo_mode must be 0, 1 or 2");
  OFDM_ARG(scr_reg15 || !sc_ref_bits_out, "tx_frames: sc_ref_bits_out needs the Scrambler register");
  if (n_frames == 0) return OFDM_OK;
  const size_t cs = csize(flags);
  const int64_t frame_samples = (int64_t)(pl->nfft + pl->t_guard) * pl->n_symb;
  const size_t fb = (size_t)pl->frame_words * 4;
  const int64_t frame_bits = (int64_t)pl->nd * pl->n_symb * pl->bps;
  Stage st(flags);
  void *drx, *dref, *dbits, *dscref, *dsto_o, *dcfo_o;
  OFDM_TRY(st.out(rx_out, cs * (size_t)frame_samples * n_frames, &drx));
  OFDM_TRY(st.out(ref_bits_out, fb * n_frames, &dref));
  OFDM_TRY(st.out(bits_out, (size_t)frame_bits * n_frames, &dbits));
  OFDM_TRY(st.out(sc_ref_bits_out, fb * n_frames, &dscref));
  OFDM_TRY(st.out(sto_out, sizeof(int64_t) * (size_t)n_frames, &dsto_o));
  OFDM_TRY(st.out(cfo_out, sizeof(double) * (size_t)n_frames, &dcfo_o));
  if (!pl->d_dict) {                                              // constellation table in the plan's precision
    const size_t n = pl->dict.size();
    OFDM_HIP(hipMalloc(&pl->d_dict, cs * n));
    if (pl->f64) {
      OFDM_HIP(hipMemcpy(pl->d_dict, pl->dict.data(), sizeof(c64) * n, hipMemcpyHostToDevice));
    } else {
      std::vector<c32> d32(n);
      for (size_t i = 0; i < n; ++i) d32[i] = c32{(float)pl->dict[i].x, (float)pl->dict[i].y};
      OFDM_HIP(hipMemcpy(pl->d_dict, d32.data(), sizeof(c32) * n, hipMemcpyHostToDevice));
    }
  }
  // chunks of frames through a plan-owned scratch: X [nfft x S], two guarded time signals [(nfft+tg) x S], and with the
  // Scrambler on the chunk's payload and scrambled bits (one byte each), plus the per-frame draws
  const int64_t CH = std::min<int64_t>(n_frames, 1024);
  const size_t x_bytes = cs * (size_t)pl->nfft * pl->n_symb * CH;
  const size_t t_bytes = cs * (size_t)frame_samples * CH;
  const size_t b_bytes = scr_reg15 ? (((size_t)frame_bits * CH + 255) & ~size_t(255)) : 0;
  const size_t d_bytes = 16 * (size_t)CH;
  const size_t need = x_bytes + 2 * t_bytes + 2 * b_bytes + d_bytes;
  if (pl->ws_gen_bytes < need) {
    OFDM_HIP(hipStreamSynchronize(ctx().stream));
    if (pl->ws_gen) { (void)hipFree(pl->ws_gen); pl->ws_gen = nullptr; }
    OFDM_HIP(hipMalloc(&pl->ws_gen, need));
    pl->ws_gen_bytes = need;
  }
  unsigned char* base = (unsigned char*)pl->ws_gen;
  void* dX = base;
  void* buf[2] = {base + x_bytes, base + x_bytes + t_bytes};
  uint8_t* dB0 = base + x_bytes + 2 * t_bytes;                     // payload bits of the chunk
  uint8_t* dB1 = dB0 + b_bytes;                                    // scrambled bits of the chunk
  int64_t* dsto = (int64_t*)(dB1 + b_bytes);
  double* dcfo = (double*)(dsto + CH);
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  hipStream_t s = ctx().stream;
  const int devflags = (flags & ~OFDM_DEVICE) | OFDM_DEVICE;
  const bool chan = h && h_len > 0;
  const bool imp = sto_mode != 0 || cfo_mode != 0;
  for (int64_t c0 = 0; c0 < n_frames; c0 += CH) {
    const int64_t nf = std::min<int64_t>(CH, n_frames - c0);
    const uint32_t stream0 = (uint32_t)(frame0 + c0);
    const int64_t total = (int64_t)pl->nfft * pl->n_symb * nf;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)ctx().num_cu * 16);
    const int64_t groups = (int64_t)(((int64_t)pl->nd * pl->n_symb + 31) >> 5) * nf;
    const unsigned g2 = (unsigned)std::min<int64_t>((groups + 255) / 256, (int64_t)ctx().num_cu * 16);
    uint8_t* chunk_bits = dbits ? (uint8_t*)dbits + frame_bits * c0 : (scr_reg15 ? dB0 : nullptr);
    if (dref || chunk_bits) {
      hipLaunchKernelGGL(tx_bits_kernel, dim3(g2), dim3(256), 0, s, dref ? (uint32_t*)((uint8_t*)dref + fb * c0) : nullptr,
                         chunk_bits, pl->n_symb, pl->nd, pl->bps, pl->frame_words, k0, k1, stream0, nf);
      OFDM_TRY(check_launch("tx_bits_kernel"));
    }
    if (scr_reg15) {                                              // Scrambler.m per frame, register reset (T5:58-69)
      OFDM_TRY(ofdm_Scrambler_frames(scr_reg15, chunk_bits, frame_bits, nf, dB1, devflags));
      if (dscref) {
        const int64_t words = (int64_t)pl->frame_words * nf;
        hipLaunchKernelGGL(tx_pack_bits_kernel, dim3((unsigned)std::min<int64_t>((words + 255) / 256, (int64_t)ctx().num_cu * 16)),
                           dim3(256), 0, s, dB1, (uint32_t*)((uint8_t*)dscref + fb * c0), frame_bits, pl->frame_words, nf);
        OFDM_TRY(check_launch("tx_pack_bits_kernel"));
      }
    }
    if (pl->f64)
      hipLaunchKernelGGL(tx_fill_kernel<double>, dim3(grid), dim3(256), 0, s, (c64*)dX, (const int16_t*)pl->d_prole,
                         (const int16_t*)pl->d_drole, (const c64*)pl->d_pilots, (const c64*)pl->d_dict, pl->nfft, pl->n_symb,
                         pl->nd, pl->bps, k0, k1, stream0, nf, scr_reg15 ? dB1 : nullptr);
    else
      hipLaunchKernelGGL(tx_fill_kernel<float>, dim3(grid), dim3(256), 0, s, (c32*)dX, (const int16_t*)pl->d_prole,
                         (const int16_t*)pl->d_drole, (const c32*)pl->d_pilots, (const c32*)pl->d_dict, pl->nfft, pl->n_symb,
                         pl->nd, pl->bps, k0, k1, stream0, nf, scr_reg15 ? dB1 : nullptr);
    OFDM_TRY(check_launch("tx_fill_kernel"));
    OFDM_TRY(mod_device(dX, buf[0], pl->nfft, (int64_t)pl->n_symb * nf, pl->t_guard, pl->f64 != 0));
    if (imp) {
      hipLaunchKernelGGL(tx_draw_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, sto_mode ? dsto : nullptr,
                         cfo_mode ? dcfo : nullptr, sto_mode, sto_value, cfo_mode, cfo_value,
                         (uint32_t)(pl->nfft + pl->t_guard + 1), k0, k1, stream0, nf);
      OFDM_TRY(check_launch("tx_draw_kernel"));
      if (dsto_o) {
        if (sto_mode) OFDM_HIP(hipMemcpyAsync((int64_t*)dsto_o + c0, dsto, sizeof(int64_t) * nf, hipMemcpyDeviceToDevice, s));
        else OFDM_HIP(hipMemsetAsync((int64_t*)dsto_o + c0, 0, sizeof(int64_t) * nf, s));
      }
      if (dcfo_o) {
        if (cfo_mode) OFDM_HIP(hipMemcpyAsync((double*)dcfo_o + c0, dcfo, sizeof(double) * nf, hipMemcpyDeviceToDevice, s));
        else OFDM_HIP(hipMemsetAsync((double*)dcfo_o + c0, 0, sizeof(double) * nf, s));
      }
    } else {
      if (dsto_o) OFDM_HIP(hipMemsetAsync((int64_t*)dsto_o + c0, 0, sizeof(int64_t) * nf, s));
      if (dcfo_o) OFDM_HIP(hipMemsetAsync((double*)dcfo_o + c0, 0, sizeof(double) * nf, s));
    }
    // channel stages in the requested order; every stage reads `cur` and writes the other scratch buffer, the last one `dst`
    void* dst = (unsigned char*)drx + cs * (size_t)frame_samples * c0;
    enum { ST_NOISE, ST_IMP, ST_CONV };
    int order[3], n_st = 0;
    if (noise_first) { if (noise_on) order[n_st++] = ST_NOISE; if (imp) order[n_st++] = ST_IMP; if (chan) order[n_st++] = ST_CONV; }
    else { if (imp) order[n_st++] = ST_IMP; if (chan) order[n_st++] = ST_CONV; if (noise_on) order[n_st++] = ST_NOISE; }
    int cur = 0;
    for (int q = 0; q < n_st; ++q) {
      void* to = q + 1 == n_st ? dst : buf[1 - cur];
      if (order[q] == ST_NOISE) OFDM_TRY(ofdm_Noise_frames(snr_db, buf[cur], frame_samples, nf, seed, stream0, to, devflags));
      else if (order[q] == ST_IMP)
        OFDM_TRY(sto_cfo_frames_device(buf[cur], to, frame_samples, nf, sto_mode ? dsto : nullptr, cfo_mode ? dcfo : nullptr,
                                       pl->nfft, pl->f64 != 0));
      else OFDM_TRY(ofdm_channel_conv_frames(buf[cur], frame_samples, nf, h, h_len, to, devflags));
      cur = 1 - cur;
    }
    if (n_st == 0) OFDM_HIP(hipMemcpyAsync(dst, buf[0], cs * (size_t)frame_samples * nf, hipMemcpyDeviceToDevice, s));
  }
  return st.finish();
}

extern "C" int ofdm_tx_frames(ofdm_rx_plan* pl, const void* h, int h_len, double snr_db, int noise_on, uint64_t seed,
                              int64_t frame0, int64_t n_frames, void* rx_out, uint8_t* ref_bits_out, uint8_t* bits_out,
                              int flags) {
  return ofdm_tx_frames_ex(pl, h, h_len, snr_db, noise_on, seed, frame0, n_frames, nullptr, 0, 0, 0, 0.0, 0, rx_out, ref_bits_out,
                           bits_out, nullptr, nullptr, nullptr, flags);
}
