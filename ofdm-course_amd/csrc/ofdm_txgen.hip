// ofdm_tx_frames: synthetic RX frames of an RX plan's geometry, generated entirely on the device,
// so that Monte-Carlo sweeps never touch host memory.  Per frame, in the TX + channel call order of
// T5/Main_model_Task_5.m:50-127 / T5/Task5_part2.m:96-134:
//
//   payload   one Philox4x32-10 draw per QAM symbol j (counter (j_lo, j_hi, stream, 1), key = seed, stream = frame0 + f):
//             code = top bps bits of word 0; its bits MSB-first are bits j*bps .. j*bps+bps-1 of the frame (mapping.m:15-18)
//   mapping + OFDM_map_carriers   X(dataCarriers, s) = dict(code + 1), X(pilotCarriers, s) = the plan's pilot column
//   OFDM_modulator -> conv(x, h) truncated per frame -> Noise(SNR) with Philox stream frame0 + f (domain tag 0)
//
// The payload draw is an INPUT of every parity test (restated in oracle/ofdm_oracle.py:payload_codes_philox); the
// stages after it are the library's own kernels (mod_kernel, fir_kernel, awgn_kernel).
#include <algorithm>

#include "rx_plan.hpp"

namespace ofdm {
int mod_device(const void* x, void* y, int nfft, int64_t n_symb, int t_guard, bool f64);     // ofdm_modem.hip

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

// X [nfft x n_symb * n_frames]: one thread per carrier of a symbol (coalesced), roles from the plan's tables
template <typename T>
__global__ __launch_bounds__(256) void tx_fill_kernel(cx<T>* __restrict__ X, const int16_t* __restrict__ prole,
                                                      const int16_t* __restrict__ drole, const cx<T>* __restrict__ pilots,
                                                      const cx<T>* __restrict__ dict, int nfft, int n_symb, int nd, int bps,
                                                      uint32_t k0, uint32_t k1, uint32_t stream0, int64_t n_frames) {
  const int64_t total = (int64_t)nfft * n_symb * n_frames;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % nfft);
    const int64_t col = i / nfft;
    const int s = (int)(col % n_symb);
    const int64_t f = col / n_symb;
    cx<T> v = mk<T>(0, 0);
    const int p = prole[k], d = drole[k];
    if (p >= 0) v = pilots[p];                                    // pilot rows written last in OFDM_map_carriers.m:8
    else if (d >= 0) v = dict[payload_code((uint64_t)s * nd + d, stream0 + (uint32_t)f, k0, k1, bps)];
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

extern "C" int ofdm_tx_frames(ofdm_rx_plan* pl, const void* h, int h_len, double snr_db, int noise_on, uint64_t seed,
                              int64_t frame0, int64_t n_frames, void* rx_out, uint8_t* ref_bits_out, uint8_t* bits_out,
                              int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(pl && n_frames >= 0 && rx_out, "tx_frames: bad arguments");
  OFDM_ARG((is_f64(flags) ? 1 : 0) == pl->f64, "tx_frames: precision flag differs from the plan's");
  OFDM_ARG(pl->pilots_in_band, "tx_frames: pilots outside 1..N_carrier are not supported");
  OFDM_ARG(frame0 >= 0 && frame0 + n_frames < ((int64_t)1 << 32), "tx_frames: frame index outside the 32-bit stream range");
  if (n_frames == 0) return OFDM_OK;
  const size_t cs = csize(flags);
  const int64_t frame_samples = (int64_t)(pl->nfft + pl->t_guard) * pl->n_symb;
  const size_t fb = (size_t)pl->frame_words * 4;
  const int64_t frame_bits = (int64_t)pl->nd * pl->n_symb * pl->bps;
  Stage st(flags);
  void *drx, *dref, *dbits;
  OFDM_TRY(st.out(rx_out, cs * (size_t)frame_samples * n_frames, &drx));
  OFDM_TRY(st.out(ref_bits_out, fb * n_frames, &dref));
  OFDM_TRY(st.out(bits_out, (size_t)frame_bits * n_frames, &dbits));
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
  // chunks of frames through a plan-owned scratch: X [nfft x S] and the guarded time signal [(nfft+tg) x S]
  const int64_t CH = std::min<int64_t>(n_frames, 1024);
  const size_t x_bytes = cs * (size_t)pl->nfft * pl->n_symb * CH;
  const size_t t_bytes = cs * (size_t)frame_samples * CH;
  const size_t need = x_bytes + 2 * t_bytes;
  if (pl->ws_gen_bytes < need) {
    OFDM_HIP(hipStreamSynchronize(ctx().stream));
    if (pl->ws_gen) { (void)hipFree(pl->ws_gen); pl->ws_gen = nullptr; }
    OFDM_HIP(hipMalloc(&pl->ws_gen, need));
    pl->ws_gen_bytes = need;
  }
  unsigned char* base = (unsigned char*)pl->ws_gen;
  void* dX = base;
  void* dT = base + x_bytes;
  void* dY = base + x_bytes + t_bytes;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  hipStream_t s = ctx().stream;
  const int devflags = (flags & ~OFDM_DEVICE) | OFDM_DEVICE;
  for (int64_t c0 = 0; c0 < n_frames; c0 += CH) {
    const int64_t nf = std::min<int64_t>(CH, n_frames - c0);
    const uint32_t stream0 = (uint32_t)(frame0 + c0);
    const int64_t total = (int64_t)pl->nfft * pl->n_symb * nf;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)ctx().num_cu * 16);
    if (pl->f64)
      hipLaunchKernelGGL(tx_fill_kernel<double>, dim3(grid), dim3(256), 0, s, (c64*)dX, (const int16_t*)pl->d_prole,
                         (const int16_t*)pl->d_drole, (const c64*)pl->d_pilots, (const c64*)pl->d_dict, pl->nfft, pl->n_symb,
                         pl->nd, pl->bps, k0, k1, stream0, nf);
    else
      hipLaunchKernelGGL(tx_fill_kernel<float>, dim3(grid), dim3(256), 0, s, (c32*)dX, (const int16_t*)pl->d_prole,
                         (const int16_t*)pl->d_drole, (const c32*)pl->d_pilots, (const c32*)pl->d_dict, pl->nfft, pl->n_symb,
                         pl->nd, pl->bps, k0, k1, stream0, nf);
    OFDM_TRY(check_launch("tx_fill_kernel"));
    if (dref || dbits) {
      const int64_t groups = (int64_t)(((int64_t)pl->nd * pl->n_symb + 31) >> 5) * nf;
      const unsigned g2 = (unsigned)std::min<int64_t>((groups + 255) / 256, (int64_t)ctx().num_cu * 16);
      hipLaunchKernelGGL(tx_bits_kernel, dim3(g2), dim3(256), 0, s, dref ? (uint32_t*)((uint8_t*)dref + fb * c0) : nullptr,
                         dbits ? (uint8_t*)dbits + frame_bits * c0 : nullptr, pl->n_symb, pl->nd, pl->bps, pl->frame_words, k0,
                         k1, stream0, nf);
      OFDM_TRY(check_launch("tx_bits_kernel"));
    }
    OFDM_TRY(mod_device(dX, dT, pl->nfft, (int64_t)pl->n_symb * nf, pl->t_guard, pl->f64 != 0));
    void* dst = (unsigned char*)drx + cs * (size_t)frame_samples * c0;
    const void* chan_out = dT;
    if (h && h_len > 0) {
      OFDM_TRY(ofdm_channel_conv_frames(dT, frame_samples, nf, h, h_len, noise_on ? dY : dst, devflags));
      chan_out = noise_on ? dY : dst;
    }
    if (noise_on) OFDM_TRY(ofdm_Noise_frames(snr_db, chan_out, frame_samples, nf, seed, stream0, dst, devflags));
    else if (chan_out == dT) OFDM_HIP(hipMemcpyAsync(dst, dT, cs * (size_t)frame_samples * nf, hipMemcpyDeviceToDevice, s));
  }
  return st.finish();
}
