// Channel side: get_MP_channel_resp + the inline conv of the drivers, Noise (counter-based AWGN),
// add_STO, add_CFO.
#include <algorithm>
#include <type_traits>

#include "ofdm_common.hpp"

namespace ofdm {

int demod_device(const void* y, void* x, int nfft, int64_t n_symb, int t_guard, bool f64);   // ofdm_modem.hip

static unsigned ew_grid(int64_t total, int per_block = 256) {
  int64_t b = (total + per_block - 1) / per_block;
  int64_t cap = (int64_t)ctx().num_cu * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------
// conv(x,h.','full')(1:L) -- T5/Main_model_Task_5.m:126-127 as a sparse-tap gather FIR:
// y[n] = sum_t a_t * x[n - d_t].  Zero taps of the dense h are skipped (exact: adds of +0).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct TapList {
  const int32_t* delay;
  const cx<T>* amp;
  int n;
};

// frame_len > 0: the stream is a batch of independent frames (each starts from silence).
template <typename T>
__global__ void fir_kernel(const cx<T>* __restrict__ x, cx<T>* __restrict__ y, int64_t len, int64_t frame_len,
                           TapList<T> taps) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < len; n += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in_frame = n % frame_len;
    cx<T> acc = mk<T>(0, 0);
    for (int t = 0; t < taps.n; ++t) {
      const int64_t d = taps.delay[t];
      if (in_frame - d >= 0) acc = acc + x[n - d] * taps.amp[t];
    }
    y[n] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Noise -- T5/Noise.m:1-12.  Pass 1: sum |x|^2 (per-block partials in double, fixed order).
// Pass 2 (one block): P -> sigma = sqrt(P / 10^(snr/10) / 2).  Pass 3: y = x + sigma*(n_re + i n_im)
// with Philox4x32-10(counter = (i_lo, i_hi, stream, 0), key = seed) + Box-Muller on words 0,1.
// ---------------------------------------------------------------------------------------------
// grid = (blocks_per_frame, n_frames); partial[frame * blocks_per_frame + block]
template <typename T>
__global__ void power_partial_kernel(const cx<T>* __restrict__ xall, int64_t len, double* __restrict__ partial) {
  const cx<T>* x = xall + (int64_t)blockIdx.y * len;
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    const cx<T> v = x[i];
    s += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ double ws[16];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += ws[w];
    partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

// out[0] = sqrt(NoisePower/2) (per-component sigma), out[1] = sqrt(NoisePower) (N_var of Noise.m:11).
// One wavefront per frame (a single thread walking 2048 partial sums cost more than the noise pass it feeds).
// snr_lin_v (optional): the frame's own 10^(SNR/10) -- a sweep whose frames are its SNR points (T5/Main_model_Task_5.m:305-307)
__global__ __launch_bounds__(64) void noise_sigma_kernel(const double* __restrict__ partial, int n_part, int64_t len, double snr_lin,
                                                         double* __restrict__ out, int64_t n_frames,
                                                         const double* __restrict__ snr_lin_v = nullptr) {
  const int64_t f = blockIdx.x;
  double s = 0;
  for (int i = threadIdx.x; i < n_part; i += 64) s += partial[f * n_part + i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (threadIdx.x == 0) {
    const double p = s / (double)len;             // Noise.m:3
    const double np = p / (snr_lin_v ? snr_lin_v[f] : snr_lin);   // :5
    out[2 * f] = sqrt(np / 2.0);
    out[2 * f + 1] = sqrt(np);
  }
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&r)[4]) {
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

// grid.y = frame; frame f uses Philox stream `stream0 + f` and its own sigma
template <typename T>
__global__ void awgn_kernel(const cx<T>* __restrict__ xall, cx<T>* __restrict__ yall, int64_t len,
                            const double* __restrict__ sigma, uint32_t k0, uint32_t k1, uint32_t stream0) {
  const cx<T>* x = xall + (int64_t)blockIdx.y * len;
  cx<T>* y = yall + (int64_t)blockIdx.y * len;
  const double sg = sigma[2 * blockIdx.y];
  const uint32_t stream = stream0 + blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), stream, 0u, k0, k1, r);
    // Box-Muller on the same Philox words in both precisions (the draw is an INPUT of the chain): double arithmetic in
    // parity mode; in fp32 mode the uniform words are still formed in double, log / sincospi / sqrt run in float --
    // the same realisation to ~1e-7 of a noise sample, at twice the rate (the double form made Noise compute-bound)
    const double u0 = ((double)r[0] + 0.5) * 2.3283064365386963e-10;
    const double u1 = ((double)r[1] + 0.5) * 2.3283064365386963e-10;
    const cx<T> v = nt_load(x + i);                 // last reader of x (the power pass was the first)
    if constexpr (std::is_same<T, float>::value) {
      const float rad = sqrtf(-2.0f * logf((float)u0)) * (float)sg;
      float sn, cs;
      sincospif((float)(2.0 * u1), &sn, &cs);
      nt_store(y + i, mk<T>(v.x + rad * cs, v.y + rad * sn));
    } else {
      const double rad = sqrt(-2.0 * log(u0));
      double sn, cs;
      sincospi(2.0 * u1, &sn, &cs);
      nt_store(y + i, mk<T>((T)((double)v.x + sg * rad * cs), (T)((double)v.y + sg * rad * sn)));
    }
  }
}

// add_STO -- T5/add_STO.m:1-10
template <typename T>
__global__ void sto_kernel(const cx<T>* __restrict__ y, cx<T>* __restrict__ out, int64_t len, int64_t n_sto) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t src = i + n_sto;
    out[i] = (src >= 0 && src < len) ? y[src] : mk<T>(0, 0);
  }
}

// add_CFO -- T5/add_CFO.m:1-8: y .* exp(2j*pi*CFO*n/Nfft).  The turn count CFO*n/Nfft is formed
// in double and reduced to its fractional part before the sincos, for any stream length.
template <typename T>
__global__ void cfo_kernel(const cx<T>* __restrict__ y, cx<T>* __restrict__ out, int64_t len, double cfo,
                           double inv_nfft) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    const double t = cfo * (double)i * inv_nfft;
    const double fr = t - floor(t);
    double sn, cs;
    sincospi(2.0 * fr, &sn, &cs);
    const cx<T> v = y[i];
    out[i] = mk<T>((T)((double)v.x * cs - (double)v.y * sn), (T)((double)v.x * sn + (double)v.y * cs));
  }
}

// add_STO then add_CFO of a batch of frames, every frame with its own draw (T4/Main_model_Task_4.m:101-110 per Monte-Carlo
// run): out[f][i] = (i + sto_f < len ? y[f][i + sto_f] : 0) * exp(2j*pi*cfo_f*i/Nfft) -- add_STO.m:5-9 (either sign) followed
// by add_CFO.m:6-7 on the shifted stream, the rotation with the arithmetic of cfo_kernel.  sto == nullptr / cfo == nullptr
// leave the stage out.  grid.y = frame.
template <typename T>
__global__ void sto_cfo_frames_kernel(const cx<T>* __restrict__ yall, cx<T>* __restrict__ outall, int64_t len,
                                      const int64_t* __restrict__ sto, const double* __restrict__ cfo, double inv_nfft) {
  const cx<T>* y = yall + (int64_t)blockIdx.y * len;
  cx<T>* out = outall + (int64_t)blockIdx.y * len;
  const int64_t n_sto = sto ? sto[blockIdx.y] : 0;
  const double f = cfo ? cfo[blockIdx.y] : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t src = i + n_sto;
    cx<T> v = (src >= 0 && src < len) ? y[src] : mk<T>(0, 0);
    if (cfo) {
      const double t = f * (double)i * inv_nfft;
      const double fr = t - floor(t);
      double sn, cs;
      sincospi(2.0 * fr, &sn, &cs);
      v = mk<T>((T)((double)v.x * cs - (double)v.y * sn), (T)((double)v.x * sn + (double)v.y * cs));
    }
    out[i] = v;
  }
}

int sto_cfo_frames_device(const void* y, void* out, int64_t len, int64_t n_frames, const int64_t* d_sto, const double* d_cfo,
                          int nfft, bool f64) {
  if (len == 0 || n_frames == 0) return OFDM_OK;
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((len + 255) / 256, 64));
  for (int64_t f0 = 0; f0 < n_frames; f0 += 65535) {
    const int64_t nf = std::min<int64_t>(65535, n_frames - f0);
    const dim3 grid(gx, (unsigned)nf);
    if (f64)
      hipLaunchKernelGGL(sto_cfo_frames_kernel<double>, grid, dim3(256), 0, ctx().stream, (const c64*)y + f0 * len,
                         (c64*)out + f0 * len, len, d_sto ? d_sto + f0 : nullptr, d_cfo ? d_cfo + f0 : nullptr, 1.0 / (double)nfft);
    else
      hipLaunchKernelGGL(sto_cfo_frames_kernel<float>, grid, dim3(256), 0, ctx().stream, (const c32*)y + f0 * len,
                         (c32*)out + f0 * len, len, d_sto ? d_sto + f0 : nullptr, d_cfo ? d_cfo + f0 : nullptr, 1.0 / (double)nfft);
  }
  return check_launch("sto_cfo_frames_kernel");
}

// device-pointer helpers reused by the sync code
int cfo_device(const void* y, void* out, int64_t len, double cfo, int nfft, bool f64) {
  if (len == 0) return OFDM_OK;
  if (f64)
    hipLaunchKernelGGL(cfo_kernel<double>, dim3(ew_grid(len)), dim3(256), 0, ctx().stream, (const c64*)y, (c64*)out,
                       len, cfo, 1.0 / (double)nfft);
  else
    hipLaunchKernelGGL(cfo_kernel<float>, dim3(ew_grid(len)), dim3(256), 0, ctx().stream, (const c32*)y, (c32*)out,
                       len, cfo, 1.0 / (double)nfft);
  return check_launch("cfo_kernel");
}

}  // namespace ofdm

using namespace ofdm;

extern "C" {

int ofdm_get_MP_channel_resp(const double* taps, const double* taps_im, int n_taps, int nfft, void* h_out,
                             int* h_len_out, void* H_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(taps && n_taps > 0, "get_MP_channel_resp: empty tap list");
  int max_delay = 0;
  for (int i = 0; i < n_taps; ++i) {
    const double d = taps[i];                                   // column 1 = delays (column-major)
    OFDM_ARG(d >= 0 && d == std::floor(d) && d < (1 << 24), "get_MP_channel_resp: delay %g is not a non-negative integer", d);
    if ((int)d > max_delay) max_delay = (int)d;                 // :4
  }
  const int hl = max_delay + 1;                                 // :5
  std::vector<c64> h(hl, c64{0, 0});
  for (int i = 0; i < n_taps; ++i)                              // :11-15 (later duplicates overwrite)
    h[(int)taps[i]] = c64{taps[n_taps + i], taps_im ? taps_im[i] : 0.0};
  if (h_len_out) *h_len_out = hl;
  if (h_out) {
    if (is_f64(flags)) memcpy(h_out, h.data(), sizeof(c64) * hl);
    else for (int i = 0; i < hl; ++i) ((c32*)h_out)[i] = c32{(float)h[i].x, (float)h[i].y};
  }
  if (H_out) {
    // fft(h, Nfft): zero-pad / truncate to Nfft (:18), on the device FFT
    const bool f64 = is_f64(flags);
    const size_t cs = f64 ? sizeof(c64) : sizeof(c32);
    std::vector<c64> hp(nfft, c64{0, 0});
    for (int i = 0; i < hl && i < nfft; ++i) hp[i] = h[i];
    std::vector<c32> hp32;
    const void* src = hp.data();
    if (!f64) {
      hp32.resize(nfft);
      for (int i = 0; i < nfft; ++i) hp32[i] = c32{(float)hp[i].x, (float)hp[i].y};
      src = hp32.data();
    }
    Stage st(OFDM_HOST | (flags & OFDM_F64));
    const void* din; void* dout;
    OFDM_TRY(st.in(src, cs * nfft, &din));
    OFDM_TRY(st.out(H_out, cs * nfft, &dout));
    OFDM_TRY(demod_device(din, dout, nfft, 1, 0, f64));
    OFDM_TRY(st.finish());
  }
  return OFDM_OK;
}

static int conv_common(const void* x, int64_t frame_len, int64_t n_frames, const void* h, int h_len, void* y, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(frame_len >= 0 && n_frames >= 0 && h_len > 0 && h, "channel_conv: bad sizes");
  const int64_t len = frame_len * n_frames;
  const bool f64 = is_f64(flags);
  std::vector<int32_t> delays;
  std::vector<c64> a64;
  std::vector<c32> a32;
  for (int d = 0; d < h_len; ++d) {
    double re = f64 ? ((const c64*)h)[d].x : ((const c32*)h)[d].x;
    double im = f64 ? ((const c64*)h)[d].y : ((const c32*)h)[d].y;
    if (re != 0.0 || im != 0.0) {
      delays.push_back(d);
      a64.push_back(c64{re, im});
      a32.push_back(c32{(float)re, (float)im});
    }
  }
  Stage st(flags);
  const void *dx, *dd, *da; void* dy;
  OFDM_TRY(st.in(x, csize(flags) * (size_t)len, &dx));
  OFDM_TRY(st.out(y, csize(flags) * (size_t)len, &dy));
  const int nt = (int)delays.size();
  if (nt == 0) {
    if (len) OFDM_HIP(hipMemsetAsync(dy, 0, csize(flags) * (size_t)len, ctx().stream));
    return st.finish();
  }
  OFDM_TRY(st.upload(delays.data(), sizeof(int32_t) * nt, &dd));
  OFDM_TRY(st.upload(f64 ? (const void*)a64.data() : (const void*)a32.data(), csize(flags) * nt, &da));
  if (len > 0) {
    if (f64) {
      TapList<double> tl{(const int32_t*)dd, (const c64*)da, nt};
      hipLaunchKernelGGL(fir_kernel<double>, dim3(ew_grid(len)), dim3(256), 0, ctx().stream, (const c64*)dx,
                         (c64*)dy, len, frame_len, tl);
    } else {
      TapList<float> tl{(const int32_t*)dd, (const c32*)da, nt};
      hipLaunchKernelGGL(fir_kernel<float>, dim3(ew_grid(len)), dim3(256), 0, ctx().stream, (const c32*)dx,
                         (c32*)dy, len, frame_len, tl);
    }
    OFDM_TRY(check_launch("fir_kernel"));
  }
  return st.finish();
}

int ofdm_channel_conv(const void* x, int64_t len, const void* h, int h_len, void* y, int flags) {
  return conv_common(x, len, 1, h, h_len, y, flags);
}
int ofdm_channel_conv_frames(const void* x, int64_t frame_len, int64_t n_frames, const void* h, int h_len, void* y,
                             int flags) {
  return conv_common(x, frame_len, n_frames, h, h_len, y, flags);
}

static int noise_common(double snr_db, const void* x, int64_t len, int64_t n_frames, uint64_t seed, uint32_t stream,
                        void* y, double* n_var_out, int flags, const double* snr_db_v = nullptr) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(len >= 0 && n_frames >= 0 && n_frames < 65536 * 1024, "Noise: bad sizes");
  if (len == 0 || n_frames == 0) { if (n_var_out) *n_var_out = NAN; return OFDM_OK; }
  const bool f64 = is_f64(flags);
  Stage st(flags);
  const void* dx; void *dy, *dpart, *dsig;
  OFDM_TRY(st.in(x, csize(flags) * (size_t)len * n_frames, &dx));
  OFDM_TRY(st.out(y, csize(flags) * (size_t)len * n_frames, &dy));
  unsigned bpf = (unsigned)std::min<int64_t>((len + 2047) / 2048, n_frames > 1 ? 16 : 2048);   // blocks per frame
  if (bpf < 1) bpf = 1;
  OFDM_ARG(n_frames <= 65535, "Noise: at most 65535 frames per call");
  OFDM_TRY(st.scratch(sizeof(double) * (size_t)bpf * n_frames, &dpart));
  double sig_host[2] = {0, 0};
  const bool want = n_var_out && n_frames == 1;
  if (want) OFDM_TRY(st.fetch(sig_host, sizeof(sig_host), &dsig));
  else OFDM_TRY(st.scratch(sizeof(double) * 2 * (size_t)n_frames, &dsig));
  const double snr_lin = std::pow(10.0, snr_db / 10.0);
  const void* dsnr = nullptr;
  std::vector<double> lin;
  if (snr_db_v) {
    lin.resize((size_t)n_frames);
    for (int64_t f = 0; f < n_frames; ++f) lin[f] = std::pow(10.0, snr_db_v[f] / 10.0);
    OFDM_TRY(st.upload(lin.data(), sizeof(double) * lin.size(), &dsnr));
  }
  const dim3 pgrid(bpf, (unsigned)n_frames);
  if (f64) hipLaunchKernelGGL(power_partial_kernel<double>, pgrid, dim3(256), 0, ctx().stream, (const c64*)dx, len, (double*)dpart);
  else hipLaunchKernelGGL(power_partial_kernel<float>, pgrid, dim3(256), 0, ctx().stream, (const c32*)dx, len, (double*)dpart);
  OFDM_TRY(check_launch("power_partial_kernel"));
  hipLaunchKernelGGL(noise_sigma_kernel, dim3((unsigned)n_frames), dim3(64), 0, ctx().stream, (const double*)dpart,
                     (int)bpf, len, snr_lin, (double*)dsig, n_frames, (const double*)dsnr);
  OFDM_TRY(check_launch("noise_sigma_kernel"));
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  unsigned abx = (unsigned)std::min<int64_t>((len + 255) / 256, n_frames > 1 ? 64 : (int64_t)ctx().num_cu * 8);
  if (abx < 1) abx = 1;
  const dim3 agrid(abx, (unsigned)n_frames);
  if (f64) hipLaunchKernelGGL(awgn_kernel<double>, agrid, dim3(256), 0, ctx().stream, (const c64*)dx, (c64*)dy, len, (const double*)dsig, k0, k1, stream);
  else hipLaunchKernelGGL(awgn_kernel<float>, agrid, dim3(256), 0, ctx().stream, (const c32*)dx, (c32*)dy, len, (const double*)dsig, k0, k1, stream);
  OFDM_TRY(check_launch("awgn_kernel"));
  OFDM_TRY(st.finish());
  if (want) *n_var_out = sig_host[1];
  return OFDM_OK;
}

int ofdm_Noise(double snr_db, const void* x, int64_t len, uint64_t seed, uint32_t stream, void* y,
               double* n_var_out, int flags) {
  return noise_common(snr_db, x, len, 1, seed, stream, y, n_var_out, flags);
}
int ofdm_Noise_frames(double snr_db, const void* x, int64_t frame_len, int64_t n_frames, uint64_t seed,
                      uint32_t stream0, void* y, int flags) {
  return noise_common(snr_db, x, frame_len, n_frames, seed, stream0, y, nullptr, flags);
}

int ofdm_Noise_frames_snr(const double* snr_db, const void* x, int64_t frame_len, int64_t n_frames, uint64_t seed,
                          uint32_t stream0, void* y, int flags) {
  OFDM_ARG(snr_db, "Noise_frames_snr: null SNR array");
  return noise_common(0.0, x, frame_len, n_frames, seed, stream0, y, nullptr, flags, snr_db);
}

int ofdm_add_STO(const void* y, int64_t len, int64_t n_sto, void* out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(len >= 0, "add_STO: negative length");
  Stage st(flags);
  const void* dy; void* dout;
  OFDM_TRY(st.in(y, csize(flags) * (size_t)len, &dy));
  OFDM_TRY(st.out(out, csize(flags) * (size_t)len, &dout));
  if (len > 0) {
    if (is_f64(flags)) hipLaunchKernelGGL(sto_kernel<double>, dim3(ew_grid(len)), dim3(256), 0, ctx().stream, (const c64*)dy, (c64*)dout, len, n_sto);
    else hipLaunchKernelGGL(sto_kernel<float>, dim3(ew_grid(len)), dim3(256), 0, ctx().stream, (const c32*)dy, (c32*)dout, len, n_sto);
    OFDM_TRY(check_launch("sto_kernel"));
  }
  return st.finish();
}

int ofdm_add_CFO(const void* y, int64_t len, double cfo, int nfft, void* out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(len >= 0 && nfft > 0, "add_CFO: bad sizes");
  Stage st(flags);
  const void* dy; void* dout;
  OFDM_TRY(st.in(y, csize(flags) * (size_t)len, &dy));
  OFDM_TRY(st.out(out, csize(flags) * (size_t)len, &dout));
  OFDM_TRY(cfo_device(dy, dout, len, cfo, nfft, is_f64(flags)));
  return st.finish();
}

int ofdm_add_STO_CFO_frames(const void* y, int64_t frame_len, int64_t n_frames, const int64_t* n_sto, const double* cfo, int nfft,
                            void* out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(frame_len >= 0 && n_frames >= 0 && nfft > 0, "add_STO_CFO_frames: bad sizes");
  Stage st(flags);
  const void *dy, *dsto = nullptr, *dcfo = nullptr; void* dout;
  OFDM_TRY(st.in(y, csize(flags) * (size_t)frame_len * n_frames, &dy));
  if (n_sto) OFDM_TRY(st.in(n_sto, sizeof(int64_t) * (size_t)n_frames, &dsto));
  if (cfo) OFDM_TRY(st.in(cfo, sizeof(double) * (size_t)n_frames, &dcfo));
  OFDM_TRY(st.out(out, csize(flags) * (size_t)frame_len * n_frames, &dout));
  OFDM_TRY(sto_cfo_frames_device(dy, dout, frame_len, n_frames, (const int64_t*)dsto, (const double*)dcfo, nfft, is_f64(flags)));
  return st.finish();
}

}  // extern "C"
