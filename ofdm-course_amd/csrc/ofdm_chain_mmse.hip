// MMSE mode of the fused RX chain (BASELINE config "comb pilots, MMSE_CE + interpolate, 1M-symbol Monte-Carlo").
//
// T5/MMSE_CE.m:1-39 is, for a fixed (h, SNR, pilot layout), ONE linear operator on the pilot LS values of symbol 1:
//   H_MMSE(1..N_carrier) = Sop * (Rhp(1:Np,:) / Rpp) * H_tilde,   Rpp = rf2 + I/snr,  Rhp(1:Np,:) = rf2
//                        = Sop * (I - Rpp^-1 / snr) * H_tilde  =:  W * H_tilde
// (Sop = the not-a-knot spline + end-knot rule of interpolate.m as a real [N_carrier x Np] matrix, spline_op.hpp).
// The reference rebuilds and solves the Np x Np system for every frame; a Monte-Carlo batch shares (h, SNR), so the
// plan builds W once on the host in double and every batch is one complex GEMM  H[N_carrier x F] = W * Y[Np x F]
// -- genuinely GEMM-shaped, so fp32 runs on the matrix cores (v_mfma_f32_16x16x4_f32, exact f32) in the same
// real-GEMM form as the OMP dictionary correlation; fp64 (parity mode) uses a plain VALU kernel.
#include <algorithm>
#include <complex>
#include <vector>

#include "chain_fast_core.hpp"
#include "spline_op.hpp"

namespace ofdm {

using zc = std::complex<double>;

// W^T [np][m_pad] (row index fastest: both kernels read 16 consecutive rows per pilot), rows >= n_carrier are zero.
// Also returned: the two factors -- mt = M^T [np][np_pad] (M = I - Rpp^-1 / snr, the MMSE estimate AT the pilots) and the spline
// operator sop [n_carrier x np] (column-major) -- for the factored application of the fp32 path (mmse_factored_run below).
int mmse_build_operator(const c64* h, int64_t n_h, double snr_db, const int32_t* pilot_loc, int np, int n_carrier,
                        int m_pad, std::vector<c64>& wt, std::vector<c64>* mt_out, int np_pad, std::vector<double>* sop_out) {
  OFDM_ARG(np >= 2 && np <= 512, "rx_plan_set_mmse: 2..512 pilots supported (the operator is built on the host)");
  OFDM_ARG(n_h >= 1, "rx_plan_set_mmse: empty impulse response");
  const double snr = std::pow(10.0, snr_db * 0.1);                                 // MMSE_CE.m:13
  const double nps = (double)pilot_loc[1] - (double)pilot_loc[0];                  // :15
  double s0 = 0, s1 = 0, s2 = 0;
  for (int64_t k = 0; k < n_h; ++k) {                                              // :19-24
    const double p = h[k].x * h[k].x + h[k].y * h[k].y;
    s0 += p; s1 += p * (double)k; s2 += p * (double)k * (double)k;
  }
  OFDM_ARG(s0 > 0, "rx_plan_set_mmse: impulse response is all zero");
  const double r = s1 / s0, r2 = s2 / s0;
  const double tau_rms = std::sqrt(r2 - r * r);
  const double c = 2.0 * M_PI * tau_rms * (1.0 / (double)n_carrier) * nps;          // :25-26,:30
  // Rpp = toeplitz(1 / (1 + j c (i - j))) + I / snr  (:33-35), Hermitian positive definite
  std::vector<zc> A((size_t)np * np);
  for (int i = 0; i < np; ++i)
    for (int j = 0; j < np; ++j) {
      const double k = (double)(i - j);
      A[(size_t)i * np + j] = zc(1.0, 0.0) / zc(1.0, c * k) + (i == j ? zc(1.0 / snr, 0) : zc(0, 0));
    }
  // Cholesky A = L L^H (lower, in place)
  for (int j = 0; j < np; ++j) {
    double d = A[(size_t)j * np + j].real();
    for (int k = 0; k < j; ++k) d -= std::norm(A[(size_t)j * np + k]);
    OFDM_ARG(d > 0, "rx_plan_set_mmse: Rpp is not positive definite");
    const double ljj = std::sqrt(d);
    A[(size_t)j * np + j] = zc(ljj, 0);
    for (int i = j + 1; i < np; ++i) {
      zc s = A[(size_t)i * np + j];
      for (int k = 0; k < j; ++k) s -= A[(size_t)i * np + k] * std::conj(A[(size_t)j * np + k]);
      A[(size_t)i * np + j] = s / ljj;
    }
  }
  // X = Rpp^-1 column by column; M = I - X / snr   (rows j, columns p)
  std::vector<zc> M((size_t)np * np), col(np);
  for (int p = 0; p < np; ++p) {
    for (int i = 0; i < np; ++i) {                                                  // L z = e_p
      zc s = i == p ? zc(1, 0) : zc(0, 0);
      for (int k = (i > p ? p : i); k < i; ++k) s -= A[(size_t)i * np + k] * col[k];
      col[i] = i < p ? zc(0, 0) : s / A[(size_t)i * np + i].real();
    }
    for (int i = np - 1; i >= 0; --i) {                                             // L^H x = z
      zc s = col[i];
      for (int k = i + 1; k < np; ++k) s -= std::conj(A[(size_t)k * np + i]) * col[k];
      col[i] = s / A[(size_t)i * np + i].real();
    }
    for (int j = 0; j < np; ++j) M[(size_t)j * np + p] = (j == p ? zc(1, 0) : zc(0, 0)) - col[j] / snr;
  }
  // W = Sop * M  (:38: interpolate(H_MMSE(1:Np), pilot_loc, N_carrier, 'spline'))
  std::vector<double> sop;                                                          // [n_carrier x np], column-major
  OFDM_TRY(build_interpolate_operator(pilot_loc, np, n_carrier, 's', sop));
  if (mt_out) {
    mt_out->assign((size_t)np * np_pad, c64{0, 0});
    for (int j = 0; j < np; ++j)
      for (int p = 0; p < np; ++p) (*mt_out)[(size_t)p * np_pad + j] = c64{M[(size_t)j * np + p].real(), M[(size_t)j * np + p].imag()};
  }
  if (sop_out) *sop_out = sop;
  wt.assign((size_t)np * m_pad, c64{0, 0});
  std::vector<zc> rowacc(np);
  for (int m = 0; m < n_carrier; ++m) {
    std::fill(rowacc.begin(), rowacc.end(), zc(0, 0));
    for (int j = 0; j < np; ++j) {
      const double sj = sop[m + (size_t)j * n_carrier];
      if (sj == 0.0) continue;
      const zc* mj = &M[(size_t)j * np];
      for (int p = 0; p < np; ++p) rowacc[p] += sj * mj[p];
    }
    for (int p = 0; p < np; ++p) wt[(size_t)p * m_pad + m] = c64{rowacc[p].real(), rowacc[p].imag()};
  }
  return OFDM_OK;
}

// ---------------------------------------------------------------------------------------------
// H[f][m] = sum_p W[m][p] Y[f][p]
// fp32: real GEMM  C[M x 2F] = [Wr | Wi] * B,  B(p,re;2f) = Yr, B(p,im;2f) = -Yi, B(p,re;2f+1) = Yi, B(p,im;2f+1) = Yr.
// One wavefront = 32 rows x 64 frames (2 x 8 tiles of 16x16: 8 frames each, columns 2f / 2f+1 = Re / Im).  One k-step =
// 8 pilots = 64 MFMAs fed by 4 + 8 loads (W^T: 16 consecutive rows per pilot; Y: one 16-byte load = 2 pilots per frame
// group), both straight from L2; the next k-step's operands are requested before this step's MFMAs issue.  The round-1
// tile (16 x 32, 5 loads per 8 MFMAs) was bound by the texture path, not by the matrix pipe (38.6 % MFMA busy).
// ---------------------------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

// G = frame groups of 8 per wavefront: 8 (32 rows x 64 frames) for the tall dense operator, fewer for the small [Np x Np]
// factor, whose grid would otherwise leave one workgroup per CU
// the accumulation of one wavefront's tile: rows m0 .. m0 + 31 (two 16-row tiles) x frames f0 .. f0 + 8 G - 1
template <int G>
__device__ __forceinline__ void mmse_tile_mfma(const cx<float>* __restrict__ wt, const cx<float>* __restrict__ y, int np, int m_pad, int m0,
                                               bool two, int64_t f0, int64_t n_frames, int lane, f32x4 (&acc)[2][G]) {
  const int i16 = lane & 15, q = lane >> 4, fsub = i16 >> 1, cim = i16 & 1;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[t][g] = f32x4{0, 0, 0, 0};
  const cx<float>* ap[2] = {wt + m0 + i16, wt + (two ? m0 + 16 : m0) + i16};     // + p * m_pad
  const float4* yp[G];
  bool yv[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int64_t f = f0 + 8 * g + fsub;
    yv[g] = f < n_frames;
    yp[g] = reinterpret_cast<const float4*>(y + (yv[g] ? f : 0) * np + 2 * q);   // 16-byte aligned: np is even
  }
  // one k-step = 8 pilots: the lane holds pilots p0 + 2q and p0 + 2q + 1 (one 16-byte load per frame group), the first MFMA
  // round contracts the even ones, the second the odd ones
  cx<float> a[2][2];
  float4 b[G];
  auto fetch = [&](int p0, cx<float> (&aa)[2][2], float4 (&bb)[G]) {
    const int pa = p0 + 2 * q;
    const bool ok = pa < np;                                 // np % 4 == 0: the pair is valid or not as a whole
    const int pc = ok ? pa : 0;                              // loads are unconditional (clamped); the tail is zeroed in B
    const size_t row = (size_t)pc * m_pad;
#pragma unroll
    for (int t = 0; t < 2; ++t) { aa[t][0] = ap[t][row]; aa[t][1] = ap[t][row + m_pad]; }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 v = yp[g][(pc - 2 * q) >> 1];             // frames past the end read frame 0 and are never stored
      bb[g] = ok ? v : float4{0, 0, 0, 0};
    }
  };
  fetch(0, a, b);
  for (int p0 = 0; p0 < np; p0 += 8) {
    cx<float> an[2][2];
    float4 bn[G];
    fetch(p0 + 8 < np ? p0 + 8 : p0, an, bn);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float yr = h ? b[g].z : b[g].x, yi = h ? b[g].w : b[g].y;
        const float b_re = cim ? yi : yr;                       // multiplies Re(W)
        const float b_im = cim ? yr : -yi;                      // multiplies Im(W)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][h].x, b_re, acc[t][g], 0, 0, 0);
          acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][h].y, b_im, acc[t][g], 0, 0, 0);
        }
      }
#pragma unroll
    for (int t = 0; t < 2; ++t) { a[t][0] = an[t][0]; a[t][1] = an[t][1]; }
#pragma unroll
    for (int g = 0; g < G; ++g) b[g] = bn[g];
  }
}

template <int G>
__global__ __launch_bounds__(256) void mmse_apply_mfma_kernel(const cx<float>* __restrict__ wt, const cx<float>* __restrict__ y,
                                                              cx<float>* __restrict__ hout, int np, int m_pad, int n_carrier,
                                                              int64_t n_frames) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i16 = lane & 15, q = lane >> 4, fsub = i16 >> 1, cim = i16 & 1;
  const int m0 = (blockIdx.y * 4 + wave) * 32;
  if (m0 >= m_pad) return;                                   // wavefront-uniform
  const bool two = m0 + 16 < m_pad;                          // wavefront-uniform: second carrier tile exists
  const int64_t f0 = (int64_t)blockIdx.x * (8 * G);
  f32x4 acc[2][G];
  mmse_tile_mfma<G>(wt, y, np, m_pad, m0, two, f0, n_frames, lane, acc);
  // C: lane holds rows 4*(lane>>4)+r of column lane&15
  float* ho = reinterpret_cast<float*>(hout);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t == 1 && !two) break;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int64_t f = f0 + 8 * g + fsub;
      if (f < n_frames) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + 16 * t + 4 * q + r;
          if (m < n_carrier) ho[2 * (f * n_carrier + m) + cim] = acc[t][g][r];
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mmse_apply_valu_kernel(const cx<T>* __restrict__ wt, const cx<T>* __restrict__ y,
                                                              cx<T>* __restrict__ hout, int np, int m_pad, int n_carrier,
                                                              int64_t n_frames) {
  // one thread per (carrier, 4 frames): W^T reads coalesce over the carrier, Y values are workgroup-uniform
  const int m = blockIdx.y * 256 + threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.x * 4;
  if (m >= n_carrier) return;
  cx<T> acc[4] = {mk<T>(0, 0), mk<T>(0, 0), mk<T>(0, 0), mk<T>(0, 0)};
  for (int p = 0; p < np; ++p) {
    const cx<T> w = wt[(size_t)p * m_pad + m];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (f0 + j < n_frames) acc[j] = acc[j] + w * y[(f0 + j) * np + p];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (f0 + j < n_frames) hout[(f0 + j) * n_carrier + m] = acc[j];
}

// ---------------------------------------------------------------------------------------------
// Factored application (fp32 plans): H = Sop * (M * Y).  The dense W = Sop * M [N_carrier x Np] costs N_carrier / Np times the
// flops of M [Np x Np] (4x at comb 4), and the spline operator of interpolate.m is banded for every practical purpose: the
// weight of a not-a-knot spline falls by (2 - sqrt 3) per knot, so each row keeps the `bw` columns around its interval whose
// weights reach 1e-10 of the row's largest (33 columns at 256 knots) -- below half an fp32 ulp of the result.
//   (1) v[f][:] = M * Y[f][:]     mmse_apply_mfma_kernel on M^T [np][np_pad]          (8 np^2 flop per frame)
//   (2) H[f][m] = sum_t w[t][m] v[f][c0[m] + t]     spline_band_kernel: v tile and weights in LDS, a quad of rows x 8 frames per thread
// fp64 plans keep the dense operator on VALU (parity mode: unchanged results).
// ---------------------------------------------------------------------------------------------
constexpr int SB_FT = 32;                                     // frames per workgroup (8 per wavefront)
constexpr int SB_QUADS = 64;                                  // quads of rows per workgroup = lanes: 256 carriers
constexpr int SB_VS = SB_FT + 2;                              // v tile row stride: 272 B, so the windows a wavefront's lanes start in spread over
                                                              // the 16-byte bank groups (256 B apart they were a 16-way conflict: 147 us)
// One thread = one QUAD of neighbouring rows (4q .. 4q+3: at comb 4 one knot interval, their windows coincide) x 8 frames; one
// workgroup = 64 quads x 32 frames (wavefront w takes frames 8w .. 8w+7).  Per tap a thread reads 16 B of weights and 64 B of v
// from LDS for 64 FMAs (1.25 B per FMA).  The first form -- one row x 32 frames per thread, 4 B of LDS per FMA -- was bound by
// the LDS: 67 us per 8192 frames of C4.  The weights of a row outside its own window [c0, c0 + bw) are zero, fmaf(0, x, acc) == acc:
// every row sums the same products in the same order as before.
__global__ __launch_bounds__(256) void spline_band_kernel(const float4* __restrict__ w4, const int32_t* __restrict__ c4, int bw, int span,
                                                          const cx<float>* __restrict__ v, cx<float>* __restrict__ hout, int nc,
                                                          int np, int64_t n_frames) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sb_smem[];
  cx<float>* const vs = (cx<float>*)sb_smem;                  // [span][SB_VS]
  float4* const ws = (float4*)(vs + (size_t)span * SB_VS);    // [bw][SB_QUADS]
  const int nq = (nc + 3) >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = blockIdx.x * SB_QUADS, q = q0 + lane;
  const int64_t f0 = (int64_t)blockIdx.y * SB_FT;
  const int c_lo = c4[q0];                                    // c4 is non-decreasing in q
  // every load of the two tiles is in flight before the first LDS write (a loop of dependent round trips was 3/4 of the kernel's time):
  // wavefront w takes frames w, w + 4, ... (two coalesced runs of 64 pilots each) and the taps w, w + 4, ...
  for (int jb = 0; jb < span; jb += 128) {
    cx<float> tv[8][2];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int f = wave + 4 * k, j = jb + lane + 64 * h;
        const bool ok = j < span && f0 + f < n_frames && c_lo + j < np;
        tv[k][h] = ok ? v[(f0 + f) * np + c_lo + j] : mk<float>(0, 0);
      }
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = jb + lane + 64 * h;
        if (j < span) vs[j * SB_VS + wave + 4 * k] = tv[k][h];
      }
  }
  for (int tb = 0; tb < bw; tb += 48) {
    float4 tw[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int t = tb + wave + 4 * k;
      tw[k] = t < bw && q < nq ? w4[(size_t)t * nq + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int t = tb + wave + 4 * k;
      if (t < bw) ws[t * SB_QUADS + lane] = tw[k];
    }
  }
  __syncthreads();
  if (q >= nq) return;
  const int j0 = c4[q] - c_lo;
  cx<float> acc[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[r][f] = mk<float>(0, 0);
  for (int t = 0; t < bw; ++t) {
    const float4 w = ws[t * SB_QUADS + lane];
    const float wr[4] = {w.x, w.y, w.z, w.w};
    const float4* row = reinterpret_cast<const float4*>(vs + (j0 + t) * SB_VS + 8 * wave);
#pragma unroll
    for (int f2 = 0; f2 < 4; ++f2) {
      const float4 x = row[f2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[r][2 * f2].x = fmaf(wr[r], x.x, acc[r][2 * f2].x);         acc[r][2 * f2].y = fmaf(wr[r], x.y, acc[r][2 * f2].y);
        acc[r][2 * f2 + 1].x = fmaf(wr[r], x.z, acc[r][2 * f2 + 1].x); acc[r][2 * f2 + 1].y = fmaf(wr[r], x.w, acc[r][2 * f2 + 1].y);
      }
    }
  }
  const bool wide = (nc & 1) == 0 && 4 * q + 3 < nc;          // 16-byte stores need an even row length
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const int64_t fr = f0 + 8 * wave + f;
    if (fr >= n_frames) break;
    cx<float>* o = hout + fr * nc + 4 * q;
    if (wide) {
      reinterpret_cast<float4*>(o)[0] = make_float4(acc[0][f].x, acc[0][f].y, acc[1][f].x, acc[1][f].y);
      reinterpret_cast<float4*>(o)[1] = make_float4(acc[2][f].x, acc[2][f].y, acc[3][f].x, acc[3][f].y);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * q + r < nc) o[r] = acc[r][f];
    }
  }
}

// Both factors in ONE launch (Np <= 256): a 512-thread workgroup owns 32 frames.  Phase 1: wavefront w accumulates rows 32 w .. 32 w + 31
// of v = M * Y for the 32 frames (mmse_tile_mfma<4>: the eight wavefronts read the same Y rows -- L1 hits -- and M^T once per
// workgroup) and writes them to the v tile in LDS [Np][SB_VS].  Phase 2: the banded spline from that tile, a quad of rows x 8 frames
// per thread as in spline_band_kernel, the weights straight from L2 (one batch of four taps ahead).  Against the two launches:
// no v round trip, no tile / weight load phase, 147 MB instead of 268 MB through the L2 -> L1 path.  Bit-identical results.
template <int FT>
__global__ __launch_bounds__(512) void mmse_fused_kernel(const cx<float>* __restrict__ mt, const cx<float>* __restrict__ y,
                                                         const float4* __restrict__ w4, const int32_t* __restrict__ c4, int bw,
                                                         cx<float>* __restrict__ hout, int np, int np_pad, int nc, int64_t n_frames) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mf_smem[];
  constexpr int G = FT / 8, VS = FT + 2;
  cx<float>* const vs = (cx<float>*)mf_smem;                  // [np_pad][VS]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t f0 = (int64_t)blockIdx.x * FT;
  {
    const int m0 = wave * 32;
    if (m0 < np_pad) {                                        // wavefront-uniform
      const int i16 = lane & 15, q = lane >> 4, fsub = i16 >> 1, cim = i16 & 1;
      const bool two = m0 + 16 < np_pad;
      f32x4 acc[2][G];
      mmse_tile_mfma<G>(mt, y, np, np_pad, m0, two, f0, n_frames, lane, acc);
      float* vf = reinterpret_cast<float*>(vs);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two) break;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) vf[2 * ((m0 + 16 * t + 4 * q + r) * VS + 8 * g + fsub) + cim] = acc[t][g][r];
      }
    }
  }
  __syncthreads();
  const int nq = (nc + 3) >> 2;
  const int n_task = ((nq + 63) >> 6) * G;                    // (64 quads) x (8 frames)
  for (int task = wave; task < n_task; task += 8) {
    const int q = (task / G) * 64 + lane, fg = task % G;
    const bool qok = q < nq;
    const int qc = qok ? q : nq - 1;
    const int j0 = c4[qc];
    cx<float> acc[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int f = 0; f < 8; ++f) acc[r][f] = mk<float>(0, 0);
    float4 wn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wn[k] = k < bw ? w4[(size_t)k * nq + qc] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < bw; t0 += 4) {
      float4 wc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) wc[k] = wn[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) wn[k] = t0 + 4 + k < bw ? w4[(size_t)(t0 + 4 + k) * nq + qc] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (t0 + k >= bw) break;                              // uniform
        const float wr[4] = {wc[k].x, wc[k].y, wc[k].z, wc[k].w};
        const float4* row = reinterpret_cast<const float4*>(vs + (j0 + t0 + k) * VS + 8 * fg);
#pragma unroll
        for (int f2 = 0; f2 < 4; ++f2) {
          const float4 x = row[f2];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[r][2 * f2].x = fmaf(wr[r], x.x, acc[r][2 * f2].x);         acc[r][2 * f2].y = fmaf(wr[r], x.y, acc[r][2 * f2].y);
            acc[r][2 * f2 + 1].x = fmaf(wr[r], x.z, acc[r][2 * f2 + 1].x); acc[r][2 * f2 + 1].y = fmaf(wr[r], x.w, acc[r][2 * f2 + 1].y);
          }
        }
      }
    }
    if (!qok) continue;
    const bool wide = (nc & 1) == 0 && 4 * q + 3 < nc;
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int64_t fr = f0 + 8 * fg + f;
      if (fr >= n_frames) break;
      cx<float>* o = hout + fr * nc + 4 * q;
      if (wide) {
        reinterpret_cast<float4*>(o)[0] = make_float4(acc[0][f].x, acc[0][f].y, acc[1][f].x, acc[1][f].y);
        reinterpret_cast<float4*>(o)[1] = make_float4(acc[2][f].x, acc[2][f].y, acc[3][f].x, acc[3][f].y);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * q + r < nc) o[r] = acc[r][f];
      }
    }
  }
}

// rows of the spline operator cut to their band, per quad of rows: w [bw][nq][4] (t-major), c0 [nq] = first column of the quad's
// window, nq = ceil(nc / 4).  span = the largest number of columns the 64 quads of one workgroup of spline_band_kernel touch
void mmse_band_spline(const std::vector<double>& sop, int nc, int np, std::vector<float>& w, std::vector<int32_t>& c0, int& bw, int& span) {
  // columns [lo, hi] of a row that reach 1e-10 of its largest weight (a row AT a knot is a unit vector: lo = hi = the knot)
  std::vector<int> lo(nc), hi(nc);
  for (int m = 0; m < nc; ++m) {
    double mx = 0;
    for (int j = 0; j < np; ++j) mx = std::max(mx, std::fabs(sop[m + (size_t)j * nc]));
    int a = np, b = -1;
    for (int j = 0; j < np; ++j)
      if (std::fabs(sop[m + (size_t)j * nc]) > 1e-10 * mx) { a = std::min(a, j); b = std::max(b, j); }
    if (b < 0) { a = 0; b = 0; }
    lo[m] = a; hi[m] = b;
  }
  // a ROW's window: first column = the smallest lo of this and every later row (non-decreasing in m, never to the right of the
  // row's own first column), width rbw = the largest hi - first + 1 of any row
  std::vector<int> r0(nc);
  int smin = np;
  for (int m = nc - 1; m >= 0; --m) { smin = std::min(smin, lo[m]); r0[m] = smin; }
  int rbw = 1;
  for (int m = 0; m < nc; ++m) rbw = std::max(rbw, hi[m] - r0[m] + 1);
  rbw = std::min(np, rbw);
  for (int m = 0; m < nc; ++m) r0[m] = std::max(0, std::min(r0[m], np - rbw));
  // a QUAD's window: from its first row's first column to its last row's last column
  const int nq = (nc + 3) / 4;
  bw = rbw;
  for (int q = 0; q < nq; ++q) bw = std::max(bw, r0[std::min(nc - 1, 4 * q + 3)] + rbw - r0[4 * q]);
  bw = std::min(np, bw);
  c0.resize(nq);
  w.assign((size_t)bw * nq * 4, 0.f);
  for (int q = 0; q < nq; ++q) {
    const int base = std::max(0, std::min(r0[4 * q], np - bw));
    c0[q] = base;
    for (int t = 0; t < bw; ++t)
      for (int r = 0; r < 4 && 4 * q + r < nc; ++r) {
        const int m = 4 * q + r, col = base + t;
        if (col >= r0[m] && col < r0[m] + rbw) w[((size_t)t * nq + q) * 4 + r] = (float)sop[m + (size_t)col * nc];
      }
  }
  span = bw;
  for (int q0 = 0; q0 < nq; q0 += SB_QUADS) {
    const int last = std::min(nq, q0 + SB_QUADS) - 1;
    span = std::max(span, c0[last] + bw - c0[q0]);
  }
}

// the banded product on its own (the Task-4 receiver's estimate_channel.m:8 on fp32 plans): H[f][m] = sum_t w[t][m] v[f][c0[m] + t]
int spline_band_run(const float* sb_w, const int32_t* sb_c0, int bw, int span, const void* v, void* hout, int np, int n_carrier,
                    int64_t n_frames) {
  const int nq = (n_carrier + 3) / 4;
  const dim3 g2((unsigned)((nq + SB_QUADS - 1) / SB_QUADS), (unsigned)((n_frames + SB_FT - 1) / SB_FT));
  const size_t lds = sizeof(cx<float>) * (size_t)span * SB_VS + sizeof(float4) * (size_t)bw * SB_QUADS;
  if (lds > 150 * 1024) return 1;                              // not taken: the caller keeps its dense product
  (void)hipFuncSetAttribute((const void*)spline_band_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(spline_band_kernel, g2, dim3(256), lds, ctx().stream, (const float4*)sb_w, sb_c0, bw, span, (const cx<float>*)v,
                     (cx<float>*)hout, n_carrier, np, n_frames);
  return check_launch("spline_band_kernel");
}

bool mmse_factored_usable(int np, int np_pad) { return np % 4 == 0 && np_pad % 16 == 0 && !getenv("OFDM_MMSE_NO_MFMA") && !getenv("OFDM_MMSE_DENSE"); }

int mmse_factored_run(const void* mt, int np_pad, const float* sb_w, const int32_t* sb_c0, int bw, int span, const void* y, void* v,
                      void* hout, int np, int n_carrier, int64_t n_frames) {
  hipStream_t st = ctx().stream;
  if (np_pad <= 256 && hout && !getenv("OFDM_MMSE_TWO_LAUNCHES")) {   // both factors in one launch
    constexpr int ft = 32;                                     // (16 frames per workgroup, two workgroups per CU: 0.096 against 0.086 ms)
    const size_t lds = sizeof(cx<float>) * (size_t)np_pad * (ft + 2);
    (void)hipFuncSetAttribute((const void*)mmse_fused_kernel<ft>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mmse_fused_kernel<ft>, dim3((unsigned)((n_frames + ft - 1) / ft)), dim3(512), lds, st, (const cx<float>*)mt,
                       (const cx<float>*)y, (const float4*)sb_w, sb_c0, bw, (cx<float>*)hout, np, np_pad, n_carrier, n_frames);
    return check_launch("mmse_fused_kernel");
  }
  int G = 4;                                                  // 32 rows x 32 frames per wavefront (G = 8 / 4 / 2 / 1 measured: 146 / 129 / 129 / 141 us per 8192 frames)
  if (const char* e = getenv("OFDM_MMSE_G")) G = atoi(e);
  const unsigned gy = (unsigned)(((np_pad + 31) / 32 + 3) / 4);
  auto launch = [&](auto kern, int g) {
    hipLaunchKernelGGL(kern, dim3((unsigned)((n_frames + 8 * g - 1) / (8 * g)), gy), dim3(256), 0, st, (const cx<float>*)mt,
                       (const cx<float>*)y, (cx<float>*)v, np, np_pad, np, n_frames);
  };
  if (G >= 8) launch(mmse_apply_mfma_kernel<8>, 8);
  else if (G >= 4) launch(mmse_apply_mfma_kernel<4>, 4);
  else if (G >= 2) launch(mmse_apply_mfma_kernel<2>, 2);
  else launch(mmse_apply_mfma_kernel<1>, 1);
  OFDM_TRY(check_launch("mmse_apply_mfma_kernel"));
  const int rc = spline_band_run(sb_w, sb_c0, bw, span, v, hout, np, n_carrier, n_frames);
  OFDM_ARG(rc <= 0, "rx_chain_task5 (MMSE mode): the spline band does not fit the LDS");
  return rc;
}

template <typename T>
int mmse_apply_run(const void* wt, const void* y, void* hout, int np, int m_pad, int n_carrier, int64_t n_frames) {
  hipStream_t st = ctx().stream;
  if constexpr (std::is_same<T, float>::value) {
    if (np % 4 == 0 && m_pad % 16 == 0 && !getenv("OFDM_MMSE_NO_MFMA")) {
      const dim3 grid((unsigned)((n_frames + 63) / 64), (unsigned)(((m_pad + 31) / 32 + 3) / 4));
      hipLaunchKernelGGL(mmse_apply_mfma_kernel<8>, grid, dim3(256), 0, st, (const cx<float>*)wt, (const cx<float>*)y,
                         (cx<float>*)hout, np, m_pad, n_carrier, n_frames);
      return check_launch("mmse_apply_mfma_kernel");
    }
  }
  const dim3 grid((unsigned)((n_frames + 3) / 4), (unsigned)((n_carrier + 255) / 256));
  hipLaunchKernelGGL(mmse_apply_valu_kernel<T>, grid, dim3(256), 0, st, (const cx<T>*)wt, (const cx<T>*)y, (cx<T>*)hout, np,
                     m_pad, n_carrier, n_frames);
  return check_launch("mmse_apply_valu_kernel");
}

template int mmse_apply_run<float>(const void*, const void*, void*, int, int, int, int64_t);
template int mmse_apply_run<double>(const void*, const void*, void*, int, int, int, int64_t);

}  // namespace ofdm
