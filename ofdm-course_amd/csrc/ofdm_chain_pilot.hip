// Kernels 1+2 of the fast path in one launch, for comb pilot layouts (see below).  Own translation unit: the
// kernel is instantiated per (precision, Nfft/512, pruning, tap bucket) and compiles in parallel with
// ofdm_chain_fast.hip.
#include <algorithm>

#include "chain_fast_core.hpp"

namespace ofdm {

// ---------------------------------------------------------------------------------------------
// kernels 1+2 fused for comb pilots (pilotCarriers = 1 : comb : ..., Nfft/comb = M dividing 512):
//   c0(k) = sum_p Y_p exp(+2 pi i comb p k / Nfft) = sum_p Y_p exp(+2 pi i p k / M)
// is the first K outputs of a 512-point inverse transform of Y spread with stride up = 512/M, i.e. ONE
// wave-local transform instead of the K x Np dictionary correlation.
// A workgroup takes G = NW * FPW frames at a time: their symbol-1 transforms are done cooperatively (as in
// rx_pilot_kernel) with Y left in LDS, then every wavefront owns FPW of the frames: one inverse transform each
// (c0 overwrites Y in place), then their OMP iterations SIDE BY SIDE in groups of 64/FPW lanes, so the
// group-uniform solve arithmetic is paid once per wavefront, not once per frame.  No wavefront idles and
// Y / c0 never leave LDS.
// ---------------------------------------------------------------------------------------------
template <typename T, int NW, bool PRUNE2, int RT>
__global__ __launch_bounds__(64 * NW) void rx_pilot_omp_kernel(FastParams<T> P, int lg_up, int fpw, int ystride,
                                                               const cx<T>* __restrict__ rx, int64_t n_frames) {
  constexpr int N = 512 * NW;
  constexpr int NOUT = PRUNE2 ? 2 : 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* lwv = (cx<T>*)smem;                                   // [NW][WAVE_LDS_ELEMS] exchange / private
  cx<T>* const ex = lwv;
  cx<T>* twl = lwv + NW * WAVE_LDS_ELEMS;                      // [WAVE_TW_ELEMS]
  cx<T>* gl = twl + WAVE_TW_ELEMS;                             // [k_atoms] Gram table in the working precision
  cx<T>* ybuf = gl + P.k_atoms;                                // [G][ystride]  Y, later c0, of the group's frames
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gid = threadIdx.x;
  const int np = P.np, K = P.k_atoms;
  const int G = NW * fpw;
  wave_tw_fill<T, NW>(twl, P.tw);
  for (int i = gid; i < K; i += 64 * NW) gl[i] = mk<T>((T)P.gram[i].x, (T)P.gram[i].y);
  __syncthreads();
  const int64_t L = (int64_t)(N + P.t_guard) * P.n_symb;
  const int64_t n_groups = (n_frames + G - 1) / G;
  const int up_mask = (1 << lg_up) - 1;
  const int LPF = 64 / fpw, grp = lane / LPF, sl = lane - grp * LPF;
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    // Everything the transforms keep in registers (DIF / pass-B twiddles, carrier roles, the sample prefetch) is
    // (re)built per group behind an opaque zero, so that none of it stays live across the register-hungry OMP
    // phase below: 166 -> ~120 VGPRs, one more resident workgroup per CU.  The reloads hit L1 / L2.
    int opq = 0;
    asm volatile("" : "+v"(opq));
    DifTw<T, NW> dt;
    dif_tw_init<T, NW>(dt, gid + opq, P.tw);
    cx<T> twb[7];
#pragma unroll
    for (int t = 1; t < 8; ++t) twb[t - 1] = P.tw[(t * ((lane + opq) & 7) * 8) * NW];
    int kk[NOUT], pp[NOUT];
#pragma unroll
    for (int t = 0; t < NOUT; ++t) {
      kk[t] = NW * (lane + 64 * t) + wave + opq;
      pp[t] = kk[t] < P.n_carrier ? (int)P.prole[kk[t]] : -1;
    }
    cx<T> v[8], nx[8];
    frame_load<T, NW>(nx, rx + g * G * L + P.t_guard, gid, lane);
    // ---- symbol 1 of the group's frames: FFT -> stash + Y (LDS)
    for (int j = 0; j < G; ++j) {
      const int64_t f = g * G + j;
      if (f >= n_frames) break;                                // workgroup-uniform
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = nx[e];
      if (j + 1 < G && f + 1 < n_frames) frame_load<T, NW>(nx, rx + (f + 1) * L + P.t_guard, gid, lane);
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
          if (pp[t] >= 0) ybuf[j * ystride + pp[t]] = cdiv(v[t], P.pilots[pp[t]]);
        }
      }
    }
    __syncthreads();
    // ---- wavefront `wave` owns frames g*G + wave*fpw + (0..fpw-1)
    const int64_t fw0 = g * G + wave * fpw;
    if (fw0 < n_frames) {                                      // wavefront-uniform
      cx<T>* const lw = lwv + wave * WAVE_LDS_ELEMS;
      cx<T>* const yw = ybuf + wave * fpw * ystride;
      // ||Y||^2 of the lane group's frame (before c0 overwrites Y)
      cx<T>* const yg = yw + grp * ystride;
      double ynorm = 0;
      for (int p = sl; p < np; p += LPF) ynorm += (double)yg[p].x * yg[p].x + (double)yg[p].y * yg[p].y;
      for (int off = LPF >> 1; off > 0; off >>= 1) ynorm += __shfl_xor(ynorm, off, 64);
      for (int j = 0; j < fpw; ++j) {
        if (fw0 + j >= n_frames) break;
        cx<T>* const yf = yw + j * ystride;
        cx<T> c[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int i = lane + 64 * e, p = i >> lg_up;
          c[e] = ((i & up_mask) == 0 && p < np) ? conj(yf[p]) : mk<T>(0, 0);
        }
        wave_fft512<T, false>(c, lane, twb, twl, lw);
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (lane + 64 * t < K) yf[lane + 64 * t] = conj(c[t]);
      }
      wave_sync();
      omp_frame_reg<T, RT>(P, yg, gl, K, P.taps, LPF, sl, fw0 + grp < n_frames, ynorm, fw0 + grp);
    }
    __syncthreads();                                           // ybuf / private regions are reused by the next group
  }
}

template <typename T, int NW, bool PRUNE2, int RT>
static int pilot_omp_launch(const FastParams<T>& P, int lg_up, const void* rx, int64_t n_frames) {
  // frames per wavefront: as many as keep the per-group Y / c0 buffer within 8 KB (4 workgroups of ~35 KB per CU)
  const int ystride = std::max(P.np, P.k_atoms);
  int fpw = 4;
  if (const char* e = getenv("OFDM_PILOT_FPW")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) fpw = v; }
  while (fpw > 1 && sizeof(cx<T>) * (size_t)NW * fpw * ystride > 8 * 1024) fpw >>= 1;
  const size_t dyn = sizeof(cx<T>) * ((size_t)NW * WAVE_LDS_ELEMS + WAVE_TW_ELEMS + P.k_atoms + (size_t)NW * fpw * ystride);
  OFDM_ARG(dyn <= 150 * 1024, "rx_chain_task5: pilot stage needs %zu bytes of LDS", dyn);
  auto kern = rx_pilot_omp_kernel<T, NW, PRUNE2, RT>;
  const int per_cu = resident_blocks_per_cu((const void*)kern, 64 * NW, dyn);
  const int64_t groups = (n_frames + NW * fpw - 1) / (NW * fpw);
  const unsigned grid = (unsigned)std::min<int64_t>(groups, (int64_t)ctx().num_cu * per_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), dyn, ctx().stream, P, lg_up, fpw, ystride, (const cx<T>*)rx, n_frames);
  return check_launch("rx_pilot_omp_kernel");
}

template <typename T, int NW, bool PRUNE2>
static int pilot_omp_taps(const FastParams<T>& P, int lg_up, const void* rx, int64_t n_frames) {
  if (P.taps <= 2) return pilot_omp_launch<T, NW, PRUNE2, 2>(P, lg_up, rx, n_frames);
  if (P.taps <= 4) return pilot_omp_launch<T, NW, PRUNE2, 4>(P, lg_up, rx, n_frames);
  if (P.taps <= 6) return pilot_omp_launch<T, NW, PRUNE2, 6>(P, lg_up, rx, n_frames);
  return pilot_omp_launch<T, NW, PRUNE2, 8>(P, lg_up, rx, n_frames);
}

template <typename T>
int pilot_omp_run(const FastParams<T>& P, int nfft, bool prune2, int lg_up, const void* rx, int64_t n_frames) {
  OFDM_ARG(P.taps <= OMP_RT && P.k_atoms <= 512 && lg_up >= 0, "rx_chain_task5: pilot stage called outside its domain");
#define PILOT_CALL(NWV)                                                              \
  return prune2 ? pilot_omp_taps<T, NWV, true>(P, lg_up, rx, n_frames)               \
                : pilot_omp_taps<T, NWV, false>(P, lg_up, rx, n_frames)
  switch (nfft / 512) {
    case 1: PILOT_CALL(1);
    case 2: PILOT_CALL(2);
    case 4: PILOT_CALL(4);
    case 8: PILOT_CALL(8);
  }
#undef PILOT_CALL
  set_error("rx_chain_task5(pilot stage): unsupported Nfft %d", nfft);
  return OFDM_ERR_UNSUPPORTED;
}

template int pilot_omp_run<float>(const FastParams<float>&, int, bool, int, const void*, int64_t);
template int pilot_omp_run<double>(const FastParams<double>&, int, bool, int, const void*, int64_t);

}  // namespace ofdm
