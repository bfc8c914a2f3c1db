// Device building blocks shared by the fast-path kernels of the fused Task-5 RX chain
// (ofdm_chain_fast.hip: rx_pilot / omp_batch / rx_symbols; ofdm_chain_pilot.hip: rx_pilot_omp).
#pragma once
#include <cstdlib>
#include <map>
#include <utility>
#include <vector>
#include <type_traits>

#include "demap_core.hpp"
#include "fft_core.hpp"

namespace ofdm {

constexpr int FAST_MAXT = 32;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int WAVE_LDS_ELEMS = 512 + 64;

// Twiddles of the two inner passes of the wave-local 512-point transform, kept in LDS as
// [pass][t-1][lane] (lane-contiguous: conflict-free ds_read_b64, no VALU, no live registers):
//   pass B: W_64^(t*(lane&7))      pass C: W_512^(t*lane)        t = 1..7
// 2 * 7 * 64 complex values, shared by every wavefront of the workgroup.
constexpr int WAVE_TW_ELEMS = 2 * 7 * 64;

// tw = exp(-2 pi i m / N) table of the full transform, N = 512 * NW.  Call with all threads; the
// caller synchronises the workgroup before the first transform.
template <typename T, int NW>
__device__ __forceinline__ void wave_tw_fill(cx<T>* __restrict__ twl, const cx<T>* __restrict__ tw) {
  for (int i = threadIdx.x; i < WAVE_TW_ELEMS; i += 64 * NW) {
    const int pass = i / 448, r = i - pass * 448, t = r / 64 + 1, l = r & 63;
    twl[i] = pass == 0 ? tw[(t * (l & 7) * 8) * NW] : tw[(t * l) * NW];
  }
}

// X0 and X1 of an 8-point DFT (forward): v0 <- sum v_t ; v1 <- sum v_t W8^t
template <typename T>
__device__ __forceinline__ void dft8_first2(cx<T> (&v)[8]) {
  const T h = T(0.70710678118654752440084436210485);
  const cx<T> a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
  const cx<T> b0 = v[0] - v[4];
  cx<T> b1 = v[1] - v[5], b2 = v[2] - v[6], b3 = v[3] - v[7];
  b1 = mk<T>((b1.x + b1.y) * h, (b1.y - b1.x) * h);
  b2 = mk<T>(b2.y, -b2.x);
  b3 = mk<T>((b3.y - b3.x) * h, (-b3.x - b3.y) * h);
  v[0] = (a0 + a2) + (a1 + a3);
  v[1] = (b0 + b2) + (b1 + b3);
}

// Eight ds_read_b64 at base + e*STRIDE elements, issued back to back and waited for once.
// Hand-issued because the compiler fuses neighbouring b64 reads into ds_read2_b64 / ds_read2st64_b64,
// which move half the bytes per LDS cycle (MI355X_MICROARCH.md, LDS table: 128 vs 256 B/clk).
// One asm statement: the outputs only become valid at its trailing s_waitcnt.
template <int STRIDE, bool ASM>
__device__ __forceinline__ void lds_read8(cx<float> (&v)[8], const cx<float>* base) {
  if constexpr (ASM) {
    const unsigned a = (unsigned)(uintptr_t)base;            // LDS aperture: low 32 bits = LDS byte address
    unsigned long long r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile(
        "ds_read_b64 %0, %8 offset:%9\n\t"
        "ds_read_b64 %1, %8 offset:%10\n\t"
        "ds_read_b64 %2, %8 offset:%11\n\t"
        "ds_read_b64 %3, %8 offset:%12\n\t"
        "ds_read_b64 %4, %8 offset:%13\n\t"
        "ds_read_b64 %5, %8 offset:%14\n\t"
        "ds_read_b64 %6, %8 offset:%15\n\t"
        "ds_read_b64 %7, %8 offset:%16\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(a), "n"(0 * STRIDE * 8), "n"(1 * STRIDE * 8), "n"(2 * STRIDE * 8), "n"(3 * STRIDE * 8),
          "n"(4 * STRIDE * 8), "n"(5 * STRIDE * 8), "n"(6 * STRIDE * 8), "n"(7 * STRIDE * 8)
        : "memory");
    const unsigned long long r[8] = {r0, r1, r2, r3, r4, r5, r6, r7};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e].x = __uint_as_float((unsigned)r[e]);
      v[e].y = __uint_as_float((unsigned)(r[e] >> 32));
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = base[STRIDE * e];
  }
}
template <int STRIDE, bool ASM>
__device__ __forceinline__ void lds_read8(cx<double> (&v)[8], const cx<double>* base) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = base[STRIDE * e];
}

// forward 512-point FFT inside one wavefront.  v[e] <-> y[lane + 64 e] on entry; on exit
// v[t] = Y[lane + 64 t]  (t < 2 only when PRUNE2).  lw = this wavefront's private LDS region.
// Two transposes through LDS, each laid out as a separable sum so that every access is one base register plus an
// immediate and conflict-free under the per-instruction banking of MI355X_MICROARCH.md (LDS): ds_read_b64 is served
// in two 32-lane halves over 32 bank pairs, ds_write_b64 in four groups of 16 contiguous lanes over 16 bank pairs.
//   T1 (k1 <-> n_mid):  j = 72 n_mid + (n_lo & 3) + 36 (n_lo >> 2) + 4 k1    scatter lane = 8 n_mid + n_lo, t = k1
//                                                                            gather  lane = 8 n_lo + k1,    e = n_mid
//   T2 (k2 <-> n_lo):   j = 72 n_lo + 8 k2 + k1                              scatter lane = 8 n_lo + k1,    t = k2
//                                                                            gather  lane = k1 + 8 k2,      e = n_lo
// (checked exhaustively on the host.  History: i + i/8 padding was 2-way on every gather; a first separable layout
// fixed the gathers but made the T1 scatter 2-way -- SQ_LDS_BANK_CONFLICT stayed at 32 cycles per wavefront-symbol
// through both.)  Largest index 571 < WAVE_LDS_ELEMS.
template <typename T, bool PRUNE2, bool TWB_REG = true, bool ASMRD = false>
__device__ __forceinline__ void wave_fft512(cx<T> (&v)[8], int lane, const cx<T> (&twb)[7],
                                            const cx<T>* __restrict__ twl, cx<T>* __restrict__ lw) {
  cx<T>* const sa = lw + 72 * (lane >> 3) + (lane & 3) + 36 * ((lane >> 2) & 1);
  cx<T>* const ga1 = lw + ((lane >> 3) & 3) + 36 * (lane >> 5) + 4 * (lane & 7);
  cx<T>* const sb = lw + 72 * (lane >> 3) + (lane & 7);
  cx<T>* const ga2 = lw + lane;
  dft8<T, false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
  for (int t = 0; t < 8; ++t) sa[4 * t] = v[t];
  wave_sync();
  lds_read8<72, ASMRD>(v, ga1);
  wave_sync();
#pragma unroll
  for (int t = 1; t < 8; ++t) v[t] = v[t] * (TWB_REG ? twb[t - 1] : twl[(t - 1) * 64 + lane]);
  dft8<T, false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
  for (int t = 0; t < 8; ++t) sb[8 * t] = v[t];
  wave_sync();
  lds_read8<72, ASMRD>(v, ga2);
  wave_sync();
#pragma unroll
  for (int t = 1; t < 8; ++t) v[t] = v[t] * twl[448 + (t - 1) * 64 + lane];
  if constexpr (PRUNE2) dft8_first2<T>(v);
  else dft8<T, false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}

// ---- radix-NW decimation-in-frequency front end -------------------------------------------------
// Thread gid (= 64*wave + lane) owns the butterflies m = gid*BPT + b, b < BPT = 8/NW.  Slot e = b*NW + t
// holds x[m + 512 t].  After the stage slot b*NW + s holds y_s[m] = (sum_t x[m+512t] W_NW^(ts)) W_N^(m s).
template <typename T, int NW>
struct DifTw {
  cx<T> w[NW > 1 ? (8 / NW) * (NW - 1) : 1];
};

template <typename T, int NW>
__device__ __forceinline__ void dif_tw_init(DifTw<T, NW>& d, int gid, const cx<T>* __restrict__ tw) {
  if constexpr (NW > 1) {
    constexpr int BPT = 8 / NW;
#pragma unroll
    for (int b = 0; b < BPT; ++b)
#pragma unroll
      for (int s = 1; s < NW; ++s) d.w[b * (NW - 1) + (s - 1)] = tw[(gid * BPT + b) * s];
  }
}

template <typename T, int NW>
__device__ __forceinline__ void frame_load(cx<T> (&v)[8], const cx<T>* __restrict__ src /* first useful sample */,
                                           int gid, int lane) {
  if constexpr (NW == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = nt_load(src + lane + 64 * e);
  } else {
    constexpr int BPT = 8 / NW;
#pragma unroll
    for (int t = 0; t < NW; ++t)
#pragma unroll
      for (int b = 0; b < BPT; ++b) v[b * NW + t] = nt_load(src + gid * BPT + b + 512 * t);
  }
}

template <typename T, int NW>
__device__ __forceinline__ void dif_stage(cx<T> (&v)[8], const DifTw<T, NW>& d) {
  if constexpr (NW > 1) {
    constexpr int BPT = 8 / NW;
#pragma unroll
    for (int b = 0; b < BPT; ++b) {
      if constexpr (NW == 2) dft2<T, false>(v[b * 2], v[b * 2 + 1]);
      else if constexpr (NW == 4) dft4<T, false>(v[b * 4], v[b * 4 + 1], v[b * 4 + 2], v[b * 4 + 3]);
      else dft8<T, false>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
      for (int s = 1; s < NW; ++s) v[b * NW + s] = v[b * NW + s] * d.w[b * (NW - 1) + (s - 1)];
    }
  }
}

// exchange: y_s[m] -> ex[s*576 + m]; then wavefront s gathers y_s[lane + 64 e].  Region s of the exchange
// buffer doubles as wavefront s's private scratch for its 512-point transform: the barrier in front of
// the next scatter is only reached by a wavefront that has finished its transform.
template <typename T, int NW>
__device__ __forceinline__ void dif_scatter(const cx<T> (&v)[8], int gid, cx<T>* __restrict__ ex) {
  constexpr int BPT = 8 / NW;
#pragma unroll
  for (int s = 0; s < NW; ++s)
#pragma unroll
    for (int b = 0; b < BPT; ++b) ex[s * WAVE_LDS_ELEMS + gid * BPT + b] = v[b * NW + s];
}
template <typename T, bool ASMRD = false>
__device__ __forceinline__ void dif_gather(cx<T> (&v)[8], int wave, int lane, const cx<T>* __restrict__ ex) {
  lds_read8<64, ASMRD>(v, ex + wave * WAVE_LDS_ELEMS + lane);
}

// ---------------------------------------------------------------------------------------------
// parameters shared by the three kernels
// ---------------------------------------------------------------------------------------------
template <typename T>
struct FastParams {
  int n_symb, t_guard, n_carrier, np, nd, k_atoms, taps, frame_words, bps;
  int comb_m;                // comb pilots 1 : comb : ... -> Nfft / comb (S^H Y is an inverse transform of that size), else 0
  const int16_t* prole;      // [nfft] pilot position of a carrier or -1
  const int16_t* drole;      // [nfft] data position of a carrier or -1
  const cx<T>* pilots;       // [np]
  const cx<T>* sct;          // [np][k_atoms] conj(S), atom index fastest
  const c64* gram;           // [k_atoms]
  const cx<T>* tw;           // [nfft]
  // workspace
  cx<T>* stash;              // [n_frames][n_carrier]  X(1..N_carrier, 1)
  cx<T>* ypil;               // [n_frames][np]
  int32_t* tap_idx;          // [n_frames][taps]   0-based atom index, -1 = unused
  c64* tap_x;                // [n_frames][taps]
  const cx<T>* h_in;         // [n_frames][n_carrier] channel estimate made by another stage (MMSE mode), or nullptr
  uint32_t descr;            // 0, or DESCR_ON | the DeScrambler register (bit m-1 = Register(m), m = 1..14): descramble per frame
};


// ---- reductions over an aligned group of LPF lanes (LPF = 4, 8, 16, 32 or 64), result in every lane of the
// group.  Inside a row of 16 lanes: DPP butterflies (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror) --
// no LDS round trip like ds_bpermute (__shfl_xor); across rows: v_readlane of the four row results.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }

template <typename F>
__device__ __forceinline__ int group_reduce_bits(int v, int LPF, int lane, F op) {
  v = op(v, dpp_mov<0xB1>(v));                     // quad_perm [1,0,3,2]
  v = op(v, dpp_mov<0x4E>(v));                     // quad_perm [2,3,0,1]
  if (LPF >= 8) v = op(v, dpp_mov<0x141>(v));      // row_half_mirror
  if (LPF >= 16) v = op(v, dpp_mov<0x140>(v));     // row_mirror
  if (LPF >= 32) {
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int lo = op(r0, r1), hi = op(r2, r3);
    v = LPF >= 64 ? op(lo, hi) : (lane < 32 ? lo : hi);
  }
  return v;
}
// scores are >= 0 or the initial -1: IEEE order of non-NaN floats
__device__ __forceinline__ float group_max_f(float v, int LPF, int lane) {
  return __int_as_float(group_reduce_bits(__float_as_int(v), LPF, lane, [](int a, int b) {
    return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b)));
  }));
}
__device__ __forceinline__ int group_min_i(int v, int LPF, int lane) {
  return group_reduce_bits(v, LPF, lane, [](int a, int b) { return a < b ? a : b; });
}

// ---- register-resident OMP (dominant_taps <= OMP_RT): the iteration loop is fully unrolled, so the number
// of picks made so far is the compile-time constant `it` and every index into L / z / x / pk is static.
// A frame is a group of LPF lanes (sl = lane inside the group); all lanes of the group compute the tiny solve
// redundantly from group-uniform inputs: no serial lane, no LDS state, no broadcast.  cf = c0 of the frame
// (LDS), gl = Gram table (LDS).  Writes tap_idx / tap_x of frame f.
constexpr int OMP_RT = 8;
template <typename T, int RT = OMP_RT>
__device__ __forceinline__ void omp_frame_reg(const FastParams<T>& P, const cx<T>* __restrict__ cf,
                                              const cx<T>* __restrict__ gl, int K, int taps, int LPF, int sl,
                                              bool live, double ynorm, int64_t f) {
  const int lane = threadIdx.x & 63;
  const T g0 = gl[0].x;
  // L is the lower Cholesky factor of the Gram of the picked atoms with 1/L[j][j] kept apart (tri(i,j) = i(i+1)/2 + j).
  int pk[RT];
  cx<T> xr[RT], zr[RT], Lr[RT * (RT + 1) / 2];
  T Ld[RT];                                      // 1 / L[j][j]
#pragma unroll
  for (int i = 0; i < RT; ++i) { pk[i] = -1; xr[i] = zr[i] = mk<T>(0, 0); Ld[i] = T(0); }
  bool active = live;
  int n = 0;
  double rho = ynorm;
#pragma unroll
  for (int it = 0; it < RT; ++it) {
    if (it >= taps) break;
    // residual correlation c = c0 - G(:,index) x and its first arg-max (OMP_estimate.m:7,:14)
    float bs = -1.0f;
    int bi = 0x7fffffff;
    for (int k = sl; k < K; k += LPF) {
      cx<T> c = cf[k];
#pragma unroll
      for (int qq = 0; qq < it; ++qq) {
        const int d = pk[qq] - k;
        const cx<T> gv = gl[d >= 0 ? d : -d];
        if constexpr (sizeof(T) == 4) cmsub(c, mk<T>(gv.x, d >= 0 ? gv.y : -gv.y), xr[qq]);      // four FMAs (fp32 mode)
        else c = c - (d >= 0 ? gv : conj(gv)) * xr[qq];
      }
      float sc;
      if constexpr (sizeof(T) == 4) sc = c.x * c.x + c.y * c.y;       // fp32 mode: near-ties fall under the 1e-4 rule
      else sc = (float)((double)c.x * c.x + (double)c.y * c.y);
      if (sc > bs) { bs = sc; bi = k; }          // ascending k inside a lane: strict > keeps the first
    }
    // first maximum over the group: the largest score, then the smallest index among the lanes that hold it
    const float gmax = group_max_f(bs, LPF, lane);
    bi = group_min_i(bs == gmax ? bi : 0x7fffffff, LPF, lane);
    const int kp = bi < K ? bi : 0;              // all-NaN scores: MATLAB max returns index 1
    if (active) {
      int dup = -1;
#pragma unroll
      for (int qq = 0; qq < it; ++qq) if (pk[qq] == kp) dup = qq;
      if (dup >= 0) {
        // pinv with a repeated column splits the coefficient equally; residual unchanged -> break
        cx<T> half = mk<T>(0, 0);
#pragma unroll
        for (int qq = 0; qq < it; ++qq) if (qq == dup) { xr[qq] = xr[qq] * (T)0.5; half = xr[qq]; }
        xr[it] = half;
        pk[it] = kp;
        n = it + 1;
        active = false;
      } else {
        pk[it] = kp;
        // new Cholesky row: G[it][j] = a_it^H a_j = gram[idx_j - idx_it]
        T dd = g0;
#pragma unroll
        for (int jq = 0; jq < it; ++jq) {
          const int d = pk[jq] - kp;
          cx<T> sgm = d >= 0 ? gl[d] : conj(gl[-d]);
#pragma unroll
          for (int k2 = 0; k2 < jq; ++k2) {
            if constexpr (sizeof(T) == 4) cmsubc(sgm, Lr[it * (it + 1) / 2 + k2], Lr[jq * (jq + 1) / 2 + k2]);
            else sgm = sgm - mulc(Lr[it * (it + 1) / 2 + k2], Lr[jq * (jq + 1) / 2 + k2]);
          }
          const cx<T> l = sgm * Ld[jq];
          Lr[it * (it + 1) / 2 + jq] = l;
          dd -= norm2(l);
        }
        const T inv_lnn = T(1) / sqrt(dd);
        Ld[it] = inv_lnn;
        // forward substitution (only the new entry changes): b_it = a_it^H y = c0[kp]
        cx<T> sz = cf[kp];
#pragma unroll
        for (int k2 = 0; k2 < it; ++k2) {
          if constexpr (sizeof(T) == 4) cmsub(sz, Lr[it * (it + 1) / 2 + k2], zr[k2]);
          else sz = sz - Lr[it * (it + 1) / 2 + k2] * zr[k2];
        }
        const cx<T> zn = sz * inv_lnn;
        zr[it] = zn;
        // back substitution L^H x = z
#pragma unroll
        for (int r = it; r >= 0; --r) {
          cx<T> acc = zr[r];
#pragma unroll
          for (int k2 = r + 1; k2 <= it; ++k2) {
            if constexpr (sizeof(T) == 4) cmsubc(acc, xr[k2], Lr[k2 * (k2 + 1) / 2 + r]);
            else acc = acc - mulc(xr[k2], Lr[k2 * (k2 + 1) / 2 + r]);
          }
          xr[r] = acc * Ld[r];
        }
        // ||r_n||^2 = ||r_{n-1}||^2 - |z_n|^2 ; stop when ||r_n - r_{n-1}|| / ||r_{n-1}|| < 1e-2 (:20)
        const double num = (double)zn.x * zn.x + (double)zn.y * zn.y;
        // (||r_n - r_{n-1}||^2 = |z_n|^2, compared squared: no double division / square root on the critical path)
        if (it >= 1 && (!(num > 0.0) || num < 1e-4 * rho)) active = false;
        rho -= num;
        n = it + 1;
      }
    }
  }
  // est_fade_chan(index(i1)) = x(i1): a later duplicate overwrites an earlier one (:31-33)
  if (live && sl == 0) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      if (t < taps) {
        int idx = -1;
        c64 xo{0, 0};
        if (t < n) {
          idx = pk[t];
          xo = c64{(double)xr[t].x, (double)xr[t].y};
#pragma unroll
          for (int q2 = t + 1; q2 < RT; ++q2) if (q2 < n && pk[q2] == idx) xo = c64{0, 0};
        }
        P.tap_idx[f * taps + t] = idx;
        P.tap_x[f * taps + t] = xo;
      }
    }
  }
}

// ---- dictionary correlation on the matrix cores: c0f (complex, row stride cs) [FB][16 n_tiles] = conj-dictionary^T * Y for
// the FB frames staged in LDS (Yl rows of YS elements, np pilots), columns 0 .. 16 n_tiles - 1 of sct ([np][ks], atom
// fastest).  One 256-thread workgroup (wave = 0..3).  Shared by the batch OMP (c0 = S^H Y) and the batch MP (S^H residue per
// iteration, MP_estimate.m:15).
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void corr_mfma_f32(const cx<float>* __restrict__ sct, int ks, int n_tiles,
                                              const cx<float>* __restrict__ Yl, int YS, int np, int FB,
                                              float* __restrict__ c0f, int cs, int wave, int lane) {
    // real GEMM  C[K x 2 FB] = A[K x 2np] * B[2np x 2 FB]:  A = [Re sct | Im sct]^T, column 2f = Re c0(f),
    // column 2f+1 = Im c0(f):  B(p,re ; 2f) = Yr, B(p,im ; 2f) = -Yi, B(p,re ; 2f+1) = Yi, B(p,im ; 2f+1) = Yr.
    // One k-step = 4 pilots -> two v_mfma_f32_16x16x4_f32 per 16x16 tile (real / imaginary parts of A).
    // A: lane (i = lane&15, q = lane>>4) supplies sct[p0+q][16*tile + i], loaded once per k-step and reused
    // for every 16-column group; B: lane (n = lane&15, q) supplies column n at pilot p0+q (LDS).
    // C: lane holds rows 4*(lane>>4)+r of column lane&15.
    const int i16 = lane & 15, q = lane >> 4;
    const int fsub = i16 >> 1, cim = i16 & 1;
    const int n_cg = (FB + 7) / 8;                 // 16-column groups (8 frames each): 1, 2 or 4 (FB = 4: half of one)
    for (int tile0 = wave * 2; tile0 < n_tiles; tile0 += 8) {
      const bool two = tile0 + 1 < n_tiles;
      f32x4 acc[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[a][g] = f32x4{0, 0, 0, 0};
      const cx<float>* a0p = sct + tile0 * 16 + i16;
      // The A operands come from L2 (the dictionary is shared by every workgroup); PD k-steps are kept in flight in
      // statically indexed registers (the loop is unrolled by PD): with one step in flight the stage was bound by
      // one L2 round trip per 4 pilots.
      constexpr int PD = 4;
      cx<float> aq0[PD], aq1[PD];
#pragma unroll
      for (int d = 0; d < PD; ++d) {
        const int pl = 4 * d < np ? 4 * d : 0;
        aq0[d] = a0p[(size_t)(pl + q) * ks];
        aq1[d] = two ? a0p[(size_t)(pl + q) * ks + 16] : mk<float>(0, 0);
      }
      for (int pb = 0; pb < np; pb += 4 * PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
          const int p0 = pb + 4 * d;
          if (p0 >= np) break;                                   // uniform
          const cx<float> a0 = aq0[d], a1 = aq1[d];
          const int pn = p0 + 4 * PD < np ? p0 + 4 * PD : p0;    // refill this slot for PD steps ahead
          aq0[d] = a0p[(size_t)(pn + q) * ks];
          aq1[d] = two ? a0p[(size_t)(pn + q) * ks + 16] : mk<float>(0, 0);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (g < n_cg) {
              const cx<float> yv = g * 8 + fsub < FB ? Yl[(g * 8 + fsub) * YS + p0 + q] : mk<float>(0, 0);
              const float b_re = cim ? yv.y : yv.x;        // multiplies Re(sct)
              const float b_im = cim ? yv.x : -yv.y;       // multiplies Im(sct)
              acc[0][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b_re, acc[0][g], 0, 0, 0);
              acc[0][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b_im, acc[0][g], 0, 0, 0);
              acc[1][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b_re, acc[1][g], 0, 0, 0);
              acc[1][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b_im, acc[1][g], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g < n_cg) {
          const int fcol = g * 8 + fsub;
          if (fcol >= FB) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            c0f[2 * (fcol * cs + tile0 * 16 + 4 * q + r) + cim] = acc[0][g][r];
            if (two) c0f[2 * (fcol * cs + (tile0 + 1) * 16 + 4 * q + r) + cim] = acc[1][g][r];
          }
        }
      }
    }
}

// Resident workgroups per CU of a persistent kernel (occupancy API: registers + LDS), cached per
// (kernel, dynamic LDS size); also raises the kernel's dynamic-LDS limit to `dyn`.  Workgroups are
// independent, so an optimistic answer only queues a few of them.
inline int resident_blocks_per_cu(const void* kern, int threads, size_t dyn) {
  static std::map<std::pair<const void*, size_t>, int> cache;      // entry points are not re-entrant
  const auto key = std::make_pair(kern, dyn);
  const auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int nb = 0;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, dyn) != hipSuccess || nb < 1) nb = 1;
  nb = nb < 8 ? nb : 8;
  cache[key] = nb;
  return nb;
}

// BA = bits per axis of a square QAM (2, 3, 4: the slicer thresholds become compile-time-indexed scalars;
// with a run-time switch every variant's constants stay live and the kernel spills ~80 SGPRs to VGPR lanes),
// BA = 0: any constellation through the generic decision function.
// Throughput mode (fp32 chains, EXACT = false): the arithmetic level rank of demap_square_arith -- decisions can differ from
// the threshold count only within a few ulp of a threshold (SURVEY 8c allows 1e-4), the centre threshold is compared exactly.
// Parity mode (fp64) and EXACT = true: every threshold counted.
template <typename T, int BA, bool EXACT = false>
__device__ __forceinline__ int slice_symbol(const DemapTable<T>& tab, cx<T> z) {
  if constexpr (BA == 0) return demap_decide(tab, z);
  else if constexpr (std::is_same<T, float>::value && BA >= 2 && !EXACT) return demap_square_arith<BA>(tab, z);
  else return demap_square<T, BA>(tab, z);
}

// Per-frame DeScrambler of the fused receivers (T5/DeScrambler.m:8-13 with the register reset per frame,
// T5/Main_model_Task_5.m:257-274): the register holds the last received bits, newest first, and the taps of array_xor pick
// Register(13), Register(14), so  d[i] = s[i] ^ s[i-13] ^ s[i-14]  with  s[-m] = Register0(m).  On the packed stream (a word
// in STREAM order = MSB first, i.e. before the byte swap of the little-endian store) that is two funnel shifts with the
// previous word, of which only the low 14 bits matter.  `DESCR_ON` marks the mode, the low 14 bits carry s[-1] (bit 0) ..
// s[-14] (bit 13) of the frame (or of the pack batch, for a kernel that packs a frame in several batches).
constexpr uint32_t DESCR_ON = 0x80000000u;
__device__ __forceinline__ uint32_t descr_word(uint32_t w, uint32_t prev) {
  return w ^ ((w >> 13) | (prev << 19)) ^ ((w >> 14) | (prev << 18));
}
// the 14 stream bits that precede code index `idx` (a multiple of 16, >= 16): the tail of the 16 codes before it
__device__ __forceinline__ uint32_t descr_tail(const uint8_t* __restrict__ codes, int idx, int bps) {
  const uint4 pc = *reinterpret_cast<const uint4*>(codes + idx - 16);
  const uint32_t cw[4] = {pc.x, pc.y, pc.z, pc.w};
  uint32_t acc = 0;                                   // bits shifted out at the top are older than 14 positions
#pragma unroll
  for (int i = 0; i < 16; ++i) acc = (acc << bps) | ((cw[i >> 2] >> (8 * (i & 3))) & 0xffu);
  return acc;
}
// bits of stream word `w` that belong to the frame (the padding of the last word stays 0 after descrambling as well)
__device__ __forceinline__ uint32_t descr_mask(int n_codes, int bps, int w) {
  const int valid = n_codes * bps - 32 * w;
  return valid >= 32 ? 0xffffffffu : (valid <= 0 ? 0u : ~(0xffffffffu >> valid));
}

// Pack stage of a frame: decided symbols `codes` (one byte each, LDS, zero-padded to a multiple of 32) -> packed bits
// (bit i of the frame -> byte i/8, bit 7 - i%8) + this thread's share of the BER numerator.  A group of 32 symbols is
// exactly `bps` 32-bit words.  BPS > 0 (square QAM, bps known at compile time): every shift is a constant, the group's
// reference words are requested before anything else and the stores leave at the end -- with a run-time bps the
// compiler emitted store -> load -> s_waitcnt vmcnt(0) once per word, up to eight dependent HBM round trips per thread
// (12 % of a frame's workgroup time in the in-kernel timestamps).  BPS = 0: run-time bps (any constellation).
// descr & DESCR_ON: the words are descrambled before they are stored / compared (ref_f then holds the TX's INPUT bits).
// DESCR is a template parameter because the wave-per-frame kernel packs with ~40 prefetched samples live: a run-time switch
// there cost 20 more spilled registers in the default mode.
template <int BPS, bool DESCR>
__device__ __forceinline__ unsigned pack_frame_t(const uint8_t* __restrict__ codes, int n_codes, int bps_rt, int frame_words,
                                                 uint32_t* __restrict__ out_f, const uint32_t* __restrict__ ref_f, int tid,
                                                 int nthr, uint32_t descr) {
  unsigned err = 0;
  const int n_groups = (n_codes + 31) >> 5;
  for (int grp = tid; grp < n_groups; grp += nthr) {
    if constexpr (BPS > 0) {
      const int w0 = grp * BPS;
      uint32_t refw[BPS], word[BPS];
#pragma unroll
      for (int j = 0; j < BPS; ++j) refw[j] = (ref_f && w0 + j < frame_words) ? ref_f[w0 + j] : 0u;
      const uint4 ca = *reinterpret_cast<const uint4*>(codes + 32 * grp);
      const uint4 cb = *reinterpret_cast<const uint4*>(codes + 32 * grp + 16);
      const uint32_t cw[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
      unsigned long long acc = 0;
      int nb = 0, wi = 0;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        acc = (acc << BPS) | ((cw[i >> 2] >> (8 * (i & 3))) & 0xffu);
        nb += BPS;
        if (nb >= 32) {
          nb -= 32;
          word[wi++] = (uint32_t)(acc >> nb);
        }
      }
      if constexpr (DESCR) {
        uint32_t prev = grp == 0 ? descr : descr_tail(codes, 32 * grp, BPS);
#pragma unroll
        for (int j = 0; j < BPS; ++j) {
          const uint32_t raw = word[j];
          word[j] = descr_word(raw, prev) & descr_mask(n_codes, BPS, w0 + j);
          prev = raw;
        }
      }
#pragma unroll
      for (int j = 0; j < BPS; ++j) {
        if (w0 + j < frame_words) {
          const uint32_t wd = __builtin_bswap32(word[j]);
          if (out_f) out_f[w0 + j] = wd;
          if (ref_f) err += __popc(wd ^ refw[j]);
        }
      }
    } else {
      const uint4 ca = *reinterpret_cast<const uint4*>(codes + 32 * grp);
      const uint4 cb = *reinterpret_cast<const uint4*>(codes + 32 * grp + 16);
      const uint32_t cw[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
      unsigned long long acc = 0;
      int nb = 0, w = grp * bps_rt;
      uint32_t prev = 0;
      if constexpr (DESCR) prev = grp == 0 ? descr : descr_tail(codes, 32 * grp, bps_rt);
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        acc = (acc << bps_rt) | ((cw[i >> 2] >> (8 * (i & 3))) & 0xffu);
        nb += bps_rt;
        if (nb >= 32) {
          nb -= 32;
          uint32_t raw = (uint32_t)(acc >> nb);
          if constexpr (DESCR) {
            const uint32_t d = descr_word(raw, prev) & descr_mask(n_codes, bps_rt, w);
            prev = raw;
            raw = d;
          }
          const uint32_t wd = __builtin_bswap32(raw);
          if (w < frame_words) {
            if (out_f) out_f[w] = wd;
            if (ref_f) err += __popc(wd ^ ref_f[w]);
          }
          ++w;
        }
      }
    }
  }
  return err;
}

// run-time form for the kernels whose pack stage is not register-critical (the switch is workgroup-uniform)
template <int BPS>
__device__ __forceinline__ unsigned pack_frame(const uint8_t* __restrict__ codes, int n_codes, int bps_rt, int frame_words,
                                               uint32_t* __restrict__ out_f, const uint32_t* __restrict__ ref_f, int tid,
                                               int nthr, uint32_t descr = 0) {
  if (descr & DESCR_ON) return pack_frame_t<BPS, true>(codes, n_codes, bps_rt, frame_words, out_f, ref_f, tid, nthr, descr);
  return pack_frame_t<BPS, false>(codes, n_codes, bps_rt, frame_words, out_f, ref_f, tid, nthr, 0u);
}

// What the fast / split paths need to know about an RX plan (ofdm_chain.hip owns the plan)
struct FastPlanView {
  int nfft, t_guard, n_symb, n_carrier, np, nd, k_atoms, taps, bps, f64, frame_words;
  const void *d_prole, *d_drole, *d_pilots, *d_sct, *d_gram;
  const std::vector<c64>* dict;
  const ConstellationInfo* cinfo;
  void **ws_stash, **ws_ypil, **ws_tapidx, **ws_tapx;   // workspace owned by the plan
  int64_t* ws_frames;
  hipEvent_t* ev;          // 4 events bracketing the three launches when timing is enabled, else nullptr
  int comb_lg_up;          // comb pilots with (Nfft/comb) dividing 512: log2(512 / (Nfft/comb)); -1 otherwise
  int comb_m;              // comb pilots: Nfft / comb, else 0
  int* fused_out;          // set to 1 when kernels 1+2 ran as one launch (then ev[2] is not recorded)
  const void* d_wt;        // MMSE mode: W^T [np][m_pad] in the plan's precision, else nullptr
  int m_pad;
  const void* d_mt;        // fp32 MMSE mode, factored form: M^T [np][np_pad], the banded spline per quad of rows (w [bw][nq][4], c0 [nq]); else nullptr
  int np_pad, sb_bw, sb_span;
  const float* d_sb_w;
  const int32_t* d_sb_c0;
  void** ws_v;             // [n_frames][np] MMSE estimate at the pilots
  int64_t* ws_v_frames;    // frames that workspace holds
  void** ws_h;             // MMSE mode workspace: H [n_frames][n_carrier]
  void** ws_x;             // split path workspace: X(1..N_carrier, :) of every symbol [n_frames * n_symb][n_carrier]
  int64_t* ws_x_elems;
  int data_mod4;           // bit r: a data carrier with 0-based index = r (mod 4) exists (the wave symbol kernel skips the other residues)
  uint32_t descr;          // per-frame DeScrambler of the pack stage: 0 or DESCR_ON | register bits (ofdm_rx_plan_set_descrambler)
};

template <typename T>
int fast_params_prepare(const FastPlanView& pv, const void* tw, int64_t n_frames, FastParams<T>& P);   // ofdm_chain_fast.hip
template <typename T>
int omp_batch_run(const FastParams<T>& P, int64_t n_frames);                                            // ofdm_chain_fast.hip

template <typename T>
int eq_demap_run(const FastPlanView& pv, const FastParams<T>& P, const cx<T>* xk, int x_stride, bool hext, int64_t n_frames,
                 void* bits, const void* ref, void* errs, void* h_out, void* idx_out, const double* fine_est, int time_desync,
                 int freq_desync);                                                                     // ofdm_chain_split.hip

// ofdm_chain_mmse.hip: the MMSE estimator of a plan as one operator W^T [np][m_pad] and its batched application
int mmse_build_operator(const c64* h, int64_t n_h, double snr_db, const int32_t* pilot_loc, int np, int n_carrier,
                        int m_pad, std::vector<c64>& wt, std::vector<c64>* mt_out = nullptr, int np_pad = 0,
                        std::vector<double>* sop_out = nullptr);
void mmse_band_spline(const std::vector<double>& sop, int nc, int np, std::vector<float>& w, std::vector<int32_t>& c0, int& bw, int& span);
int spline_band_run(const float* sb_w, const int32_t* sb_c0, int bw, int span, const void* v, void* hout, int np, int n_carrier,
                    int64_t n_frames);      // 1 = not taken (LDS), 0 ok, < 0 error
bool mmse_factored_usable(int np, int np_pad);
int mmse_factored_run(const void* mt, int np_pad, const float* sb_w, const int32_t* sb_c0, int bw, int span, const void* y, void* v,
                      void* hout, int np, int n_carrier, int64_t n_frames);
template <typename T>
int mmse_apply_run(const void* wt, const void* y, void* hout, int np, int m_pad, int n_carrier, int64_t n_frames);

// The MMSE estimate of every frame of a batch from its pilot LS values (P.ypil -> *pv.ws_h): fp32 plans with the factors
// take H = Sop_banded * (M * Y), everything else the dense operator.  v [n_frames][np] lives in a plan-owned workspace.
template <typename T>
inline int mmse_stage_run(const FastPlanView& pv, const FastParams<T>& P, int64_t n_frames) {
  if constexpr (std::is_same<T, float>::value) {
    if (pv.d_mt && mmse_factored_usable(pv.np, pv.np_pad)) {
      int64_t& cap = *pv.ws_v_frames;
      if (!*pv.ws_v || cap < n_frames) {
        OFDM_HIP(hipStreamSynchronize(ctx().stream));
        if (*pv.ws_v) { (void)hipFree(*pv.ws_v); *pv.ws_v = nullptr; }
        OFDM_HIP(hipMalloc(pv.ws_v, sizeof(cx<float>) * (size_t)pv.np * n_frames));
        cap = n_frames;
      }
      return mmse_factored_run(pv.d_mt, pv.np_pad, pv.d_sb_w, pv.d_sb_c0, pv.sb_bw, pv.sb_span, P.ypil, *pv.ws_v, *pv.ws_h, pv.np,
                               pv.n_carrier, n_frames);
    }
  }
  return mmse_apply_run<T>(pv.d_wt, P.ypil, *pv.ws_h, pv.np, pv.m_pad, pv.n_carrier, n_frames);
}

// ofdm_chain_wave.hip: the symbol stage with one wavefront per frame (Nfft 2048, fp32, N_carrier <= 512)
bool chain_wave_supported(const FastPlanView& pv);
int chain_wave_symbols_run(const FastPlanView& pv, const FastParams<float>& P, const void* rx, int64_t n_frames, void* bits,
                           const void* ref, void* errs, void* h_out, void* idx_out);

// ofdm_chain_coop.hip: the symbol stage for Nfft 8192 (fp32, N_carrier <= 2048) in one pass over the samples
bool chain_coop_supported(const FastPlanView& pv);
int chain_coop_symbols_run(const FastPlanView& pv, const FastParams<float>& P, const void* rx, int64_t n_frames, void* bits,
                           const void* ref, void* errs, void* h_out, void* idx_out);

// ofdm_chain_pilot.hip: symbol-1 transform + OMP of every frame in one launch (comb pilots, taps <= OMP_RT)
template <typename T>
int pilot_omp_run(const FastParams<T>& P, int nfft, bool prune2, int lg_up, const void* rx, int64_t n_frames);

}  // namespace ofdm
