// Batch OMP of ONE frame by ONE wavefront for more than OMP_RT taps (OMP_estimate.m:7-23; BASELINE config 5: 32 taps,
// K = Np = 512).  Used by omp_batch_kernel (ofdm_chain_fast.hip) after its c0 = S^H Y stage.
//
// What changed against the first form of this path (per-frame state in LDS, serial loops of dependent LDS reads: 25 k cycles
// per iteration, 0.38 ms per 2048 frames):
//   * pick q, its coefficient x_q, z_q and the scratch row l_q live in LANE q (one register each); a wave-uniform q reads
//     them with v_readlane -- no LDS round trip for anything that is indexed by the pick number;
//   * the Gram table is stored two-sided (g2[d + KP] = a_k^H a_{k+d}, negative shifts conjugated), so the residual
//     correlation c = c0 - G(:, index) x costs one LDS gather + four FMAs per (atom, pick) and no sign logic;
//   * R = L^-1 (lower triangular) sits in LDS with an ODD row stride: row j by lane j (new Cholesky row) and column j by
//     lane j (update of R) are both conflict-free; each is one pass over k with wave-uniform k;
//   * the duplicate test is one ballot, the norms are DPP reductions.
// Solve arithmetic in the data precision T (double in parity mode), as before.
#pragma once

#include "chain_fast_core.hpp"

namespace ofdm {

__device__ __forceinline__ float lane_bcast(float v, int q) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), q));
}
__device__ __forceinline__ double lane_bcast(double v, int q) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), q), hi = __builtin_amdgcn_readlane(__double2hiint(v), q);
  return __hiloint2double(hi, lo);
}
template <typename T>
__device__ __forceinline__ cx<T> lane_bcast(cx<T> v, int q) { return mk<T>(lane_bcast(v.x, q), lane_bcast(v.y, q)); }

__device__ __forceinline__ float wave_sum(float v, int lane) {
  return __int_as_float(group_reduce_bits(__float_as_int(v), 64, lane, [](int a, int b) {
    return __float_as_int(__int_as_float(a) + __int_as_float(b));
  }));
}
__device__ __forceinline__ double wave_sum(double v, int) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__host__ __device__ constexpr int omp_wave_rs(int taps) { return taps | 1; }          // odd row stride of R

// cf = c0 of the frame [K], g2 = two-sided Gram table [KP + K] (index d + KP, KP = K rounded up to 512), Rm = this wavefront's R [taps][omp_wave_rs(taps)]
// C0REG: c0 of the frame sits in registers (c0r[u] = c0[lane + 64 u], K <= 512; cf is not read) -- the LDS it came from is
// then free for the R states (omp_layout, reg_c0)
template <typename T, bool C0REG = false>
__device__ __forceinline__ void omp_frame_wave(const FastParams<T>& P, const cx<T>* __restrict__ cf, const cx<T> (&c0r)[8],
                                               const cx<T>* __restrict__ g2, cx<T>* __restrict__ Rm, int K, int taps,
                                               bool live, double ynorm, int64_t f) {
  const int lane = threadIdx.x & 63;
  const int RS = omp_wave_rs(taps);
  const int KP = (K + 511) & ~511;               // the table's zero shift sits at KP
  const T g0 = g2[KP].x;
  int pk = -1;                                   // lane q: pick q (0-based atom), x_q, z_q
  cx<T> xq = mk<T>(0, 0), zq = mk<T>(0, 0);
  int n = 0;
  double rho = ynorm;
  bool active = live;                            // wave-uniform
  if constexpr (sizeof(T) == 4) {
    for (int i = lane; i < taps * RS; i += 64) Rm[i] = mk<T>(0, 0);
    wave_sync();
  }
  for (int it = 0; it < taps && active; ++it) {
    const int ns = __builtin_amdgcn_readfirstlane(n);        // picks made so far, in a scalar register: uniform loops
    // ---- residual correlation c = c0 - G(:, index) x and its first arg-max (OMP_estimate.m:7,:14)
    float bs = -1.0f;
    int bi = 0x7fffffff;
    for (int kb = lane; kb < K; kb += 8 * 64) {
      cx<T> c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (C0REG) c[u] = c0r[u];                  // (one pass: K <= 512)
        else c[u] = kb + 64 * u < K ? cf[kb + 64 * u] : mk<T>(0, 0);
      }
      // g2[KP + pq - k], k = kb + 64 u: one address per pick (the pick is wave-uniform), eight reads at immediate offsets.
      // The table is KP + K long (KP = K rounded up to 512), so the atoms past the end of the dictionary read in bounds;
      // their scores are never looked at.
      const cx<T>* gk = g2 + KP - kb - 448;
      if (ns > 0) {
        cx<T> ga[8], gb[8];
        int pq = __builtin_amdgcn_readlane(pk, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) ga[u] = gk[pq + 64 * (7 - u)];
        for (int q = 0; q < ns; q += 2) {        // two picks per trip: the second pick's Gram values are in flight
          const cx<T> xa = lane_bcast(xq, q);
          const bool two = q + 1 < ns;
          const int pn = __builtin_amdgcn_readlane(pk, two ? q + 1 : q);
#pragma unroll
          for (int u = 0; u < 8; ++u) gb[u] = gk[pn + 64 * (7 - u)];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if constexpr (sizeof(T) == 4) cmsub(c[u], ga[u], xa); else c[u] = c[u] - ga[u] * xa;
          }
          if (two) {
            const cx<T> xb = lane_bcast(xq, q + 1);
            const int p2 = __builtin_amdgcn_readlane(pk, q + 2 < ns ? q + 2 : q + 1);
#pragma unroll
            for (int u = 0; u < 8; ++u) ga[u] = gk[p2 + 64 * (7 - u)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if constexpr (sizeof(T) == 4) cmsub(c[u], gb[u], xb); else c[u] = c[u] - gb[u] * xb;
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb + 64 * u;
        if (k < K) {
          float sc;
          if constexpr (sizeof(T) == 4) sc = c[u].x * c[u].x + c[u].y * c[u].y;     // fp32 mode: near-ties fall under the 1e-4 rule
          else sc = (float)((double)c[u].x * c[u].x + (double)c[u].y * c[u].y);
          if (sc > bs) { bs = sc; bi = k; }      // ascending k inside a lane: strict > keeps the first
        }
      }
    }
    const float gmax = group_max_f(bs, 64, lane);
    bi = group_min_i(bs == gmax ? bi : 0x7fffffff, 64, lane);
    const int kp = bi < K ? bi : 0;              // all-NaN scores: MATLAB max returns index 1
    // ---- a repeated pick: pinv with a repeated column splits the coefficient equally; residual unchanged -> break
    const unsigned long long dupm = __ballot(lane < n && pk == kp);
    if (dupm) {
      const int dup = __builtin_ctzll(dupm);
      if (lane == dup) xq = xq * (T)0.5;
      const cx<T> half = lane_bcast(xq, dup);
      if (lane == n) { xq = half; pk = kp; }
      n += 1;
      break;
    }
    // ---- new Cholesky row through R = L^-1:  l_j = sum_{k<=j} G(n,k) conj(R(j,k)),  G(n,k) = a_n^H a_k = g2[K + idx_k - idx_n]
    cx<T> g = mk<T>(0, 0);
    if (lane < n) g = g2[KP + pk - kp];
    cx<T> l = mk<T>(0, 0);
    // fp32: R's upper triangle holds zeros (cleared per frame below), so the sums run unconditionally over k < n; lanes
    // past the picks read a clamped row / column and their results are never used
    const int lc = lane < taps ? lane : taps - 1;
    const cx<T>* Rrow = Rm + (sizeof(T) == 4 ? lc : lane) * RS;
    for (int k = 0; k < ns; ++k) {
      const cx<T> gq = lane_bcast(g, k);
      if constexpr (sizeof(T) == 4) cmaddc(l, gq, Rrow[k]);
      else if (k <= lane && lane < n) l = l + mulc(gq, Rrow[k]);
    }
    const T nrm = wave_sum(lane < n ? norm2(l) : T(0), lane);
    const cx<T> lzp = l * zq;                    // zq = 0 in lanes >= n
    const cx<T> lz = mk<T>(wave_sum(lane < n ? lzp.x : T(0), lane), wave_sum(lane < n ? lzp.y : T(0), lane));
    const T inv = T(1) / sqrt(g0 - nrm);
    cx<T> bn;                                    // b_n = a_n^H y = c0[kp]
    if constexpr (C0REG) {
      cx<T> pick = c0r[0];
#pragma unroll
      for (int u = 1; u < 8; ++u) pick = (kp >> 6) == u ? c0r[u] : pick;       // kp is wave-uniform
      bn = lane_bcast(pick, kp & 63);
    } else {
      bn = cf[kp];
    }
    const cx<T> zn = (bn - lz) * inv;
    // ---- R(n, j) = -(1/lambda) sum_{k>=j} l_k R(k, j);  x_j += conj(R(n, j)) z_n (j < n);  x_n = z_n / lambda
    cx<T> r = mk<T>(0, 0);
    for (int k = 0; k < ns; ++k) {
      const cx<T> lq = lane_bcast(l, k);
      if constexpr (sizeof(T) == 4) cmadd(r, lq, Rm[k * RS + lc]);
      else if (k >= lane && lane < n) r = r + lq * Rm[k * RS + lane];
    }
    r = r * (-inv);
    if (lane < n) {
      Rm[n * RS + lane] = r;
      xq = xq + conj(r) * zn;
    }
    if (lane == n) {
      Rm[n * RS + n] = mk<T>(inv, 0);
      xq = zn * inv;
      zq = zn;
      pk = kp;
    }
    // ||r_n||^2 = ||r_{n-1}||^2 - |z_n|^2 ; stop when ||r_n - r_{n-1}|| / ||r_{n-1}|| < 1e-2 (:20), compared squared
    const double num = (double)zn.x * zn.x + (double)zn.y * zn.y;
    if (it >= 1 && (!(num > 0.0) || num < 1e-4 * rho)) active = false;
    rho -= num;
    n += 1;
    wave_sync();                                 // row n of R is read by other lanes in the next iteration
  }
  // est_fade_chan(index(i1)) = x(i1): a later duplicate overwrites an earlier one (:31-33)
  if (live) {
    bool later = false;
    for (int q = 1; q < n; ++q) {
      const int pq = __builtin_amdgcn_readlane(pk, q);
      if (lane < q && pk == pq) later = true;
    }
    if (lane < taps) {
      const bool have = lane < n;
      P.tap_idx[f * taps + lane] = have ? pk : -1;
      P.tap_x[f * taps + lane] = (have && !later) ? c64{(double)xq.x, (double)xq.y} : c64{0, 0};
    }
  }
}

}  // namespace ofdm
