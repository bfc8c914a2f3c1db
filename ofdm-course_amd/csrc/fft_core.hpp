// Workgroup-level radix-8 Stockham FFT for gfx950, data in registers, exchanges through LDS.
//
// One N-point transform is done by N/8 threads; thread j (0 <= j < N/8) owns eight points in
// registers.  Before the first pass and after the last pass slot e of thread j is the point with
// index j + e*N/8 -- so global loads / stores of one slot are contiguous across the 64 lanes of a
// wavefront (512 B per wave-instruction for complex float).  Between passes the points are
// transposed through LDS (autosort Stockham: no bit reversal anywhere).
//
// Pass with radix R and stride NS (= product of earlier radices), butterfly index jj:
//     k    = jj mod NS
//     in   = x[jj + t*N/R] * W_{NS*R}^{t*k}                 t = 0..R-1
//     out  = DFT_R(in) -> y[(jj/NS)*NS*R + k + t*NS]
// Radices: as many 8s as fit, then one 4 or 2.  A thread does 8/R butterflies in the last pass.
//
// LDS addressing is padded (one element every 8) so that the stride-8 scatter of the first
// pass is conflict-free for ds_write_b64 and later passes are at most 2-way.
#pragma once

#include "ofdm_common.hpp"

namespace ofdm {

__host__ __device__ constexpr int fft_pad(int i) { return i + (i >> 3); }
// LDS elements needed for one N-point transform
__host__ __device__ constexpr int fft_lds_elems(int n) { return n + (n >> 3) + 8; }

template <typename T, bool INV>
__device__ __forceinline__ void dft2(cx<T>& a, cx<T>& b) {
  cx<T> t = a - b;
  a = a + b;
  b = t;
}

// natural-order 4-point DFT
template <typename T, bool INV>
__device__ __forceinline__ void dft4(cx<T>& a0, cx<T>& a1, cx<T>& a2, cx<T>& a3) {
  cx<T> c0 = a0 + a2, c1 = a1 + a3, d0 = a0 - a2, d1 = mul_mi<T, INV>(a1 - a3);
  a0 = c0 + c1;
  a1 = d0 + d1;
  a2 = c0 - c1;
  a3 = d0 - d1;
}

// natural-order 8-point DFT (decimation in frequency, 3 radix-2 levels)
template <typename T, bool INV>
__device__ __forceinline__ void dft8(cx<T>& v0, cx<T>& v1, cx<T>& v2, cx<T>& v3,
                                     cx<T>& v4, cx<T>& v5, cx<T>& v6, cx<T>& v7) {
  const T h = T(0.70710678118654752440084436210485);
  cx<T> a0 = v0 + v4, a1 = v1 + v5, a2 = v2 + v6, a3 = v3 + v7;
  cx<T> b0 = v0 - v4, b1 = v1 - v5, b2 = v2 - v6, b3 = v3 - v7;
  // b_i *= W8^i  (forward: W8 = (1 - i)/sqrt2 ; inverse: conj)
  if (INV) {
    b1 = mk<T>((b1.x - b1.y) * h, (b1.x + b1.y) * h);
    b2 = mk<T>(-b2.y, b2.x);
    b3 = mk<T>((-b3.x - b3.y) * h, (b3.x - b3.y) * h);
  } else {
    b1 = mk<T>((b1.x + b1.y) * h, (b1.y - b1.x) * h);
    b2 = mk<T>(b2.y, -b2.x);
    b3 = mk<T>((b3.y - b3.x) * h, (-b3.x - b3.y) * h);
  }
  dft4<T, INV>(a0, a1, a2, a3);   // X0 X2 X4 X6
  dft4<T, INV>(b0, b1, b2, b3);   // X1 X3 X5 X7
  v0 = a0; v2 = a1; v4 = a2; v6 = a3;
  v1 = b0; v3 = b1; v5 = b2; v7 = b3;
}

template <typename T, int N, int R, int NS, bool INV>
__device__ __forceinline__ void fft_pass_compute(cx<T> (&v)[8], int j, const cx<T>* __restrict__ tw) {
  constexpr int NB = 8 / R;          // butterflies per thread
  constexpr int TSTEP = N / (NS * R);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if constexpr (NS > 1) {
      const int jj = j + b * (N / 8);
      const int k = jj & (NS - 1);
#pragma unroll
      for (int t = 1; t < R; ++t) {
        cx<T> w = tw[t * k * TSTEP];
        v[b + NB * t] = INV ? mulc(v[b + NB * t], w) : v[b + NB * t] * w;
      }
    }
    if constexpr (R == 8) {
      dft8<T, INV>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    } else if constexpr (R == 4) {
      dft4<T, INV>(v[b], v[b + 2], v[b + 4], v[b + 6]);
    } else {
      dft2<T, INV>(v[b], v[b + 4]);
    }
  }
}

template <typename T, int N, int R, int NS>
__device__ __forceinline__ void fft_pass_scatter(const cx<T> (&v)[8], int j, cx<T>* __restrict__ lds) {
  constexpr int NB = 8 / R;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int jj = j + b * (N / 8);
    const int base = ((jj / NS) * NS * R) + (jj & (NS - 1));
#pragma unroll
    for (int t = 0; t < R; ++t) lds[fft_pad(base + t * NS)] = v[b + NB * t];
  }
}

template <typename T, int N>
__device__ __forceinline__ void fft_gather(cx<T> (&v)[8], int j, const cx<T>* __restrict__ lds) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = lds[fft_pad(j + e * (N / 8))];
}

template <typename T, int N, int NS, bool INV>
__device__ __forceinline__ void fft_passes(cx<T> (&v)[8], int j, const cx<T>* __restrict__ tw,
                                           cx<T>* __restrict__ lds) {
  constexpr int REM = N / NS;
  constexpr int R = REM >= 8 ? 8 : REM;
  static_assert(R == 8 || R == 4 || R == 2, "N must be a power of two >= 8");
  fft_pass_compute<T, N, R, NS, INV>(v, j, tw);
  if constexpr (NS * R < N) {
    __syncthreads();                       // earlier gathers from this LDS region are done
    fft_pass_scatter<T, N, R, NS>(v, j, lds);
    __syncthreads();
    fft_gather<T, N>(v, j, lds);
    fft_passes<T, N, NS * R, INV>(v, j, tw, lds);
  }
}

// Full transform.  All threads of the workgroup must call it (it contains __syncthreads).
// v[e] <-> point j + e*N/8 on entry and on exit.  No 1/N scaling.
template <typename T, int N, bool INV>
__device__ __forceinline__ void wg_fft(cx<T> (&v)[8], int j, const cx<T>* __restrict__ tw,
                                       cx<T>* __restrict__ lds) {
  fft_passes<T, N, 1, INV>(v, j, tw, lds);
}

// threads per workgroup and transforms per workgroup for size N
__host__ __device__ constexpr int fft_threads_per_xform(int n) { return n / 8; }
__host__ __device__ constexpr int fft_wg_threads(int n) { return n / 8 >= 256 ? n / 8 : 256; }
__host__ __device__ constexpr int fft_xforms_per_wg(int n) { return fft_wg_threads(n) / (n / 8); }

// dispatch helper: calls F.template operator()<N>() for the supported sizes
#define OFDM_FFT_DISPATCH(nfft, CALL)                \
  switch (nfft) {                                    \
    case 64: { CALL(64); break; }                    \
    case 128: { CALL(128); break; }                  \
    case 256: { CALL(256); break; }                  \
    case 512: { CALL(512); break; }                  \
    case 1024: { CALL(1024); break; }                \
    case 2048: { CALL(2048); break; }                \
    case 4096: { CALL(4096); break; }                \
    case 8192: { CALL(8192); break; }                \
    default:                                         \
      ::ofdm::set_error("unsupported FFT size %d (power of two, 64..8192)", (int)(nfft)); \
      return OFDM_ERR_UNSUPPORTED;                   \
  }

inline bool fft_size_ok(int n) { return n >= 64 && n <= 8192 && (n & (n - 1)) == 0; }

}  // namespace ofdm
