// Small dense helpers shared by the single-problem OMP (ofdm_chanest.hip) and the fused chain.
#pragma once

#include "ofdm_common.hpp"

namespace ofdm {

// Hermitian positive-definite solve G x = b by Cholesky (n <= PUR_MAXT), single thread, double.
__device__ inline void chol_solve(const c64* __restrict__ G, const c64* __restrict__ b, c64* __restrict__ Lm,
                           c64* __restrict__ x, int n, int ld) {
  for (int j = 0; j < n; ++j) {
    double d = G[j * ld + j].x;
    for (int k = 0; k < j; ++k) d -= norm2(Lm[j * ld + k]);
    const double ljj = sqrt(d);
    Lm[j * ld + j] = c64{ljj, 0};
    for (int i = j + 1; i < n; ++i) {
      c64 s = G[i * ld + j];
      for (int k = 0; k < j; ++k) s = s - mulc(Lm[i * ld + k], Lm[j * ld + k]);
      Lm[i * ld + j] = c64{s.x / ljj, s.y / ljj};
    }
  }
  // forward: L z = b
  for (int i = 0; i < n; ++i) {
    c64 s = b[i];
    for (int k = 0; k < i; ++k) s = s - Lm[i * ld + k] * x[k];
    const double l = Lm[i * ld + i].x;
    x[i] = c64{s.x / l, s.y / l};
  }
  // backward: L^H x = z
  for (int i = n - 1; i >= 0; --i) {
    c64 s = x[i];
    for (int k = i + 1; k < n; ++k) s = s - mulc(x[k], Lm[k * ld + i]) ;   // conj(L[k][i]) * x[k]
    const double l = Lm[i * ld + i].x;
    x[i] = c64{s.x / l, s.y / l};
  }
}


}  // namespace ofdm
