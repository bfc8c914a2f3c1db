// MATLAB interp1(...,'spline') (not-a-knot cubic, extrapolating) and interp1 linear as LINEAR
// OPERATORS on the knot values.  The knots are the pilot carriers -- fixed per pilot pattern -- so
// the global tridiagonal solve is done once on the host in double and the device only applies
// a real [n_out x n_in] matrix to complex pilot estimates ("not-a-knot spline").
//
// Restates: T5/interpolate.m:1-24 (end-knot extension folded into the operator),
// T5/estimate_channel.m:8.  MATLAB `spline` degeneracies: 2 knots -> line, 3 knots -> parabola.
#pragma once

#include <algorithm>
#include <vector>

#include "ofdm_common.hpp"

namespace ofdm {

// Tridiagonal solve with partial pivoting (dgtsv-style), in place.  dl/d/du: sub/main/super
// diagonals (dl[0] and du[n-1] unused), b: right-hand side -> solution.  Returns false if singular.
inline bool tridiag_solve(std::vector<double> dl, std::vector<double> d, std::vector<double> du,
                          std::vector<double>& b) {
  const int n = (int)d.size();
  std::vector<double> du2(n, 0.0);
  for (int i = 0; i < n - 1; ++i) {
    if (std::fabs(d[i]) >= std::fabs(dl[i + 1])) {
      if (d[i] == 0.0) return false;
      const double f = dl[i + 1] / d[i];
      d[i + 1] -= f * du[i];
      b[i + 1] -= f * b[i];
      dl[i + 1] = 0.0;
    } else {
      // swap rows i and i+1
      const double f = d[i] / dl[i + 1];
      d[i] = dl[i + 1];
      const double t = d[i + 1];
      d[i + 1] = du[i] - f * t;
      if (i + 2 < n) {
        du2[i] = du[i + 1];
        du[i + 1] = -f * du[i + 1];
      }
      du[i] = t;
      std::swap(b[i], b[i + 1]);
      b[i + 1] -= f * b[i];
    }
  }
  if (d[n - 1] == 0.0) return false;
  b[n - 1] /= d[n - 1];
  if (n > 1) b[n - 2] = (b[n - 2] - du[n - 2] * b[n - 1]) / d[n - 2];
  for (int i = n - 3; i >= 0; --i) b[i] = (b[i] - du[i] * b[i + 1] - du2[i] * b[i + 2]) / d[i];
  return true;
}

// W[q + j*nq] = d(spline(xq[q]))/d(y[j]) : column-major [nq x n]
inline int build_spline_operator(const std::vector<double>& x, const std::vector<double>& xq, std::vector<double>& W) {
  const int n = (int)x.size(), nq = (int)xq.size();
  OFDM_ARG(n >= 2, "spline: needs at least two knots");
  for (int i = 1; i < n; ++i) OFDM_ARG(x[i] > x[i - 1], "spline: knots must be strictly increasing");
  W.assign((size_t)n * nq, 0.0);
  if (n == 2) {
    for (int q = 0; q < nq; ++q) {
      const double t = (xq[q] - x[0]) / (x[1] - x[0]);
      W[q] = 1.0 - t;
      W[q + (size_t)nq] = t;
    }
    return OFDM_OK;
  }
  if (n == 3) {
    // Lagrange parabola through three points
    for (int q = 0; q < nq; ++q) {
      const double v = xq[q];
      W[q] = (v - x[1]) * (v - x[2]) / ((x[0] - x[1]) * (x[0] - x[2]));
      W[q + (size_t)nq] = (v - x[0]) * (v - x[2]) / ((x[1] - x[0]) * (x[1] - x[2]));
      W[q + 2 * (size_t)nq] = (v - x[0]) * (v - x[1]) / ((x[2] - x[0]) * (x[2] - x[1]));
    }
    return OFDM_OK;
  }
  std::vector<double> h(n - 1);
  for (int i = 0; i < n - 1; ++i) h[i] = x[i + 1] - x[i];
  // slope system (unknown knot slopes s_i), not-a-knot ends
  std::vector<double> dl(n, 0.0), d(n, 0.0), du(n, 0.0);
  for (int i = 1; i < n - 1; ++i) { dl[i] = h[i]; d[i] = 2.0 * (h[i - 1] + h[i]); du[i] = h[i - 1]; }
  d[0] = h[1]; du[0] = h[0] + h[1];
  d[n - 1] = h[n - 3]; dl[n - 1] = h[n - 2] + h[n - 3];
  // segment of every query point (extrapolation uses the end segments)
  std::vector<int> seg(nq);
  for (int q = 0; q < nq; ++q) {
    int s = (int)(std::upper_bound(x.begin(), x.end(), xq[q]) - x.begin()) - 1;
    seg[q] = std::min(std::max(s, 0), n - 2);
  }
  std::vector<double> delta(n - 1), rhs(n);
  for (int j = 0; j < n; ++j) {
    // unit knot vector e_j
    for (int i = 0; i < n - 1; ++i) delta[i] = ((i + 1 == j) ? 1.0 : 0.0) / h[i] - ((i == j) ? 1.0 : 0.0) / h[i];
    for (int i = 1; i < n - 1; ++i) rhs[i] = 3.0 * (h[i] * delta[i - 1] + h[i - 1] * delta[i]);
    rhs[0] = ((3.0 * h[0] + 2.0 * h[1]) * h[1] * delta[0] + h[0] * h[0] * delta[1]) / (h[0] + h[1]);
    rhs[n - 1] = (h[n - 2] * h[n - 2] * delta[n - 3] + (2.0 * h[n - 3] + 3.0 * h[n - 2]) * h[n - 3] * delta[n - 2]) /
                 (h[n - 3] + h[n - 2]);
    OFDM_ARG(tridiag_solve(dl, d, du, rhs), "spline: singular slope system");
    const std::vector<double>& s = rhs;
    for (int q = 0; q < nq; ++q) {
      const int g = seg[q];
      const double t = xq[q] - x[g], hs = h[g];
      const double c2 = (3.0 * delta[g] - 2.0 * s[g] - s[g + 1]) / hs;
      const double c3 = (s[g] + s[g + 1] - 2.0 * delta[g]) / (hs * hs);
      const double yj = (g == j) ? 1.0 : 0.0;
      W[q + (size_t)j * nq] = yj + t * (s[g] + t * (c2 + t * c3));
    }
  }
  return OFDM_OK;
}

inline int build_linear_operator(const std::vector<double>& x, const std::vector<double>& xq, std::vector<double>& W) {
  const int n = (int)x.size(), nq = (int)xq.size();
  OFDM_ARG(n >= 2, "interp1: needs at least two knots");
  for (int i = 1; i < n; ++i) OFDM_ARG(x[i] > x[i - 1], "interp1: knots must be strictly increasing");
  W.assign((size_t)n * nq, 0.0);
  for (int q = 0; q < nq; ++q) {
    const double v = xq[q];
    if (v < x[0] || v > x[n - 1]) {               // interp1 linear: NaN outside the knots
      for (int j = 0; j < n; ++j) W[q + (size_t)j * nq] = NAN;
      continue;
    }
    int s = (int)(std::upper_bound(x.begin(), x.end(), v) - x.begin()) - 1;
    s = std::min(std::max(s, 0), n - 2);
    const double t = (v - x[s]) / (x[s + 1] - x[s]);
    W[q + (size_t)s * nq] = 1.0 - t;
    W[q + (size_t)(s + 1) * nq] = t;
  }
  return OFDM_OK;
}

// T5/interpolate.m:1-24 as one operator on the ORIGINAL pilot values: end knots at 1 and N are
// linear extrapolations of the first / last two pilots (:7-16), then interp1 on 1..N (:18-22).
inline int build_interpolate_operator(const int32_t* pilot_loc, int np, int n_out, char method, std::vector<double>& W) {
  OFDM_ARG(np >= 2, "interpolate: needs at least two pilots");
  std::vector<double> x(pilot_loc, pilot_loc + np);
  // extension matrix E [n_ext x np] (row-major, sparse by construction)
  const bool front = x[0] > 1.0, back = x[np - 1] < (double)n_out;
  const int n_ext = np + (front ? 1 : 0) + (back ? 1 : 0);
  std::vector<double> xe;
  std::vector<std::vector<std::pair<int, double>>> E(n_ext);
  int row = 0;
  if (front) {
    // H(1)-slope*(loc(1)-1), slope = (H(2)-H(1))/(loc(2)-loc(1))
    const double a = (x[0] - 1.0) / (x[1] - x[0]);
    E[row] = {{0, 1.0 + a}, {1, -a}};
    xe.push_back(1.0);
    ++row;
  }
  for (int i = 0; i < np; ++i, ++row) { E[row] = {{i, 1.0}}; xe.push_back(x[i]); }
  if (back) {
    const double a = ((double)n_out - x[np - 1]) / (x[np - 1] - x[np - 2]);
    E[row] = {{np - 1, 1.0 + a}, {np - 2, -a}};
    xe.push_back((double)n_out);
    ++row;
  }
  std::vector<double> xq(n_out);
  for (int i = 0; i < n_out; ++i) xq[i] = i + 1.0;
  std::vector<double> We;
  if (method == 'l' || method == 'L') OFDM_TRY(build_linear_operator(xe, xq, We));
  else OFDM_TRY(build_spline_operator(xe, xq, We));
  W.assign((size_t)np * n_out, 0.0);
  for (int r = 0; r < n_ext; ++r)
    for (auto& e : E[r])
      for (int q = 0; q < n_out; ++q) W[q + (size_t)e.first * n_out] += We[q + (size_t)r * n_out] * e.second;
  return OFDM_OK;
}

}  // namespace ofdm
