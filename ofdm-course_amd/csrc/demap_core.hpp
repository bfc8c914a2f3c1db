// Hard-decision demapper shared by demapping / MER_func / the fused RX chain.
//
// T5/demapping.m:7-12 searches all 2^bps points for the minimum of
// (re-Dre)^2 + (im-Dim)^2 and `min` returns the FIRST minimum.  For the square QAMs the squared
// distance separates per axis, so the search is done per axis over the 2^(bps/2) axis codes in
// ascending code order with a strict '<' -- this reproduces the first-minimum rule exactly for
// exact ties (e.g. the all-zero carriers of a blanked symbol) because the flattened index is
// (I-code << bits_per_axis) | Q-code.  PSK/BPSK/QPSK use the full (<= 8 entry) search.
#pragma once

#include "ofdm_common.hpp"

namespace ofdm {

template <typename T>
struct DemapTable {
  // kind 0: pts[0..M) full table.  kind 1: axis_i[code], axis_q[code] (code < 2^bits_per_axis)
  int kind;
  int bps;
  int bits_per_axis;
  cx<T> pts[8];
  T axis_i[16];
  T axis_q[16];
  // square QAM slicer: decision thresholds between neighbouring levels in ascending order of the
  // coordinate (I axis: level rank l <-> code gray(l); Q axis: rank l <-> code gray(L-1-l))
  T thr_i[15];
  T thr_q[15];
  // level guess of demap_square_lut: floor(x * inv_d + off) is the level rank up to +-1 (thresholds are uniform up to rounding)
  T inv_d, off;
};

inline int gray_encode(int l) { return l ^ (l >> 1); }

// Largest x in [lo, hi] for which the reference's rule still prefers the LOWER level:
//   upper preferred  <=>  (x-hi)^2 < (x-lo)^2  ||  ((x-hi)^2 == (x-lo)^2 && upper_wins_tie)
// evaluated in T with separate multiply and add exactly like demapping.m:9, located by bisection
// down to adjacent floating-point numbers.  The device then decides with `x > threshold`.
template <typename T>
inline T demap_threshold(T lo, T hi, bool upper_wins_tie) {
#pragma clang fp contract(off)
  auto upper = [&](T x) {
    volatile T dl = (x - lo) * (x - lo);
    volatile T du = (x - hi) * (x - hi);
    return du < dl || (du == dl && upper_wins_tie);
  };
  T a = lo, b = hi;                       // upper(a) false, upper(b) true
  for (int it = 0; it < 4096; ++it) {
    const T m = a + (b - a) / 2;
    if (!(m > a && m < b)) break;         // a and b are adjacent
    if (upper(m)) b = m; else a = m;
  }
  return a;
}

template <typename T>
inline void fill_demap_table(const std::vector<c64>& dict, const ConstellationInfo& info, DemapTable<T>& t) {
  t.kind = info.kind;
  t.bps = info.bps;
  t.bits_per_axis = info.bits_per_axis;
  for (int i = 0; i < 8; ++i) t.pts[i] = mk<T>(0, 0);
  for (int i = 0; i < 16; ++i) t.axis_i[i] = t.axis_q[i] = T(0);
  if (info.kind == 0) {
    for (int i = 0; i < (1 << info.bps); ++i) t.pts[i] = mk<T>((T)dict[i].x, (T)dict[i].y);
  } else {
    const int ba = info.bits_per_axis, L = 1 << ba;
    for (int c = 0; c < L; ++c) {
      t.axis_i[c] = (T)dict[c << ba].x;   // any Q code: the I level only depends on the I code
      t.axis_q[c] = (T)dict[c].y;
    }
    for (int l = 0; l + 1 < L; ++l) {
      // first-minimum rule: on an exact tie the smaller CODE wins
      const int ci_lo = gray_encode(l), ci_hi = gray_encode(l + 1);
      t.thr_i[l] = demap_threshold<T>(t.axis_i[ci_lo], t.axis_i[ci_hi], ci_hi < ci_lo);
      const int cq_lo = gray_encode(L - 1 - l), cq_hi = gray_encode(L - 2 - l);
      t.thr_q[l] = demap_threshold<T>(t.axis_q[cq_lo], t.axis_q[cq_hi], cq_hi < cq_lo);
    }
  }
  for (int l = (info.kind == 1 ? (1 << info.bits_per_axis) - 1 : 0); l < 15; ++l) t.thr_i[l] = t.thr_q[l] = T(0);
  t.inv_d = T(1);
  t.off = T(0);
  if (info.kind == 1 && info.bits_per_axis >= 2) {
    const int L = 1 << info.bits_per_axis;
    const double d = ((double)t.thr_i[L - 2] - (double)t.thr_i[0]) / (L - 2);
    t.inv_d = (T)(1.0 / d);
    t.off = (T)(1.0 - (double)t.thr_i[0] / d);          // x just above thr[0] -> 1
  }
}

// Decision-threshold table of demap_square_lut in LDS: per axis L + 1 entries T[0] = -inf, T[l] = thr[l-1], T[L] = +inf
// (I axis first, Q axis at DEMAP_LUT_Q); the true level rank li is the one with T[li] < x <= T[li + 1].
constexpr int DEMAP_LUT_Q = 20, DEMAP_LUT_ELEMS = 40;
template <typename T, int BA>
__device__ __forceinline__ void demap_lut_fill(const DemapTable<T>& t, T* __restrict__ lut, int tid) {
  constexpr int L = 1 << BA;
  if (tid <= L) {
    const T inf = T(__builtin_huge_valf());
    T vi = tid == 0 ? -inf : inf, vq = vi;
#pragma unroll
    for (int l = 0; l < L - 1; ++l)
      if (tid == l + 1) { vi = t.thr_i[l]; vq = t.thr_q[l]; }
    lut[tid] = vi;
    lut[DEMAP_LUT_Q + tid] = vq;
  }
}

// The same decision as demap_square with 9 instead of 2 (L - 1) VALU operations per axis: an arithmetic guess of the level
// rank (exact up to +-1 because the thresholds are uniformly spaced up to rounding), then ONE comparison against each of the
// two neighbouring thresholds -- the very thresholds demap_square counts, so ties and the last ulp resolve identically.
template <typename T, int BA>
__device__ __forceinline__ int demap_square_lut(const DemapTable<T>& t, const T* __restrict__ lut, cx<T> z) {
  constexpr int L = 1 << BA;
  int g[2];
#pragma unroll
  for (int ax = 0; ax < 2; ++ax) {
    const T x = ax == 0 ? z.x : z.y;
    const T* tb = lut + (ax == 0 ? 0 : DEMAP_LUT_Q);
    int l0 = (int)floor(x * t.inv_d + t.off);                  // NaN -> 0
    l0 = l0 < 0 ? 0 : (l0 > L - 1 ? L - 1 : l0);
    const T lo = tb[l0], hi = tb[l0 + 1];
    int l = l0 + (x > hi ? 1 : 0) - (x > lo ? 0 : 1);
    g[ax] = l < 0 ? 0 : l;                                     // NaN: every comparison false, rank 0 like demap_square
  }
  int li = g[0], lq = g[1];
  if (z.y != z.y) lq = L - 1;              // NaN: every distance is NaN and `min` returns index 1
  const int ci = li ^ (li >> 1);
  const int lr = (L - 1) - lq;
  const int cq = lr ^ (lr >> 1);
  return (ci << BA) | cq;
}

// Throughput-mode slicer (fp32 chain only): the level rank from the signed distance to the CENTRE threshold in units of the level
// spacing.  On I the rank is the number of thresholds strictly below x = HALF - 1 + ceil((x - mid) / d); on Q the kernel needs
// the reversed rank L - 1 - (thresholds strictly below y) = HALF + floor((mid - y) / d).  Both hold the reference's tie rule at
// the centre exactly (x == mid: rank HALF - 1; the difference x - mid is exact there), so blanked carriers decide as in
// demap_square; the other thresholds are k * d from the centre up to rounding, so a decision can differ from demap_square only
// for a coordinate within a few ulp (~1e-7 relative) of a threshold -- inside the 1e-4 band SURVEY.md section 8c allows the fp32
// mode.  The clamp is done on the float (v_max / v_min return the non-NaN operand: NaN lands on rank 0 on I and on reversed
// rank 0 = rank L - 1 on Q, demap_square's outcome).  6 VALU operations per axis (sub, mul, max, min, ceil / floor, cvt + the
// add folded into the Gray step) -- the first arithmetic form (|.|, truncate, min, select by sign, two NaN selects) took 10,
// the threshold count 14 at 64-QAM.
template <int BA>
__device__ __forceinline__ int demap_square_arith(const DemapTable<float>& t, cx<float> z) {
  constexpr int L = 1 << BA, HALF = L / 2;
  float ui = (z.x - t.thr_i[HALF - 1]) * t.inv_d;
  float uq = (t.thr_q[HALF - 1] - z.y) * t.inv_d;
  ui = fminf(fmaxf(ui, -(float)(HALF - 1)), (float)HALF);
  uq = fminf(fmaxf(uq, -(float)HALF), (float)(HALF - 1));
  const int li = (HALF - 1) + (int)ceilf(ui);
  const int lr = HALF + (int)floorf(uq);
  const int ci = li ^ (li >> 1);
  const int cq = lr ^ (lr >> 1);
  return (ci << BA) | cq;
}

// Square QAM: the squared distance separates per axis and is unimodal along an axis, so the
// per-axis first-minimum search equals "count the decision thresholds below the coordinate".  The
// thresholds come from demap_threshold (host), which evaluates the reference's own comparison, ties
// included.  All table indices are compile-time constants (scalar registers, no table loads).
template <typename T, int BA>
__device__ __forceinline__ int demap_square(const DemapTable<T>& t, cx<T> z) {
  constexpr int L = 1 << BA;
  int li = 0, lq = 0;
#pragma unroll
  for (int c = 0; c < L - 1; ++c) {
    li += (z.x > t.thr_i[c]) ? 1 : 0;
    lq += (z.y > t.thr_q[c]) ? 1 : 0;
  }
  if (z.y != z.y) lq = L - 1;              // NaN: every distance is NaN and `min` returns index 1
  const int ci = li ^ (li >> 1);
  const int lr = (L - 1) - lq;
  const int cq = lr ^ (lr >> 1);
  return (ci << BA) | cq;
}

template <typename T, int M>
__device__ __forceinline__ int demap_table(const DemapTable<T>& t, cx<T> z) {
#pragma clang fp contract(off)
  int best = 0;
  T bd = (z.x - t.pts[0].x) * (z.x - t.pts[0].x) + (z.y - t.pts[0].y) * (z.y - t.pts[0].y);
#pragma unroll
  for (int i = 1; i < M; ++i) {
    const T d = (z.x - t.pts[i].x) * (z.x - t.pts[i].x) + (z.y - t.pts[i].y) * (z.y - t.pts[i].y);
    const bool l = d < bd;
    bd = l ? d : bd;
    best = l ? i : best;
  }
  return best;
}

// demapping.m:9 squares and adds as separate element-wise operations: no fused multiply-add
// (contract off), so exact ties (and only those) resolve to the first minimum like the reference.
template <typename T>
__device__ __forceinline__ int demap_decide(const DemapTable<T>& t, cx<T> z) {
  if (t.kind == 0) {
    switch (t.bps) {
      case 1: return demap_table<T, 2>(t, z);
      case 2: return demap_table<T, 4>(t, z);
      default: return demap_table<T, 8>(t, z);
    }
  }
  switch (t.bits_per_axis) {
    case 1: return demap_square<T, 1>(t, z);
    case 2: return demap_square<T, 2>(t, z);
    case 3: return demap_square<T, 3>(t, z);
    default: return demap_square<T, 4>(t, z);
  }
}

// constellation point of a symbol index (for MER and mapping)
template <typename T>
__device__ __forceinline__ cx<T> demap_point(const DemapTable<T>& t, int idx) {
  if (t.kind == 0) return t.pts[idx];
  return mk<T>(t.axis_i[idx >> t.bits_per_axis], t.axis_q[idx & ((1 << t.bits_per_axis) - 1)]);
}

}  // namespace ofdm
