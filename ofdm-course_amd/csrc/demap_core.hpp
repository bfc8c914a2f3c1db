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
};

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
  }
}

template <typename T>
__device__ __forceinline__ int demap_decide(const DemapTable<T>& t, cx<T> z) {
  // demapping.m:9 squares and adds as separate element-wise operations: no fused multiply-add here,
  // so exact ties (and only those) resolve to the first minimum exactly like the reference.
#pragma clang fp contract(off)
  if (t.kind == 0) {
    const int M = 1 << t.bps;
    int best = 0;
    T bd = (z.x - t.pts[0].x) * (z.x - t.pts[0].x) + (z.y - t.pts[0].y) * (z.y - t.pts[0].y);
    for (int i = 1; i < M; ++i) {
      T d = (z.x - t.pts[i].x) * (z.x - t.pts[i].x) + (z.y - t.pts[i].y) * (z.y - t.pts[i].y);
      if (d < bd) { bd = d; best = i; }
    }
    return best;
  }
  const int L = 1 << t.bits_per_axis;
  int bi = 0, bq = 0;
  T di = (z.x - t.axis_i[0]) * (z.x - t.axis_i[0]);
  T dq = (z.y - t.axis_q[0]) * (z.y - t.axis_q[0]);
  for (int c = 1; c < L; ++c) {
    T d = (z.x - t.axis_i[c]) * (z.x - t.axis_i[c]);
    if (d < di) { di = d; bi = c; }
    T e = (z.y - t.axis_q[c]) * (z.y - t.axis_q[c]);
    if (e < dq) { dq = e; bq = c; }
  }
  return (bi << t.bits_per_axis) | bq;
}

// constellation point of a symbol index (for MER and mapping)
template <typename T>
__device__ __forceinline__ cx<T> demap_point(const DemapTable<T>& t, int idx) {
  if (t.kind == 0) return t.pts[idx];
  return mk<T>(t.axis_i[idx >> t.bits_per_axis], t.axis_q[idx & ((1 << t.bits_per_axis) - 1)]);
}

}  // namespace ofdm
