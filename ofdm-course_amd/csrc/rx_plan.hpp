// The RX plan object behind the opaque `ofdm_rx_plan*` of include/ofdm_mi355x.h (owned by ofdm_chain.hip; the frame
// generator of ofdm_txgen.hip reads its geometry and tables).
#pragma once
#include <vector>

#include "chain_fast_core.hpp"

struct ofdm_rx_plan {
  int nfft, t_guard, n_symb, n_carrier, np, nd, k_atoms, taps, bps, f64;
  int frame_words;
  int pilots_in_band;
  int device = -1;         // the device the plan's buffers live on (= the context's device at creation)
  int comb_m = 0;          // comb pilots 1 : comb : ... -> Nfft / comb, else 0
  int comb_lg_up = -1;     // comb pilots 1 : comb : ... with (Nfft/comb) dividing 512 -> log2(512 / (Nfft/comb))
  void *d_prole, *d_drole, *d_pilots, *d_sct, *d_gram, *d_pc0;
  void *ws_stash = nullptr, *ws_ypil = nullptr, *ws_tapidx = nullptr, *ws_tapx = nullptr, *ws_h = nullptr;
  void* ws_x = nullptr;    // split path: X(1..N_carrier, :) of every symbol
  int64_t ws_x_elems = 0;
  void* d_wt = nullptr;    // MMSE mode (ofdm_rx_plan_set_mmse): W^T [np][m_pad]
  int m_pad = 0;
  void *d_mt = nullptr, *d_sb_w = nullptr, *d_sb_c0 = nullptr, *ws_v = nullptr;   // fp32 MMSE mode, factored: M^T, banded spline, v workspace
  int np_pad = 0, sb_bw = 0, sb_span = 0;
  int64_t ws_v_frames = 0;
  std::vector<int32_t> pilot_loc;      // 1-based, as given
  int data_mod4 = 15;                  // bit r set: some data carrier has (0-based) index = r mod 4
  int64_t ws_frames = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int timing = 0;          // ofdm_rx_plan_set_timing
  hipEvent_t ev_t4[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // stage brackets of ofdm_rx_chain_task4
  int t4_timed = 0;        // the last ofdm_rx_chain_task4 call recorded them
  int last_fast = 0;
  int last_fused = 0;      // fast path ran rx_pilot_omp_kernel instead of rx_pilot_kernel + omp_batch_kernel
  ofdm::ConstellationInfo cinfo;
  std::vector<ofdm::c64> dict;
  void* ws_gen = nullptr;  // ofdm_tx_frames: X / time-domain scratch for one chunk of frames
  size_t ws_gen_bytes = 0;
  void* d_dict = nullptr;  // constellation table in the plan's precision (ofdm_tx_frames)
  void* d_t4_tx = nullptr; // ofdm_rx_chain_task4: pilot matrix [np x n_symb] and spline operator [n_carrier x np] (built once)
  void* d_t4_w = nullptr;
  void *d_t4_bw = nullptr, *d_t4_bc0 = nullptr;      // fp32: the same operator cut to its band (spline_band_kernel)
  int t4_bw = 0, t4_span = 0;
  void* ws_t4 = nullptr;   // ofdm_rx_chain_task4: arena for its per-batch intermediates
  size_t ws_t4_bytes = 0;
  void* ws_raw = nullptr;  // raw packed decisions of a batch when the DeScrambler runs as a pass of its own (descr_pass_kernel)
  size_t ws_raw_bytes = 0;
  uint32_t descr = 0;      // per-frame DeScrambler of the pack stages (ofdm_rx_plan_set_descrambler): 0 or DESCR_ON | register bits
  void* d_p2_sop = nullptr; // ofdm_task5_part2_tile: spline operator of interpolate.m [n_carrier x np], double
};

// every entry that takes a plan: the context must still be on the plan's device
#define OFDM_PLAN_DEVICE(pl)                                                                                      \
  do {                                                                                                           \
    if ((pl)->device != ::ofdm::ctx().device) {                                                                  \
      ::ofdm::set_error("RX plan belongs to device %d, the context is on device %d", (pl)->device, ::ofdm::ctx().device); \
      return OFDM_ERR_STATE;                                                                                     \
    }                                                                                                            \
  } while (0)

// ofdm_chain.hip: DeScrambler as a pass over the packed decisions (every path but the wave-per-frame symbol kernel)
namespace ofdm {
int descr_raw_workspace(ofdm_rx_plan* pl, int64_t n_frames, void** raw);
int descr_pass_run(ofdm_rx_plan* pl, const void* raw, void* bits, const void* ref, void* errs, int64_t n_frames);
}

// view of a plan for the fast / split stages
inline void make_plan_view(ofdm_rx_plan* pl, ofdm::FastPlanView& pv) {
  pv.nfft = pl->nfft; pv.t_guard = pl->t_guard; pv.n_symb = pl->n_symb; pv.n_carrier = pl->n_carrier;
  pv.np = pl->np; pv.nd = pl->nd; pv.k_atoms = pl->k_atoms; pv.taps = pl->taps; pv.bps = pl->bps;
  pv.f64 = pl->f64; pv.frame_words = pl->frame_words;
  pv.d_prole = pl->d_prole; pv.d_drole = pl->d_drole; pv.d_pilots = pl->d_pilots; pv.d_sct = pl->d_sct;
  pv.d_gram = pl->d_gram; pv.dict = &pl->dict; pv.cinfo = &pl->cinfo;
  pv.ws_stash = &pl->ws_stash; pv.ws_ypil = &pl->ws_ypil; pv.ws_tapidx = &pl->ws_tapidx; pv.ws_tapx = &pl->ws_tapx;
  pv.ws_frames = &pl->ws_frames;
  pv.ev = pl->timing ? pl->ev : nullptr;
  pv.comb_lg_up = pl->comb_lg_up;
  pv.comb_m = pl->comb_m;
  pv.fused_out = &pl->last_fused;
  pv.d_wt = pl->d_wt; pv.m_pad = pl->m_pad; pv.ws_h = &pl->ws_h;
  pv.d_mt = pl->d_mt; pv.np_pad = pl->np_pad; pv.sb_bw = pl->sb_bw; pv.sb_span = pl->sb_span; pv.d_sb_w = (const float*)pl->d_sb_w;
  pv.d_sb_c0 = (const int32_t*)pl->d_sb_c0; pv.ws_v = &pl->ws_v; pv.ws_v_frames = &pl->ws_v_frames;
  pv.ws_x = &pl->ws_x; pv.ws_x_elems = &pl->ws_x_elems;
  pv.data_mod4 = pl->data_mod4;
  pv.descr = pl->descr;
}
