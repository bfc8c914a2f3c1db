// MEX gateway for the Task-4 receiver of a batch of frames -- replaces, per frame, Task 4/Main_model_Task_4.m:278-347
// (AutoCorrFunction -> add_STO x2 -> add_CFO -> remove_IFO -> OFDM_demodulator -> fine_sync -> estimate_channel ->
// equalize_signal -> get_payload -> demapping) and, with ref_bits, BER_func's numerator.
//
//   [bits, TgPosition, FreqOffset, IFO, status, errors, H_est] = ofdm_rx_chain_task4(Rx, Nfft, T_guard, N_carrier, ...
//        pilotCarriers, dataCarriers, pilotValues, Constellation, Time_Desync, Freq_Desync, MP_Desync, ref_bits, Register)
//   Rx   [(Nfft+T_guard)*N_symb x n_frames] complex, one received frame per column; the three flags are the script's
//        switches of :34-36; ref_bits optional [bits_per_frame x n_frames]; Register optional [1 x 15]: with it the frames go
//        through the per-frame DeScrambler of :354-364 before `bits` / `errors` (ref_bits = the payload before the Scrambler),
//        without it `bits` are the demapped (scrambled) bits and ref_bits the scrambled ones
//   status: 0 ok, 1 AutoCorrFunction's catch branch (TgPosition 65), -1 no IFO line above 0.77, -2 TgPosition out of range
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

namespace {
ofdm_rx_plan* g_plan = nullptr;
std::string g_key;
void drop_plan() { if (g_plan) { ofdm_rx_plan_destroy(g_plan); g_plan = nullptr; } }
void at_exit_chain() { drop_plan(); ofdm_shutdown(); }
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "ofdm_rx_chain_task4";
  need(nrhs >= 11 && nrhs <= 13, fn, "eleven to thirteen inputs expected");
  ensure_init();
  mexAtExit(at_exit_chain);
  const int nfft = (int)get_scalar(prhs[1], fn), tg = (int)get_scalar(prhs[2], fn), nc = (int)get_scalar(prhs[3], fn);
  const std::vector<int32_t> pc = get_index(prhs[4], fn), dc = get_index(prhs[5], fn);
  const std::string con = get_string(prhs[7], fn);
  const int td = get_scalar(prhs[8], fn) != 0.0, fd = get_scalar(prhs[9], fn) != 0.0, md = get_scalar(prhs[10], fn) != 0.0;
  const size_t rows = mxGetM(prhs[0]), n_frames = mxGetN(prhs[0]);
  need(nfft > 0 && tg >= 0 && rows % (size_t)(nfft + tg) == 0 && rows > 0, fn, "size(Rx,1) must be (Nfft+T_guard)*N_symb");
  const int n_symb = (int)(rows / (size_t)(nfft + tg));
  CBuf pv = get_complex(prhs[6], fn);
  need(pv.n >= pc.size(), fn, "pilotValues must have numel(pilotCarriers) rows");
  std::string key = std::to_string(nfft) + "/" + std::to_string(tg) + "/" + std::to_string(n_symb) + "/" + std::to_string(nc) + "/" + con +
                    "/" + (use_f32() ? "f32" : "f64");
  for (int32_t v : pc) key += "," + std::to_string(v);
  key += ";";
  for (int32_t v : dc) key += "," + std::to_string(v);
  key += ";";
  for (size_t i = 0; i < pc.size(); ++i)
    key += use_f32() ? std::to_string(pv.f[i].re) + "_" + std::to_string(pv.f[i].im) : std::to_string(pv.d[i].re) + "_" + std::to_string(pv.d[i].im);
  if (!g_plan || key != g_key) {
    drop_plan();
    // the Task-4 receiver uses neither a dictionary nor a tap count; the plan wants valid ones
    check(ofdm_rx_plan_create(&g_plan, nfft, tg, n_symb, nc, pc.data(), (int)pc.size(), dc.data(), (int)dc.size(), pv.ptr(),
                              (nc + 5) / 6, 3, con.c_str(), flags()), fn);
    g_key = key;
  }
  int bps = 0;
  {
    c64 dict[256];
    check(ofdm_constellation_func(con.c_str(), dict, &bps, OFDM_F64), fn);
  }
  const int64_t fb = ofdm_rx_plan_frame_bytes(g_plan);
  const size_t frame_bits = dc.size() * (size_t)n_symb * (size_t)bps;
  CBuf rx = get_complex(prhs[0], fn);
  std::vector<uint8_t> ref_packed;
  const bool have_ref = nrhs >= 12 && mxGetNumberOfElements(prhs[11]) > 0;
  if (have_ref) {
    const std::vector<uint8_t> rb = get_bits(prhs[11], fn);
    need(rb.size() == frame_bits * n_frames, fn, "ref_bits must be [bits_per_frame x n_frames]");
    ref_packed.assign((size_t)fb * n_frames, 0);
    for (size_t f = 0; f < n_frames; ++f)
      for (size_t i = 0; i < frame_bits; ++i)
        if (rb[f * frame_bits + i]) ref_packed[f * fb + i / 8] |= (uint8_t)(0x80u >> (i % 8));
  }
  // optional Register (1 x 15, Main_model_Task_4.m:44): the per-frame DeScrambler of :354-364 inside the call; [] = off
  if (nrhs == 13 && mxGetNumberOfElements(prhs[12]) > 0) {
    const std::vector<uint8_t> reg = get_bits(prhs[12], fn);
    need(reg.size() == 15, fn, "Register must have 15 entries");
    check(ofdm_rx_plan_set_descrambler(g_plan, reg.data()), fn);
  } else {
    check(ofdm_rx_plan_set_descrambler(g_plan, nullptr), fn);
  }
  std::vector<uint8_t> bits_packed((size_t)fb * n_frames);
  std::vector<uint32_t> errs(n_frames);
  std::vector<int64_t> tgp(n_frames);
  std::vector<double> fo(n_frames);
  std::vector<int32_t> ifo(n_frames), stat(n_frames);
  CBuf H = alloc_complex((size_t)nc * n_frames);
  check(ofdm_rx_chain_task4(g_plan, rx.ptr(), (int64_t)n_frames, td, fd, md, bits_packed.data(), have_ref ? ref_packed.data() : nullptr,
                            have_ref ? errs.data() : nullptr, tgp.data(), fo.data(), ifo.data(), stat.data(), nlhs > 6 ? H.ptr() : nullptr,
                            flags()), fn);
  std::vector<uint8_t> bits01(frame_bits * n_frames);
  for (size_t f = 0; f < n_frames; ++f)
    for (size_t i = 0; i < frame_bits; ++i) bits01[f * frame_bits + i] = (bits_packed[f * fb + i / 8] >> (7 - i % 8)) & 1u;
  plhs[0] = put_bits(bits01, frame_bits, n_frames);
  auto row = [&](int k, auto&& value) {
    if (nlhs > k) {
      plhs[k] = mxCreateDoubleMatrix(1, n_frames, mxREAL);
      for (size_t f = 0; f < n_frames; ++f) mxGetDoubles(plhs[k])[f] = value(f);
    }
  };
  row(1, [&](size_t f) { return (double)tgp[f]; });
  row(2, [&](size_t f) { return fo[f]; });
  row(3, [&](size_t f) { return (double)ifo[f]; });
  row(4, [&](size_t f) { return (double)stat[f]; });
  row(5, [&](size_t f) { return have_ref ? (double)errs[f] : 0.0; });
  if (nlhs > 6) plhs[6] = put_complex(H, nc, n_frames);
}
