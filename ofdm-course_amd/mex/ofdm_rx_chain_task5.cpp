// MEX gateway for the fused Task-5 receiver of a batch of frames -- the benchmark path.  It replaces, per frame, the call
// sequence Task 5/Task5_part2.m:169-193,:296-303 (OFDM_demodulator -> Y = RX(pilotCarriers,1)./pilotValues(:,1) ->
// OMP_estimate -> equalize_signal -> get_payload -> demapping) and, with ref_bits, BER_func's numerator.
//
//   [bits, H_OMP, index, errors] = ofdm_rx_chain_task5(Rx, Nfft, T_guard, N_carrier, pilotCarriers, dataCarriers, ...
//                                                      pilotValues, K, dominant_taps, Constellation, ref_bits, Register)
//   Rx            [(Nfft+T_guard)*N_symb x n_frames] complex: every column one received frame (Rx_OFDM_mapped_carriers of :160-166)
//   pilotValues   [Np x N_symb] (or [Np x 1]): the pilot column of the first symbol is what the estimator uses (:190)
//   K             columns of the dictionary F(:,1:K) (:182-184); dominant_taps (:187)
//   ref_bits      optional [bits_per_frame x n_frames] 0/1: the transmitted payload
//   Register      optional [1 x 15]: the frames are descrambled (DeScrambler, register reset per frame, Main_model_Task_5.m:257-274)
//                 before `bits` and `errors` -- ref_bits then holds the payload BEFORE the Scrambler
//   bits          [bits_per_frame x n_frames] 0/1 demapped bits (get_payload order: column-major over [Nd x N_symb])
//   H_OMP         [N_carrier x n_frames];  index [dominant_taps x n_frames] (1-based picks, 0 = unused);  errors [1 x n_frames]
// N_symb is taken from size(Rx,1).  The plan (carrier tables, dictionary in closed form, Gram table) is built on the first
// call and kept until the geometry changes or MATLAB clears the MEX file.
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

namespace {
ofdm_rx_plan* g_plan = nullptr;
std::string g_key;
void drop_plan() { if (g_plan) { ofdm_rx_plan_destroy(g_plan); g_plan = nullptr; } }
void at_exit_chain() { drop_plan(); ofdm_shutdown(); }
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "ofdm_rx_chain_task5";
  need(nrhs >= 10 && nrhs <= 12, fn, "ten to twelve inputs expected");
  ensure_init();
  mexAtExit(at_exit_chain);
  const int nfft = (int)get_scalar(prhs[1], fn), tg = (int)get_scalar(prhs[2], fn), nc = (int)get_scalar(prhs[3], fn);
  const std::vector<int32_t> pc = get_index(prhs[4], fn), dc = get_index(prhs[5], fn);
  const int k_atoms = (int)get_scalar(prhs[7], fn), taps = (int)get_scalar(prhs[8], fn);
  const std::string con = get_string(prhs[9], fn);
  const size_t rows = mxGetM(prhs[0]), n_frames = mxGetN(prhs[0]);
  need(nfft > 0 && tg >= 0 && rows % (size_t)(nfft + tg) == 0 && rows > 0, fn, "size(Rx,1) must be (Nfft+T_guard)*N_symb");
  const int n_symb = (int)(rows / (size_t)(nfft + tg));
  CBuf pv = get_complex(prhs[6], fn);
  need(pv.n >= pc.size(), fn, "pilotValues must have numel(pilotCarriers) rows");
  // plan key: everything the plan depends on
  std::string key = std::to_string(nfft) + "/" + std::to_string(tg) + "/" + std::to_string(n_symb) + "/" + std::to_string(nc) + "/" +
                    std::to_string(k_atoms) + "/" + std::to_string(taps) + "/" + con + "/" + (use_f32() ? "f32" : "f64");
  for (int32_t v : pc) key += "," + std::to_string(v);
  key += ";";
  for (int32_t v : dc) key += "," + std::to_string(v);
  key += ";";
  for (size_t i = 0; i < pc.size(); ++i)
    key += use_f32() ? std::to_string(pv.f[i].re) + "_" + std::to_string(pv.f[i].im) : std::to_string(pv.d[i].re) + "_" + std::to_string(pv.d[i].im);
  if (!g_plan || key != g_key) {
    drop_plan();
    check(ofdm_rx_plan_create(&g_plan, nfft, tg, n_symb, nc, pc.data(), (int)pc.size(), dc.data(), (int)dc.size(), pv.ptr(), k_atoms,
                              taps, con.c_str(), flags()), fn);
    g_key = key;
  }
  const int64_t fb = ofdm_rx_plan_frame_bytes(g_plan);
  int bps = 0;
  {
    c64 dict[256];
    check(ofdm_constellation_func(con.c_str(), dict, &bps, OFDM_F64), fn);
  }
  const size_t frame_bits = dc.size() * (size_t)n_symb * (size_t)bps;
  CBuf rx = get_complex(prhs[0], fn);
  auto pack = [&](const std::vector<uint8_t>& bits01, std::vector<uint8_t>& packed) {
    packed.assign((size_t)fb * n_frames, 0);
    for (size_t f = 0; f < n_frames; ++f)
      for (size_t i = 0; i < frame_bits; ++i)
        if (bits01[f * frame_bits + i]) packed[f * fb + i / 8] |= (uint8_t)(0x80u >> (i % 8));
  };
  std::vector<uint8_t> ref_packed;
  const bool have_ref = nrhs >= 11 && mxGetNumberOfElements(prhs[10]) > 0;
  if (have_ref) {
    const std::vector<uint8_t> rb = get_bits(prhs[10], fn);
    need(rb.size() == frame_bits * n_frames, fn, "ref_bits must be [bits_per_frame x n_frames]");
    pack(rb, ref_packed);
  }
  // optional Register (1 x 15, Main_model_Task_5.m:55): DeScrambler per frame before bits / errors (:257-274); [] = off
  if (nrhs == 12 && mxGetNumberOfElements(prhs[11]) > 0) {
    const std::vector<uint8_t> reg = get_bits(prhs[11], fn);
    need(reg.size() == 15, fn, "Register must have 15 entries");
    check(ofdm_rx_plan_set_descrambler(g_plan, reg.data()), fn);
  } else {
    check(ofdm_rx_plan_set_descrambler(g_plan, nullptr), fn);
  }
  std::vector<uint8_t> bits_packed((size_t)fb * n_frames);
  std::vector<uint32_t> errs(n_frames);
  std::vector<int32_t> idx((size_t)taps * n_frames);
  CBuf H = alloc_complex((size_t)nc * n_frames);
  check(ofdm_rx_chain_task5(g_plan, rx.ptr(), (int64_t)n_frames, bits_packed.data(), have_ref ? ref_packed.data() : nullptr,
                            have_ref ? errs.data() : nullptr, nlhs > 1 ? H.ptr() : nullptr, nlhs > 2 ? idx.data() : nullptr, flags()), fn);
  std::vector<uint8_t> bits01(frame_bits * n_frames);
  for (size_t f = 0; f < n_frames; ++f)
    for (size_t i = 0; i < frame_bits; ++i) bits01[f * frame_bits + i] = (bits_packed[f * fb + i / 8] >> (7 - i % 8)) & 1u;
  plhs[0] = put_bits(bits01, frame_bits, n_frames);
  if (nlhs > 1) plhs[1] = put_complex(H, nc, n_frames);
  if (nlhs > 2) {
    plhs[2] = mxCreateDoubleMatrix(taps, n_frames, mxREAL);
    for (size_t i = 0; i < idx.size(); ++i) mxGetDoubles(plhs[2])[i] = idx[i];
  }
  if (nlhs > 3) {
    plhs[3] = mxCreateDoubleMatrix(1, n_frames, mxREAL);
    for (size_t f = 0; f < n_frames; ++f) mxGetDoubles(plhs[3])[f] = have_ref ? (double)errs[f] : 0.0;
  }
}
