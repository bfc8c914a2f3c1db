// MEX gateway for one scenario of the pilot-count study -- replaces the body of the `for jj = 1:monteCarloRuns` loop of
// Task 5/Task5_part2.m:148-306 (the loop its comment at :146 offers to `parfor`): every channel realisation of the scenario,
// LS_CE / MMSE_CE / MP_estimate / OMP_estimate on each, NMSE against fft(h) and the four equalise / demap / BER passes.
//
//   [nmse, errors] = ofdm_task5_part2_tile(Tx_noised, Nfft, T_guard, N_carrier, pilotCarriers, dataCarriers, pilotValues, ...
//                                          K, dominant_taps, Constellation, SNR_dB, tap_delay, tap_amp, ref_bits)
//   Tx_noised     [(Nfft+T_guard)*N_symb x 1]: the scenario's noisy TX stream (:130-134)
//   tap_delay     [n_taps x n_runs] 0-based sample delays, tap_amp [n_taps x n_runs] complex: the realisations' channels (:150-155)
//   ref_bits      [bits_per_frame x 1] 0/1 payload of the scenario
//   nmse, errors  [4 x n_runs]: rows LS, MMSE, MP, OMP (:202-205, :269-304) -- the caller sums them over jj as :309-318 do
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "ofdm_task5_part2_tile";
  need(nrhs == 14, fn, "fourteen inputs expected");
  ensure_init();
  const int nfft = (int)get_scalar(prhs[1], fn), tg = (int)get_scalar(prhs[2], fn), nc = (int)get_scalar(prhs[3], fn);
  const std::vector<int32_t> pc = get_index(prhs[4], fn), dc = get_index(prhs[5], fn);
  const int k_atoms = (int)get_scalar(prhs[7], fn), taps = (int)get_scalar(prhs[8], fn);
  const std::string con = get_string(prhs[9], fn);
  const double snr_db = get_scalar(prhs[10], fn);
  const size_t rows = mxGetNumberOfElements(prhs[0]);
  need(nfft > 0 && tg >= 0 && rows % (size_t)(nfft + tg) == 0 && rows > 0, fn, "numel(Tx_noised) must be (Nfft+T_guard)*N_symb");
  const int n_symb = (int)(rows / (size_t)(nfft + tg));
  CBuf pv = get_complex(prhs[6], fn), tx = get_complex(prhs[0], fn);
  need(pv.n >= pc.size(), fn, "pilotValues must have numel(pilotCarriers) rows");
  const size_t n_taps = mxGetM(prhs[11]), n_runs = mxGetN(prhs[11]);
  need(mxGetM(prhs[12]) == n_taps && mxGetN(prhs[12]) == n_runs && n_taps >= 1, fn, "tap_delay and tap_amp must have the same size");
  const std::vector<int32_t> delay = get_index(prhs[11], fn);                 // column-major [n_taps x n_runs] = [run][tap] rows
  std::vector<double> amp(2 * n_taps * n_runs);
  if (mxIsComplex(prhs[12])) {
    const mxComplexDouble* p = mxGetComplexDoubles(prhs[12]);
    for (size_t i = 0; i < n_taps * n_runs; ++i) { amp[2 * i] = p[i].real; amp[2 * i + 1] = p[i].imag; }
  } else {
    const double* p = mxGetDoubles(prhs[12]);
    for (size_t i = 0; i < n_taps * n_runs; ++i) { amp[2 * i] = p[i]; amp[2 * i + 1] = 0.0; }
  }
  ofdm_rx_plan* plan = nullptr;
  check(ofdm_rx_plan_create(&plan, nfft, tg, n_symb, nc, pc.data(), (int)pc.size(), dc.data(), (int)dc.size(), pv.ptr(), k_atoms, taps,
                            con.c_str(), flags()), fn);
  int bps = 0;
  {
    c64 dict[256];
    check(ofdm_constellation_func(con.c_str(), dict, &bps, OFDM_F64), fn);
  }
  const size_t frame_bits = dc.size() * (size_t)n_symb * (size_t)bps;
  const int64_t fb = ofdm_rx_plan_frame_bytes(plan);
  const std::vector<uint8_t> rb = get_bits(prhs[13], fn);
  std::vector<uint8_t> ref((size_t)fb, 0);
  if (rb.size() != frame_bits) { ofdm_rx_plan_destroy(plan); need(false, fn, "ref_bits must hold one frame of payload bits"); }
  for (size_t i = 0; i < frame_bits; ++i)
    if (rb[i]) ref[i / 8] |= (uint8_t)(0x80u >> (i % 8));
  std::vector<double> nmse(4 * n_runs);
  std::vector<uint32_t> errs(4 * n_runs);
  const int rc = ofdm_task5_part2_tile(plan, tx.ptr(), delay.data(), amp.data(), (int)n_taps, (int64_t)n_runs, snr_db, ref.data(), nmse.data(),
                                       errs.data(), flags());
  ofdm_rx_plan_destroy(plan);
  check(rc, fn);
  // library layout [4][n_runs] -> MATLAB [4 x n_runs] column-major
  plhs[0] = mxCreateDoubleMatrix(4, n_runs, mxREAL);
  for (size_t e = 0; e < 4; ++e)
    for (size_t j = 0; j < n_runs; ++j) mxGetDoubles(plhs[0])[e + 4 * j] = nmse[e * n_runs + j];
  if (nlhs > 1) {
    plhs[1] = mxCreateDoubleMatrix(4, n_runs, mxREAL);
    for (size_t e = 0; e < 4; ++e)
      for (size_t j = 0; j < n_runs; ++j) mxGetDoubles(plhs[1])[e + 4 * j] = (double)errs[e * n_runs + j];
  }
}
