// MEX gateway for the MSE(SNR) sweep of Task 5/Main_model_Task_5.m:303-346 -- replaces the body of `for i=1:length(SNRs)`:
// Noise -> conv -> OFDM_demodulator -> LS_CE -> MMSE_CE(h = ifft(H_est_LS), SNR) -> MP_estimate -> OMP_estimate -> four errors,
// for every SNR point in one device-resident call.
//
//   MSEs = ofdm_task5_mse_tile(Tx_OFDM_Signal, Nfft, T_guard, N_carrier, pilotCarriers, dataCarriers, pilotValues, ...
//                              K, dominant_taps, Constellation, channel_taps, SNRs, seed)
//   Tx_OFDM_Signal [(Nfft+T_guard)*N_symb x 1]: the clean TX stream (:85)
//   dataCarriers   may be empty (comb = 1, the script as committed: pilots on every carrier)
//   channel_taps   [n_taps x 2]: (delay, amplitude) rows (:289-296); SNRs [1 x n] (:303); seed optional (noise key)
//   MSEs           [4 x n]: rows LS, MMSE, MP, OMP (:304, :341-344)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "ofdm_task5_mse_tile";
  need(nrhs == 12 || nrhs == 13, fn, "twelve or thirteen inputs expected");
  ensure_init();
  const int nfft = (int)get_scalar(prhs[1], fn), tg = (int)get_scalar(prhs[2], fn), nc = (int)get_scalar(prhs[3], fn);
  const std::vector<int32_t> pc = get_index(prhs[4], fn);
  std::vector<int32_t> dc;
  if (mxGetNumberOfElements(prhs[5]) > 0) dc = get_index(prhs[5], fn);
  const int k_atoms = (int)get_scalar(prhs[7], fn), taps = (int)get_scalar(prhs[8], fn);
  const std::string con = get_string(prhs[9], fn);
  const size_t rows = mxGetNumberOfElements(prhs[0]);
  need(nfft > 0 && tg >= 0 && rows % (size_t)(nfft + tg) == 0 && rows > 0, fn, "numel(Tx) must be (Nfft+T_guard)*N_symb");
  const int n_symb = (int)(rows / (size_t)(nfft + tg));
  CBuf pv = get_complex(prhs[6], fn), tx = get_complex(prhs[0], fn);
  need(pv.n >= pc.size(), fn, "pilotValues must have numel(pilotCarriers) rows");
  const size_t n_taps = mxGetM(prhs[10]);
  need(mxGetN(prhs[10]) == 2 && n_taps >= 1, fn, "channel_taps must be [n_taps x 2]");
  std::vector<int32_t> delay(n_taps);
  std::vector<double> amp(2 * n_taps);
  if (mxIsComplex(prhs[10])) {
    const mxComplexDouble* p = mxGetComplexDoubles(prhs[10]);
    for (size_t t = 0; t < n_taps; ++t) { delay[t] = (int32_t)p[t].real; amp[2 * t] = p[n_taps + t].real; amp[2 * t + 1] = p[n_taps + t].imag; }
  } else {
    const double* p = mxGetDoubles(prhs[10]);
    for (size_t t = 0; t < n_taps; ++t) { delay[t] = (int32_t)p[t]; amp[2 * t] = p[n_taps + t]; amp[2 * t + 1] = 0.0; }
  }
  const size_t n = mxGetNumberOfElements(prhs[11]);
  need(!mxIsComplex(prhs[11]) && mxIsDouble(prhs[11]), fn, "SNRs must be a real double vector");
  const double* snr = mxGetDoubles(prhs[11]);
  const uint64_t seed = nrhs == 13 ? (uint64_t)get_scalar(prhs[12], fn) : 0;
  ofdm_rx_plan* plan = nullptr;
  check(ofdm_rx_plan_create(&plan, nfft, tg, n_symb, nc, pc.data(), (int)pc.size(), dc.data(), (int)dc.size(), pv.ptr(), k_atoms, taps,
                            con.c_str(), flags()), fn);
  std::vector<double> mse(4 * n);
  const int rc = ofdm_task5_mse_tile(plan, tx.ptr(), delay.data(), amp.data(), (int)n_taps, snr, (int64_t)n, seed, 1u, mse.data(), flags());
  ofdm_rx_plan_destroy(plan);
  check(rc, fn);
  plhs[0] = mxCreateDoubleMatrix(4, n, mxREAL);                     // library layout [4][n] -> MATLAB [4 x n] column-major
  for (size_t e = 0; e < 4; ++e)
    for (size_t j = 0; j < n; ++j) mxGetDoubles(plhs[0])[e + 4 * j] = mse[e * n + j];
}
