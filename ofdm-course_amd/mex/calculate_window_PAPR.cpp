// MEX gateway for calculate_window_PAPR -- replaces Task 2/calculate_window_PAPR.m:2-15
// MATLAB signature kept verbatim: PAPRs = calculate_window_PAPR(Tx_OFDM_Signal, Nfft)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "calculate_window_PAPR";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  CBuf x = get_complex(prhs[0], fn);
  const int nfft = (int)get_scalar(prhs[1], fn);
  const int64_t n_out = (int64_t)x.n - nfft + 1;                           // :4
  plhs[0] = mxCreateDoubleMatrix(1, n_out > 0 ? (size_t)n_out : 0, mxREAL);
  check(ofdm_calculate_window_PAPR(x.ptr(), (int64_t)x.n, nfft, mxGetDoubles(plhs[0]), flags()), fn);
}
