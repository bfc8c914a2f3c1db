// MEX gateway for calculatePAPR -- replaces Task 2/calculatePAPR.m:2-11
// MATLAB signature kept verbatim: PAPR = calculatePAPR(OFDM_signal)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "calculatePAPR";
  (void)nlhs;
  need(nrhs == 1, fn, "one input expected");
  ensure_init();
  CBuf x = get_complex(prhs[0], fn);
  double papr = 0;
  check(ofdm_calculatePAPR(x.ptr(), (int64_t)x.n, &papr, flags()), fn);
  plhs[0] = mxCreateDoubleScalar(papr);
}
