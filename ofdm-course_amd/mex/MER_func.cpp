// MEX gateway for MER_func -- replaces Task 5/MER_func.m:1-26
// MATLAB signature kept verbatim: MER = MER_func(IQ_RX, Constellation)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "MER_func";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  CBuf x = get_complex(prhs[0], fn);
  double mer = 0;
  check(ofdm_MER_func(x.ptr(), (int64_t)x.n, get_string(prhs[1], fn).c_str(), &mer, flags()), fn);
  plhs[0] = mxCreateDoubleScalar(mer);
}
