// MEX gateway for constellation_func -- replaces Task 5/constellation_func.m:4-35
// MATLAB signature kept verbatim: [Dictionary, Bit_depth_Dict] = constellation_func(Constellation)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "constellation_func";
  (void)nlhs;
  need(nrhs == 1, fn, "one input expected");
  const std::string name = get_string(prhs[0], fn);
  c64 d[256];
  int bps = 0;
  check(ofdm_constellation_func(name.c_str(), d, &bps, OFDM_F64), fn);      // host-only entry, always double
  plhs[0] = mxCreateDoubleMatrix(1, (size_t)1 << bps, mxCOMPLEX);
  std::memcpy(mxGetComplexDoubles(plhs[0]), d, sizeof(c64) * ((size_t)1 << bps));
  if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)bps);
}
