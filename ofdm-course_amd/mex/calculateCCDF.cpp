// MEX gateway for calculateCCDF -- replaces Task 2/calculateCCDF.m:2-6
// MATLAB signature kept verbatim: [PAPR_ccdf, CCDF] = calculateCCDF(PAPR_values)   (column outputs, like ecdf)
#include <vector>

#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "calculateCCDF";
  need(nrhs == 1 && mxIsDouble(prhs[0]), fn, "one real double input expected");
  ensure_init();
  const size_t n = mxGetNumberOfElements(prhs[0]);
  std::vector<double> xs(n + 1), cs(n + 1);
  int64_t n_out = 0;
  check(ofdm_calculateCCDF(mxGetDoubles(prhs[0]), (int64_t)n, xs.data(), cs.data(), &n_out, OFDM_F64 | OFDM_HOST), fn);
  plhs[0] = mxCreateDoubleMatrix((size_t)n_out, 1, mxREAL);
  for (int64_t i = 0; i < n_out; ++i) mxGetDoubles(plhs[0])[i] = xs[(size_t)i];
  if (nlhs > 1) {
    plhs[1] = mxCreateDoubleMatrix((size_t)n_out, 1, mxREAL);
    for (int64_t i = 0; i < n_out; ++i) mxGetDoubles(plhs[1])[i] = cs[(size_t)i];
  }
}
