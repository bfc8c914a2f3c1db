// MEX gateway for add_CFO -- replaces Task 5/add_CFO.m:1-8
// MATLAB signature kept verbatim: y_CFO = add_CFO(y, CFO, Nfft)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "add_CFO";
  (void)nlhs;
  need(nrhs == 3, fn, "three inputs expected");
  ensure_init();
  need(mxGetN(prhs[0]) == 1, fn, "y must be a column vector (y.*exp(...nn') would expand a row into a matrix)");
  CBuf x = get_complex(prhs[0], fn), y = alloc_complex(x.n);
  check(ofdm_add_CFO(x.ptr(), (int64_t)x.n, get_scalar(prhs[1], fn), (int)get_scalar(prhs[2], fn), y.ptr(), flags()), fn);
  plhs[0] = put_complex(y, x.n, 1);
}
