// MEX gateway for add_STO -- replaces Task 5/add_STO.m:1-10
// MATLAB signature kept verbatim: y_STO = add_STO(y, nSTO)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "add_STO";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  need(mxGetN(prhs[0]) == 1, fn, "Dimensions of arrays being concatenated are not consistent (y must be a column).");
  CBuf x = get_complex(prhs[0], fn), y = alloc_complex(x.n);
  check(ofdm_add_STO(x.ptr(), (int64_t)x.n, (int64_t)get_scalar(prhs[1], fn), y.ptr(), flags()), fn);
  plhs[0] = put_complex(y, x.n, 1);
}
