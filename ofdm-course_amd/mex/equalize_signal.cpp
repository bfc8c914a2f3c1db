// MEX gateway for equalize_signal -- replaces Task 5/equalize_signal.m:1-8
// MATLAB signature kept verbatim: equalized_Hest = equalize_signal(OFDM_demod, Hest, N_carrier)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "equalize_signal";
  (void)nlhs;
  need(nrhs == 3, fn, "three inputs expected");
  ensure_init();
  const size_t nfft = mxGetM(prhs[0]), ns = mxGetN(prhs[0]);
  const int nc = (int)get_scalar(prhs[2], fn);
  CBuf x = get_complex(prhs[0], fn), h = get_complex(prhs[1], fn), y = alloc_complex(nfft * ns);
  need(h.n >= (size_t)nc, fn, "Index exceeds the number of array elements (Hest).");
  check(ofdm_equalize_signal(x.ptr(), (int)nfft, (int64_t)ns, h.ptr(), nc, y.ptr(), flags()), fn);
  plhs[0] = put_complex(y, nfft, ns);
}
