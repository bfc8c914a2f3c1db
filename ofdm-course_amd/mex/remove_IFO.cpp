// MEX gateway for remove_IFO -- replaces Task 5/remove_IFO.m:1-11
// MATLAB signature kept verbatim: [fixed_rx_signal, IFO] = remove_IFO(rx_signal, Nfft)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "remove_IFO";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  CBuf x = get_complex(prhs[0], fn), y = alloc_complex(x.n);
  int ifo = 0;
  check(ofdm_remove_IFO(x.ptr(), (int64_t)x.n, (int)get_scalar(prhs[1], fn), y.ptr(), &ifo, flags()), fn);
  plhs[0] = put_complex(y, x.n, 1);
  if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)ifo);
}
