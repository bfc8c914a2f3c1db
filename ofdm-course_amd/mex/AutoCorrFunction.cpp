// MEX gateway for AutoCorrFunction -- replaces Task 5/AutoCorrFunction.m:1-28
// MATLAB signature kept verbatim: [AutoCorr, TgPosition, FreqOffset] = AutoCorrFunction(RxSignal, WidthWindow, Nfft)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "AutoCorrFunction";
  (void)nlhs;
  need(nrhs == 3, fn, "three inputs expected");
  ensure_init();
  CBuf x = get_complex(prhs[0], fn);
  const int w = (int)get_scalar(prhs[1], fn), nfft = (int)get_scalar(prhs[2], fn);
  const int64_t n_out = (int64_t)x.n - w - nfft;
  CBuf rho = alloc_complex(n_out > 0 ? (size_t)n_out : 1);
  int64_t pos = 0;
  double fo = 0;
  const int rc = ofdm_AutoCorrFunction(x.ptr(), (int64_t)x.n, w, nfft, rho.ptr(), &pos, &fo, flags());
  check(rc, fn);
  if (rc == OFDM_SOFT_ACF_FALLBACK)          // the reference's catch branch (:21-24)
    mexWarnMsgIdAndTxt("ofdm:AutoCorrFunction:fallback", "Problem locating the guard interval position.");
  plhs[0] = put_complex(rho, 1, (size_t)n_out);
  if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)pos);
  if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(fo);
}
