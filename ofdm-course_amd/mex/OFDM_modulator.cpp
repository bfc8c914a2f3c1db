// MEX gateway for OFDM_modulator -- replaces Task 5/OFDM_modulator.m:2-11
// MATLAB signature kept verbatim: OFDM_time_guarded = OFDM_modulator(OFDM_symbols, T_guard)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "OFDM_modulator";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  const size_t nfft = mxGetM(prhs[0]), ns = mxGetN(prhs[0]);
  const int tg = (int)get_scalar(prhs[1], fn);
  CBuf x = get_complex(prhs[0], fn), y = alloc_complex((nfft + tg) * ns);
  check(ofdm_OFDM_modulator(x.ptr(), y.ptr(), (int)nfft, (int64_t)ns, tg, flags()), fn);
  plhs[0] = put_complex(y, nfft + tg, ns);
}
