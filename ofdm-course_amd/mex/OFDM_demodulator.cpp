// MEX gateway for OFDM_demodulator -- replaces Task 5/OFDM_demodulator.m:2-10
// MATLAB signature kept verbatim: TX_IQ = OFDM_demodulator(OFDM_time_guarded, T_guard)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "OFDM_demodulator";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  const size_t rows = mxGetM(prhs[0]), ns = mxGetN(prhs[0]);
  const int tg = (int)get_scalar(prhs[1], fn);
  need((size_t)tg < rows, fn, "T_guard must be smaller than the number of rows");
  const size_t nfft = rows - tg;
  CBuf y = get_complex(prhs[0], fn), x = alloc_complex(nfft * ns);
  check(ofdm_OFDM_demodulator(y.ptr(), x.ptr(), (int)nfft, (int64_t)ns, tg, flags()), fn);
  plhs[0] = put_complex(x, nfft, ns);
}
