// MEX gateway for get_MP_channel_resp -- replaces Task 5/get_MP_channel_resp.m:2-19
// MATLAB signature kept verbatim: [impulse_response, frequency_response] = get_MP_channel_resp(channel_taps, Nfft)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "get_MP_channel_resp";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  need(mxGetN(prhs[0]) == 2, fn, "channel_taps must be [n x 2] (delay, amplitude)");
  const int nt = (int)mxGetM(prhs[0]), nfft = (int)get_scalar(prhs[1], fn);
  std::vector<double> re(2 * (size_t)nt), im(nt, 0.0);
  if (mxIsComplex(prhs[0])) {
    const mxComplexDouble* p = mxGetComplexDoubles(prhs[0]);
    for (int i = 0; i < 2 * nt; ++i) re[i] = p[i].real;
    for (int i = 0; i < nt; ++i) im[i] = p[nt + i].imag;
  } else {
    std::memcpy(re.data(), mxGetDoubles(prhs[0]), sizeof(double) * 2 * nt);
  }
  int maxd = 0;
  for (int i = 0; i < nt; ++i) if ((int)re[i] > maxd) maxd = (int)re[i];
  std::vector<c64> h(maxd + 1), H(nfft);
  int hl = 0;
  check(ofdm_get_MP_channel_resp(re.data(), mxIsComplex(prhs[0]) ? im.data() : nullptr, nt, nfft, h.data(), &hl, H.data(), OFDM_F64), fn);
  plhs[0] = mxCreateDoubleMatrix(1, hl, mxCOMPLEX);
  std::memcpy(mxGetComplexDoubles(plhs[0]), h.data(), sizeof(c64) * hl);
  if (nlhs > 1) {
    plhs[1] = mxCreateDoubleMatrix(1, nfft, mxCOMPLEX);
    std::memcpy(mxGetComplexDoubles(plhs[1]), H.data(), sizeof(c64) * nfft);
  }
}
