// MEX gateway for MP_estimate -- replaces Task 5/MP_estimate.m:1-34
// MATLAB signature kept verbatim: [H_MP, h_impulse_est] = MP_estimate(Y, sensing_matrix, Nfft, dominant_taps)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "MP_estimate";
  (void)nlhs;
  need(nrhs == 4, fn, "four inputs expected");
  ensure_init();
  const size_t np = mxGetM(prhs[1]), k = mxGetN(prhs[1]);
  const int nfft = (int)get_scalar(prhs[2], fn), taps = (int)get_scalar(prhs[3], fn);
  CBuf y = get_complex(prhs[0], fn), s = get_complex(prhs[1], fn), H = alloc_complex(nfft), h = alloc_complex(nfft);
  need(y.n == np, fn, "Y must have size(sensing_matrix,1) elements");
  check(ofdm_MP_estimate(y.ptr(), s.ptr(), (int)np, (int)k, nfft, taps, H.ptr(), h.ptr(), nullptr, flags()), fn);
  plhs[0] = put_complex(H, 1, nfft);                                          // fft(h_impulse_est).'
  if (nlhs > 1) plhs[1] = put_complex(h, nfft, 1);
}
