// MEX gateway for OMP_estimate -- replaces Task 5/OMP_estimate.m:1-37
// MATLAB signature kept verbatim: [H_OMP, h_impulse_est, index] = OMP_estimate(Y, sensing_matrix, Nfft, dominant_taps, SNR_dB)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "OMP_estimate";
  (void)nlhs;
  need(nrhs == 5, fn, "five inputs expected");
  ensure_init();
  const size_t np = mxGetM(prhs[1]), k = mxGetN(prhs[1]);
  const int nfft = (int)get_scalar(prhs[2], fn), taps = (int)get_scalar(prhs[3], fn);
  CBuf y = get_complex(prhs[0], fn), s = get_complex(prhs[1], fn), H = alloc_complex(nfft), h = alloc_complex(nfft);
  need(y.n == np && taps >= 1, fn, "Y must have size(sensing_matrix,1) elements");
  std::vector<int32_t> idx(taps);
  int n_idx = 0;
  check(ofdm_OMP_estimate(y.ptr(), s.ptr(), (int)np, (int)k, nfft, taps, get_scalar(prhs[4], fn), H.ptr(), h.ptr(),
                          idx.data(), &n_idx, flags()), fn);
  plhs[0] = put_complex(H, 1, nfft);
  if (nlhs > 1) plhs[1] = put_complex(h, 1, nfft);                            // est_fade_chan.' is a row
  if (nlhs > 2) {
    plhs[2] = mxCreateDoubleMatrix(1, n_idx, mxREAL);
    for (int i = 0; i < n_idx; ++i) mxGetDoubles(plhs[2])[i] = idx[i];
  }
}
