// MEX gateway for MMSE_CE -- replaces Task 5/MMSE_CE.m:1-39
// MATLAB signature kept verbatim: H_MMSE = MMSE_CE(Y, Xp, pilot_loc, Nfft, N_carrier, h, SNR)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "MMSE_CE";
  (void)nlhs;
  need(nrhs == 7, fn, "seven inputs expected");
  ensure_init();
  const size_t nfft = mxGetM(prhs[0]), ns = mxGetN(prhs[0]);
  std::vector<int32_t> loc = get_index(prhs[2], fn);
  const int nc = (int)get_scalar(prhs[4], fn);
  CBuf y = get_complex(prhs[0], fn), xp = get_complex(prhs[1], fn), h = get_complex(prhs[5], fn), out = alloc_complex(nc);
  check(ofdm_MMSE_CE(y.ptr(), (int)nfft, (int64_t)ns, xp.ptr(), loc.data(), (int)loc.size(), nc, h.ptr(), (int)h.n,
                     get_scalar(prhs[6], fn), out.ptr(), flags()), fn);
  plhs[0] = put_complex(out, 1, nc);
}
