// MEX gateway for LS_CE -- replaces Task 5/LS_CE.m:1-34
// MATLAB signature kept verbatim: H_LS = LS_CE(Y, Xp, pilot_loc, N_carrier)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "LS_CE";
  (void)nlhs;
  need(nrhs == 4, fn, "four inputs expected");
  ensure_init();
  const size_t nfft = mxGetM(prhs[0]), ns = mxGetN(prhs[0]);
  std::vector<int32_t> loc = get_index(prhs[2], fn);
  const int nc = (int)get_scalar(prhs[3], fn);
  CBuf y = get_complex(prhs[0], fn), xp = get_complex(prhs[1], fn), out = alloc_complex(nc);
  need(xp.n >= loc.size(), fn, "Index exceeds the number of array elements (Xp).");
  check(ofdm_LS_CE(y.ptr(), (int)nfft, (int64_t)ns, xp.ptr(), loc.data(), (int)loc.size(), nc, out.ptr(), flags()), fn);
  plhs[0] = put_complex(out, 1, nc);
}
