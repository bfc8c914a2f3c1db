// MEX gateway for interpolate -- replaces Task 5/interpolate.m:1-24
// MATLAB signature kept verbatim: H_interpolated = interpolate(H, pilot_loc, Nfft, method)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "interpolate";
  (void)nlhs;
  need(nrhs == 4, fn, "four inputs expected");
  ensure_init();
  std::vector<int32_t> loc = get_index(prhs[1], fn);
  const int n = (int)get_scalar(prhs[2], fn);
  const std::string method = get_string(prhs[3], fn);
  CBuf h = get_complex(prhs[0], fn), out = alloc_complex(n);
  need(h.n == loc.size() && !method.empty(), fn, "H and pilot_loc must have the same length");
  check(ofdm_interpolate(h.ptr(), loc.data(), (int)loc.size(), n, method[0], out.ptr(), flags()), fn);
  plhs[0] = put_complex(out, 1, n);
}
