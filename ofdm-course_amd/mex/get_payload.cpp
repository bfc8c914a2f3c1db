// MEX gateway for get_payload -- replaces Task 5/get_payload.m:2-4
// MATLAB signature kept verbatim: RX_IQ = get_payload(RX_OFDM_symbols, dataCarriers)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "get_payload";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  const size_t nfft = mxGetM(prhs[0]), ns = mxGetN(prhs[0]);
  std::vector<int32_t> dc = get_index(prhs[1], fn);
  CBuf x = get_complex(prhs[0], fn), out = alloc_complex(dc.size() * ns);
  check(ofdm_get_payload(x.ptr(), (int)nfft, (int64_t)ns, dc.data(), (int)dc.size(), out.ptr(), flags()), fn);
  plhs[0] = put_complex(out, dc.size(), ns);
}
