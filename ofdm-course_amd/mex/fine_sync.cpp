// MEX gateway for fine_sync -- replaces Task 5/fine_sync.m:1-45
// MATLAB signature kept verbatim: sync_signal = fine_sync(rx_signal, pilotCarriers, pilotValues, time_desync, freq_desync)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "fine_sync";
  (void)nlhs;
  need(nrhs == 5, fn, "five inputs expected");
  ensure_init();
  const size_t nfft = mxGetM(prhs[0]), ns = mxGetN(prhs[0]);
  std::vector<int32_t> pc = get_index(prhs[1], fn);
  CBuf x = get_complex(prhs[0], fn), pv = get_complex(prhs[2], fn), y = alloc_complex(nfft * ns);
  need(pv.n == pc.size() * ns, fn, "pilotValues must be [numel(pilotCarriers) x N_symb]");
  // variant 0 = Task 5 file; set OFDM_MEX_FINE_SYNC_T4=1 for the Task 4 variant (extra diff~=0 mask)
  const int variant = std::getenv("OFDM_MEX_FINE_SYNC_T4") ? 1 : 0;
  check(ofdm_fine_sync(x.ptr(), (int)nfft, (int64_t)ns, pc.data(), (int)pc.size(), pv.ptr(),
                       get_scalar(prhs[3], fn) != 0, get_scalar(prhs[4], fn) != 0, variant, y.ptr(), nullptr, nullptr, flags()), fn);
  plhs[0] = put_complex(y, nfft, ns);
}
