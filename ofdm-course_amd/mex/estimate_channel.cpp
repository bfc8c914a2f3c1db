// MEX gateway for estimate_channel -- replaces Task 5/estimate_channel.m:1-9
// MATLAB signature kept verbatim: [H_est, Hest_at_pilots] = estimate_channel(rx_signal, allCarriers, pilotCarriers, pilotValues)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "estimate_channel";
  (void)nlhs;
  need(nrhs == 4, fn, "four inputs expected");
  ensure_init();
  const size_t nfft = mxGetM(prhs[0]), ns = mxGetN(prhs[0]);
  std::vector<int32_t> all = get_index(prhs[1], fn), pc = get_index(prhs[2], fn);
  CBuf x = get_complex(prhs[0], fn), pv = get_complex(prhs[3], fn);
  CBuf h = alloc_complex(all.size()), hp = alloc_complex(pc.size());
  check(ofdm_estimate_channel(x.ptr(), (int)nfft, (int64_t)ns, all.data(), (int)all.size(), pc.data(), (int)pc.size(),
                              pv.ptr(), h.ptr(), hp.ptr(), flags()), fn);
  plhs[0] = put_complex(h, mxGetM(prhs[1]), mxGetN(prhs[1]));                 // interp1 returns the query's orientation
  if (nlhs > 1) plhs[1] = put_complex(hp, pc.size(), 1);
}
