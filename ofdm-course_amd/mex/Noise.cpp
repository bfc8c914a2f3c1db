// MEX gateway for Noise -- replaces Task 5/Noise.m:1-12
// MATLAB signature kept verbatim: [IQ_RX, N_var] = Noise(SNR, IQ_TX)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "Noise";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  // normrnd draws are replaced by the library's counter-based generator; the stream id advances per call so
  // successive calls are independent (MATLAB's rng(1) sequence itself is not reproducible outside MATLAB).
  static uint32_t stream = 0;
  const char* seed_env = std::getenv("OFDM_MEX_SEED");
  const uint64_t seed = seed_env ? std::strtoull(seed_env, nullptr, 10) : 1;
  CBuf x = get_complex(prhs[1], fn), y = alloc_complex(x.n);
  double nvar = 0;
  check(ofdm_Noise(get_scalar(prhs[0], fn), x.ptr(), (int64_t)x.n, seed, stream++, y.ptr(), &nvar, flags()), fn);
  plhs[0] = put_complex(y, mxGetM(prhs[1]), mxGetN(prhs[1]));
  if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(nvar);
}
