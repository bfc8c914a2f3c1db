// MEX gateway for demapping -- replaces Task 5/demapping.m:1-25
// MATLAB signature kept verbatim: de_bits = demapping(pad, IQ, Constellation)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "demapping";
  (void)nlhs;
  need(nrhs == 3, fn, "three inputs expected");
  ensure_init();
  const int pad = (int)get_scalar(prhs[0], fn);
  const std::string name = get_string(prhs[2], fn);
  int bps = 0;
  check(ofdm_constellation_func(name.c_str(), nullptr, &bps, 0), fn);
  CBuf iq = get_complex(prhs[1], fn);
  const size_t nb = iq.n * bps - (pad != -1 ? (size_t)pad : 0);
  std::vector<uint8_t> bits(nb ? nb : 1);
  check(ofdm_demapping(pad, iq.ptr(), (int64_t)iq.n, name.c_str(), bits.data(), flags()), fn);
  plhs[0] = put_bits(bits, 1, nb);                                           // reshape(de_bits, 1, [])
}
