// MEX gateway for mapping -- replaces Task 5/mapping.m:1-25
// MATLAB signature kept verbatim: [IQ, pad] = mapping(bits, constellation)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "mapping";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  const std::string name = get_string(prhs[1], fn);
  int bps = 0;
  check(ofdm_constellation_func(name.c_str(), nullptr, &bps, 0), fn);
  std::vector<uint8_t> bits = get_bits(prhs[0], fn);
  // mapping.m:11 vertcat(bits, zeros(pad,1)) is a dimension error for a row vector that needs padding
  need(bits.size() % bps == 0 || mxGetN(prhs[0]) == 1, fn, "Dimensions of arrays being concatenated are not consistent.");
  const size_t n_iq = (bits.size() + bps - 1) / bps;
  CBuf iq = alloc_complex(n_iq);
  int pad = -1;
  check(ofdm_mapping(bits.data(), (int64_t)bits.size(), name.c_str(), iq.ptr(), &pad, flags()), fn);
  plhs[0] = put_complex(iq, 1, n_iq);                                        // dictionary(idx) keeps the row orientation
  if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)pad);
}
