// MEX gateway for BER_func -- replaces Task 5/BER_func.m:1-7
// MATLAB signature kept verbatim: BER = BER_func(Bit_Tx, Bit_Rx)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "BER_func";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  std::vector<uint8_t> a = get_bits(prhs[0], fn), b = get_bits(prhs[1], fn);
  need(a.size() == b.size(), fn, "Arrays have incompatible sizes for this operation.");
  int64_t ne = 0;
  check(ofdm_BER_func(a.data(), b.data(), (int64_t)a.size(), &ne, flags()), fn);
  plhs[0] = mxCreateDoubleScalar((double)ne / (double)a.size());
}
