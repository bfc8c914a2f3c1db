// MEX gateway for DeScrambler -- replaces Task 5/DeScrambler.m:1-28
// MATLAB signature kept verbatim: [sequence_out, Register] = DeScrambler(Register, sequence)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "DeScrambler";
  (void)nlhs;
  need(nrhs == 2, fn, "two inputs expected");
  ensure_init();
  std::vector<uint8_t> reg = get_bits(prhs[0], fn), seq = get_bits(prhs[1], fn), out(seq.size() ? seq.size() : 1);
  need(reg.size() == 15, fn, "Register must have 15 elements");
  check(ofdm_DeScrambler(reg.data(), seq.data(), (int64_t)seq.size(), out.data(), flags()), fn);
  plhs[0] = put_bits(out, mxGetM(prhs[1]), mxGetN(prhs[1]));                 // zeros(size(sequence))
  if (nlhs > 1) plhs[1] = put_bits(reg, mxGetM(prhs[0]), mxGetN(prhs[0]));
}
