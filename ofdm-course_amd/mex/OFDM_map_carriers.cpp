// MEX gateway for OFDM_map_carriers -- replaces Task 5/OFDM_map_carriers.m:2-9
// MATLAB signature kept verbatim: mapped_carriers = OFDM_map_carriers(QAM_payload, N_symb, Nfft, dataCarriers, pilotCarriers, pilotValues)
#include "ofdm_mex_common.hpp"
using namespace ofdm_mex;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const char* fn = "OFDM_map_carriers";
  (void)nlhs;
  need(nrhs == 6, fn, "six inputs expected");
  ensure_init();
  const int64_t ns = (int64_t)get_scalar(prhs[1], fn);
  const int nfft = (int)get_scalar(prhs[2], fn);
  std::vector<int32_t> dc = get_index(prhs[3], fn), pc = get_index(prhs[4], fn);
  CBuf pay = get_complex(prhs[0], fn), pv = get_complex(prhs[5], fn), out = alloc_complex((size_t)nfft * ns);
  need(pay.n == dc.size() * (size_t)ns, fn, "To RESHAPE the number of elements must not change.");
  const int scalar = pv.n == 1;                 // T3/Main_model_Task_3.m:59 passes a scalar amplitude
  need(scalar || pv.n == pc.size() * (size_t)ns, fn, "Unable to perform assignment: pilotValues size mismatch");
  check(ofdm_OFDM_map_carriers(pay.ptr(), ns, nfft, dc.data(), (int)dc.size(), pc.data(), (int)pc.size(), pv.ptr(),
                               scalar, out.ptr(), flags()), fn);
  plhs[0] = put_complex(out, nfft, ns);
}
