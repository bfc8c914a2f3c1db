// Shared helpers of the MEX gateways (one gateway per reference .m function, same name, so that a
// compiled <name>.mex* on the MATLAB path shadows <name>.m and the untouched Main_model_Task_*.m
// drivers call straight into libofdm_mi355x).  Build (on a host with MATLAB + ROCm):
//     mex -R2018a -I<repo>/include <name>.cpp -L<repo>/ofdm-course_amd -lofdm_mi355x
// Interleaved-complex API (-R2018a): mxGetComplexDoubles gives {real, imag} pairs = ofdm_c64 layout.
// NOT compiled in this repository's CI: neither MATLAB nor mex.h exist there;
// the gateways are exercised indirectly through the same C ABI by the Python host mirror.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mex.h"
#include "ofdm_mi355x.h"

namespace ofdm_mex {

struct c64 { double re, im; };
struct c32 { float re, im; };

inline void at_exit() { ofdm_shutdown(); }

inline void ensure_init() {
  static bool done = false;
  if (!done) {
    const char* dev = std::getenv("OFDM_MEX_DEVICE");
    if (ofdm_init(dev ? std::atoi(dev) : 0) != OFDM_OK)
      mexErrMsgIdAndTxt("ofdm:init:failed", "%s", ofdm_last_error_string());
    mexAtExit(at_exit);
    done = true;
  }
}

// parity mode (double, the default) or throughput mode (OFDM_MEX_PRECISION=fp32: convert on the host)
inline bool use_f32() {
  const char* p = std::getenv("OFDM_MEX_PRECISION");
  return p && std::string(p) == "fp32";
}
inline int flags() { return (use_f32() ? OFDM_F32 : OFDM_F64) | OFDM_HOST; }

// negative -> MATLAB error "ofdm:<fn>:<reason>", positive soft codes are handled by the caller
inline void check(int rc, const char* fn) {
  if (rc < 0) {
    const std::string id = std::string("ofdm:") + fn + (rc == OFDM_ERR_ARG ? ":badArgument" : rc == OFDM_ERR_HIP ? ":hip" : ":failed");
    mexErrMsgIdAndTxt(id.c_str(), "%s", ofdm_last_error_string());
  }
}

inline void need(bool cond, const char* fn, const char* msg) {
  if (!cond) mexErrMsgIdAndTxt((std::string("ofdm:") + fn + ":badArgument").c_str(), "%s", msg);
}

// ---- complex data: any numeric double array (real or complex) -> interleaved buffer in the working precision
struct CBuf {
  std::vector<c64> d;
  std::vector<c32> f;
  size_t n = 0;
  const void* ptr() const { return use_f32() ? (const void*)f.data() : (const void*)d.data(); }
  void* ptr() { return use_f32() ? (void*)f.data() : (void*)d.data(); }
};

inline CBuf get_complex(const mxArray* a, const char* fn) {
  need(mxIsDouble(a) || mxIsLogical(a), fn, "numeric double input expected");
  CBuf b;
  b.n = mxGetNumberOfElements(a);
  const bool f32 = use_f32();
  if (f32) b.f.resize(b.n); else b.d.resize(b.n);
  if (mxIsLogical(a)) {
    const mxLogical* p = mxGetLogicals(a);
    for (size_t i = 0; i < b.n; ++i) { if (f32) b.f[i] = {(float)p[i], 0.f}; else b.d[i] = {(double)p[i], 0.0}; }
  } else if (mxIsComplex(a)) {
    const mxComplexDouble* p = mxGetComplexDoubles(a);
    for (size_t i = 0; i < b.n; ++i) { if (f32) b.f[i] = {(float)p[i].real, (float)p[i].imag}; else b.d[i] = {p[i].real, p[i].imag}; }
  } else {
    const double* p = mxGetDoubles(a);
    for (size_t i = 0; i < b.n; ++i) { if (f32) b.f[i] = {(float)p[i], 0.f}; else b.d[i] = {p[i], 0.0}; }
  }
  return b;
}

inline CBuf alloc_complex(size_t n) {
  CBuf b;
  b.n = n;
  if (use_f32()) b.f.resize(n); else b.d.resize(n);
  return b;
}

// working buffer -> new complex double mxArray of shape [m x n] (column-major, MATLAB orientation)
inline mxArray* put_complex(const CBuf& b, size_t m, size_t n) {
  mxArray* a = mxCreateDoubleMatrix(m, n, mxCOMPLEX);
  mxComplexDouble* p = mxGetComplexDoubles(a);
  if (use_f32()) for (size_t i = 0; i < m * n; ++i) { p[i].real = b.f[i].re; p[i].imag = b.f[i].im; }
  else std::memcpy(p, b.d.data(), sizeof(c64) * m * n);
  return a;
}

// ---- bit vectors: MATLAB doubles / logicals 0/1 -> uint8
inline std::vector<uint8_t> get_bits(const mxArray* a, const char* fn) {
  const size_t n = mxGetNumberOfElements(a);
  std::vector<uint8_t> v(n);
  if (mxIsLogical(a)) { const mxLogical* p = mxGetLogicals(a); for (size_t i = 0; i < n; ++i) v[i] = p[i] ? 1 : 0; }
  else { need(mxIsDouble(a) && !mxIsComplex(a), fn, "bit vector must be real double or logical");
         const double* p = mxGetDoubles(a); for (size_t i = 0; i < n; ++i) v[i] = p[i] != 0.0; }
  return v;
}
inline mxArray* put_bits(const std::vector<uint8_t>& v, size_t m, size_t n) {
  mxArray* a = mxCreateDoubleMatrix(m, n, mxREAL);
  double* p = mxGetDoubles(a);
  for (size_t i = 0; i < m * n; ++i) p[i] = v[i];
  return a;
}

// ---- 1-based index vectors (doubles from linspace / colon) -> int32
inline std::vector<int32_t> get_index(const mxArray* a, const char* fn) {
  need(mxIsDouble(a) && !mxIsComplex(a), fn, "index vector must be real double");
  const size_t n = mxGetNumberOfElements(a);
  const double* p = mxGetDoubles(a);
  std::vector<int32_t> v(n);
  for (size_t i = 0; i < n; ++i) {
    need(p[i] == (double)(int32_t)p[i], fn, "Subscript indices must be integers");
    v[i] = (int32_t)p[i];
  }
  return v;
}

// ---- Constellation: char vector or string scalar ("16QAM" in T5/Main_model_Task_5.m:38 is a string)
inline std::string get_string(const mxArray* a, const char* fn) {
  if (mxIsChar(a)) { char* s = mxArrayToString(a); std::string r(s ? s : ""); mxFree(s); return r; }
  if (mxIsClass(a, "string")) {
    mxArray* out = nullptr;
    mxArray* in = const_cast<mxArray*>(a);
    need(mexCallMATLAB(1, &out, 1, &in, "char") == 0, fn, "cannot convert string to char");
    char* s = mxArrayToString(out);
    std::string r(s ? s : "");
    mxFree(s);
    mxDestroyArray(out);
    return r;
  }
  need(false, fn, "Constellation must be a char vector or a string scalar");
  return "";
}

inline double get_scalar(const mxArray* a, const char* fn) {
  need(mxGetNumberOfElements(a) == 1, fn, "scalar expected");
  return mxGetScalar(a);
}

}  // namespace ofdm_mex
