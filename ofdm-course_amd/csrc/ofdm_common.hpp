// Shared host/device plumbing of libofdm_mi355x: complex type, error state, stream,
// host<->HBM staging for the host-pointer (MEX) flavour of every C-ABI entry.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ofdm_mi355x.h"

namespace ofdm {

// ---------------------------------------------------------------------------------------------
// complex number, interleaved (re, im) -- the layout of MATLAB's interleaved-complex API
// ---------------------------------------------------------------------------------------------
template <typename T>
struct alignas(2 * sizeof(T)) cx {
  T x, y;
};
using c32 = cx<float>;
using c64 = cx<double>;

template <typename T> __host__ __device__ inline cx<T> mk(T a, T b) { cx<T> r; r.x = a; r.y = b; return r; }
template <typename T> __host__ __device__ inline cx<T> operator+(cx<T> a, cx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> __host__ __device__ inline cx<T> operator-(cx<T> a, cx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> __host__ __device__ inline cx<T> operator*(cx<T> a, cx<T> b) {
  return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <typename T> __host__ __device__ inline cx<T> operator*(cx<T> a, T s) { return mk<T>(a.x * s, a.y * s); }
template <typename T> __host__ __device__ inline cx<T> conj(cx<T> a) { return mk<T>(a.x, -a.y); }
// a * conj(b)
template <typename T> __host__ __device__ inline cx<T> mulc(cx<T> a, cx<T> b) {
  return mk<T>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
template <typename T> __host__ __device__ inline T norm2(cx<T> a) { return a.x * a.x + a.y * a.y; }
// a / b  (plain formula; |b| is O(1) everywhere on this path, tolerance documented in DESIGN.md)
template <typename T> __host__ __device__ inline cx<T> cdiv(cx<T> a, cx<T> b) {
  T d = T(1) / (b.x * b.x + b.y * b.y);
  return mk<T>((a.x * b.x + a.y * b.y) * d, (a.y * b.x - a.x * b.y) * d);
}
// c -= g * x and c += a * b as four fused multiply-adds each (the operator forms cost six: product, then add)
template <typename T>
__device__ __forceinline__ void cmsub(cx<T>& c, cx<T> g, cx<T> x) {
  c.x = fma(-g.x, x.x, c.x); c.x = fma(g.y, x.y, c.x);
  c.y = fma(-g.x, x.y, c.y); c.y = fma(-g.y, x.x, c.y);
}
template <typename T>
__device__ __forceinline__ void cmadd(cx<T>& c, cx<T> a, cx<T> b) {
  c.x = fma(a.x, b.x, c.x); c.x = fma(-a.y, b.y, c.x);
  c.y = fma(a.x, b.y, c.y); c.y = fma(a.y, b.x, c.y);
}
template <typename T>
__device__ __forceinline__ void cmsubc(cx<T>& c, cx<T> a, cx<T> b) {      // c -= a * conj(b)
  c.x = fma(-a.x, b.x, c.x); c.x = fma(-a.y, b.y, c.x);
  c.y = fma(-a.y, b.x, c.y); c.y = fma(a.x, b.y, c.y);
}
template <typename T>
__device__ __forceinline__ void cmaddc(cx<T>& c, cx<T> a, cx<T> b) {      // c += a * conj(b)
  c.x = fma(a.x, b.x, c.x); c.x = fma(a.y, b.y, c.x);
  c.y = fma(a.y, b.x, c.y); c.y = fma(-a.x, b.y, c.y);
}

// Load through the scalar cache: for a wavefront-uniform address into memory that no kernel of the launch writes (the
// twiddle tables).  hipcc only selects s_load for the constant address space, hence the cast.
template <typename T>
__device__ __forceinline__ cx<T> uniform_load(const cx<T>* p) {
  typedef T v2 __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(4))) v2* cptr;
  const v2 q = *(cptr)(uintptr_t)p;
  return mk<T>(q.x, q.y);
}
// multiply by -i (forward) / +i (inverse)
template <typename T, bool INV> __host__ __device__ inline cx<T> mul_mi(cx<T> a) {
  return INV ? mk<T>(-a.y, a.x) : mk<T>(a.y, -a.x);
}

// Streaming read of a sample that no later instruction of the launch touches again: nontemporal (`global_load ... nt`).
// The received samples are 90 % of the chain's bytes and are read exactly once; with the default cache policy the symbol
// kernel ran at 280 M symbols/s, with nt loads at 287 M (same box, three alternating pairs) -- MI355X_MICROARCH.md quotes
// the same 3-5 % between default-policy and nt streams.  Neighbouring nt loads still merge into 16-byte instructions.
template <typename T>
__device__ __forceinline__ cx<T> nt_load(const cx<T>* __restrict__ p) {
  typedef T v2 __attribute__((ext_vector_type(2)));
  const v2 q = __builtin_nontemporal_load(reinterpret_cast<const v2*>(p));
  return mk<T>(q.x, q.y);
}

template <typename T>
__device__ __forceinline__ void nt_store(cx<T>* __restrict__ p, cx<T> v) {
  typedef T v2 __attribute__((ext_vector_type(2)));
  v2 q;
  q.x = v.x; q.y = v.y;
  __builtin_nontemporal_store(q, reinterpret_cast<v2*>(p));
}

// ---------------------------------------------------------------------------------------------
// error state
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
const char* get_error();

#define OFDM_HIP(call)                                                                         \
  do {                                                                                         \
    hipError_t _e = (call);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      ::ofdm::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return OFDM_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

#define OFDM_ARG(cond, ...)                 \
  do {                                      \
    if (!(cond)) {                          \
      ::ofdm::set_error(__VA_ARGS__);       \
      return OFDM_ERR_ARG;                  \
    }                                       \
  } while (0)

#define OFDM_TRY(expr)            \
  do {                            \
    int _rc = (expr);             \
    if (_rc < 0) return _rc;      \
  } while (0)

// ---------------------------------------------------------------------------------------------
// context: device, stream, caches
// ---------------------------------------------------------------------------------------------
struct Context {
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  int num_cu = 256;
  int live_plans = 0;      // RX plans created on `device` and not yet destroyed (they pin the context to that device)
  // twiddle tables keyed by (nfft << 1 | is_f64): tw[m] = exp(-2*pi*i*m/nfft), m < nfft
  std::map<int64_t, void*> twiddles;
  // scratch pool for host staging (freed at shutdown)
  // ev: recorded on the launch stream when a device-flavour call hands the block back (its kernels may still be
  // running); the block is reusable once the event has completed -- the call itself does not wait
  struct Block { void* p; size_t bytes; bool busy; hipEvent_t ev = nullptr; bool pending = false; };
  std::vector<Block> pool;
  std::mutex mu;
};
Context& ctx();
int ensure_init();
int pool_get(size_t bytes, void** out);
void pool_put(void* p);
void pool_put_after_stream(void* p);   // hand back a block that work already queued on ctx().stream still uses
// returns device pointer to exp(-2*pi*i*m/n) table (m<n) in the requested precision
int get_twiddles(int n, bool f64, const void** out);

inline bool fft_size_supported(int n) { return n >= 64 && n <= 8192 && (n & (n - 1)) == 0; }
inline bool is_f64(int flags) { return (flags & OFDM_F64) != 0; }
inline bool is_dev(int flags) { return (flags & OFDM_DEVICE) != 0; }
inline size_t csize(int flags) { return is_f64(flags) ? 16 : 8; }
inline size_t rsize(int flags) { return is_f64(flags) ? 8 : 4; }

// ---------------------------------------------------------------------------------------------
// Stage: maps the caller's pointers to device pointers.  OFDM_DEVICE -> identity, async.
// OFDM_HOST -> pool buffer + H2D for inputs, D2H + sync for outputs in finish().
// Index tables / small host-only parameters always go through upload().
// ---------------------------------------------------------------------------------------------
class Stage {
 public:
  explicit Stage(int flags) : dev_(is_dev(flags)) {}
  ~Stage() { release(); }
  // input living where `flags` says
  int in(const void* p, size_t bytes, const void** d);
  // output living where `flags` says
  int out(void* p, size_t bytes, void** d);
  // host-resident parameter that must be copied to the device whatever the flags
  int upload(const void* host, size_t bytes, const void** d);
  // device scratch
  int scratch(size_t bytes, void** d);
  // host-resident result that must be fetched whatever the flags (scalars); fetched in finish()
  int fetch(void* host, size_t bytes, void** d);
  // copies outputs back (host mode), synchronises when anything must reach the host
  int finish();
  bool device_mode() const { return dev_; }

 private:
  void release();
  struct Out { void* host; void* dev; size_t bytes; };
  bool dev_;
  std::vector<void*> bufs_;
  std::vector<Out> outs_;
};

// launch helpers
inline unsigned cdiv_u(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }
int check_launch(const char* what);

// constellation tables (host side), shared by bits/chain code.
// Returns bps or 0 if the name is unknown.  dict holds 2^bps points in double precision.
int constellation_table(const char* name, std::vector<c64>& dict);
// square-QAM description for the per-axis slicer: bits per axis (0 for non-square), scale = 1/norm
struct ConstellationInfo {
  int bps = 0;
  int kind = 0;            // 0 = table search (BPSK/QPSK/8PSK), 1 = square QAM slicer
  int bits_per_axis = 0;
  double inv_norm = 1.0;   // table = levels * inv_norm
};
int constellation_info(const char* name, ConstellationInfo& info);

}  // namespace ofdm
