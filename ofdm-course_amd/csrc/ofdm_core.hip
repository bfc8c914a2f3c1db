// Library lifecycle, error state, stream, twiddle cache, host<->HBM staging.
#include <cstdarg>

#include "ofdm_common.hpp"

namespace ofdm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

Context& ctx() {
  static Context c;
  return c;
}

int ensure_init() {
  Context& c = ctx();
  if (c.ready) return OFDM_OK;
  return ofdm_init(-1);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("kernel launch %s failed: %s", what, hipGetErrorString(e));
    return OFDM_ERR_HIP;
  }
  return OFDM_OK;
}

int pool_get(size_t bytes, void** out) {
  Context& c = ctx();
  if (bytes == 0) bytes = 16;
  std::lock_guard<std::mutex> lk(c.mu);
  int best = -1;
  for (size_t i = 0; i < c.pool.size(); ++i) {
    if (c.pool[i].busy || c.pool[i].bytes < bytes) continue;
    if (c.pool[i].pending) {                                   // handed back while its last kernels were still queued
      if (hipEventQuery(c.pool[i].ev) != hipSuccess) continue;
      c.pool[i].pending = false;
    }
    if (best < 0 || c.pool[i].bytes < c.pool[best].bytes) best = (int)i;
  }
  if (best >= 0 && c.pool[best].bytes <= 2 * bytes + 4096) {
    c.pool[best].busy = true;
    *out = c.pool[best].p;
    return OFDM_OK;
  }
  void* p = nullptr;
  size_t rounded = (bytes + 255) & ~size_t(255);
  hipError_t e = hipMalloc(&p, rounded);
  if (e != hipSuccess) {
    // drop idle blocks and retry once
    for (auto& b : c.pool)
      if (!b.busy && b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    e = hipMalloc(&p, rounded);
    if (e != hipSuccess) {
      set_error("hipMalloc(%zu) failed: %s", rounded, hipGetErrorString(e));
      return OFDM_ERR_HIP;
    }
  }
  c.pool.push_back({p, rounded, true});
  *out = p;
  return OFDM_OK;
}

void pool_put(void* p) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  for (auto& b : c.pool)
    if (b.p == p) { b.busy = false; return; }
}

void pool_put_after_stream(void* p) {
  Context& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  for (auto& b : c.pool)
    if (b.p == p) {
      if (!b.ev && hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) b.ev = nullptr;
      if (b.ev && hipEventRecord(b.ev, c.stream) == hipSuccess) {
        b.pending = true;
      } else {
        (void)hipStreamSynchronize(c.stream);                      // no event: fall back to waiting
        b.pending = false;
      }
      b.busy = false;
      return;
    }
}

int get_twiddles(int n, bool f64, const void** out) {
  Context& c = ctx();
  int64_t key = ((int64_t)n << 1) | (f64 ? 1 : 0);
  {
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.twiddles.find(key);
    if (it != c.twiddles.end()) { *out = it->second; return OFDM_OK; }
  }
  // exact-quadrant evaluation in long double, then rounded once to the target type
  std::vector<c64> tw(n);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int m = 0; m < n; ++m) {
    // reduce to the first octant for accuracy
    long double a = two_pi * (long double)m / (long double)n;
    tw[m].x = (double)cosl(a);
    tw[m].y = (double)(-sinl(a));
  }
  // force exact values on the axes
  tw[0].x = 1.0; tw[0].y = 0.0;
  if (n % 4 == 0) {
    tw[n / 4].x = 0.0; tw[n / 4].y = -1.0;
    tw[n / 2].x = -1.0; tw[n / 2].y = 0.0;
    tw[3 * n / 4].x = 0.0; tw[3 * n / 4].y = 1.0;
  } else if (n % 2 == 0) {
    tw[n / 2].x = -1.0; tw[n / 2].y = 0.0;
  }
  void* d = nullptr;
  if (f64) {
    OFDM_HIP(hipMalloc(&d, sizeof(c64) * n));
    OFDM_HIP(hipMemcpy(d, tw.data(), sizeof(c64) * n, hipMemcpyHostToDevice));
  } else {
    std::vector<c32> t32(n);
    for (int m = 0; m < n; ++m) { t32[m].x = (float)tw[m].x; t32[m].y = (float)tw[m].y; }
    OFDM_HIP(hipMalloc(&d, sizeof(c32) * n));
    OFDM_HIP(hipMemcpy(d, t32.data(), sizeof(c32) * n, hipMemcpyHostToDevice));
  }
  std::lock_guard<std::mutex> lk(c.mu);
  c.twiddles[key] = d;
  *out = d;
  return OFDM_OK;
}

// ---------------------------------------------------------------------------------------------
// Stage
// ---------------------------------------------------------------------------------------------
int Stage::in(const void* p, size_t bytes, const void** d) {
  if (dev_ || p == nullptr) { *d = p; return OFDM_OK; }
  void* b = nullptr;
  OFDM_TRY(pool_get(bytes, &b));
  bufs_.push_back(b);
  if (bytes) OFDM_HIP(hipMemcpyAsync(b, p, bytes, hipMemcpyHostToDevice, ctx().stream));
  *d = b;
  return OFDM_OK;
}

int Stage::out(void* p, size_t bytes, void** d) {
  if (dev_ || p == nullptr) { *d = p; return OFDM_OK; }
  void* b = nullptr;
  OFDM_TRY(pool_get(bytes, &b));
  bufs_.push_back(b);
  outs_.push_back({p, b, bytes});
  *d = b;
  return OFDM_OK;
}

int Stage::upload(const void* host, size_t bytes, const void** d) {
  if (host == nullptr) { *d = nullptr; return OFDM_OK; }
  void* b = nullptr;
  OFDM_TRY(pool_get(bytes, &b));
  bufs_.push_back(b);
  // the host buffer may be a temporary of the caller: copy synchronously w.r.t. the host
  if (bytes) OFDM_HIP(hipMemcpyAsync(b, host, bytes, hipMemcpyHostToDevice, ctx().stream));
  if (bytes) OFDM_HIP(hipStreamSynchronize(ctx().stream));
  *d = b;
  return OFDM_OK;
}

int Stage::scratch(size_t bytes, void** d) {
  void* b = nullptr;
  OFDM_TRY(pool_get(bytes, &b));
  bufs_.push_back(b);
  *d = b;
  return OFDM_OK;
}

int Stage::fetch(void* host, size_t bytes, void** d) {
  void* b = nullptr;
  OFDM_TRY(pool_get(bytes, &b));
  bufs_.push_back(b);
  if (host) outs_.push_back({host, b, bytes});
  *d = b;
  return OFDM_OK;
}

int Stage::finish() {
  for (auto& o : outs_)
    if (o.bytes) OFDM_HIP(hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, ctx().stream));
  if (!outs_.empty() || !dev_) OFDM_HIP(hipStreamSynchronize(ctx().stream));
  outs_.clear();
  return OFDM_OK;
}

void Stage::release() {
  if (!bufs_.empty()) {
    // device flavour: the call's kernels may still be queued on the stream -- the blocks go back with an event recorded
    // behind them and become reusable when it has completed; the call does not wait (host flavour: finish() has synchronised)
    for (void* b : bufs_) {
      if (dev_) pool_put_after_stream(b); else pool_put(b);
    }
    bufs_.clear();
  }
}

// ---------------------------------------------------------------------------------------------
// constellation tables -- T5/constellation_func.m:4-35 (+ 64QAM/256QAM extension)
// ---------------------------------------------------------------------------------------------
static int gray_decode(int g) {
  int b = 0;
  while (g) { b ^= g; g >>= 1; }
  return b;
}

int constellation_info(const char* name, ConstellationInfo& info) {
  std::string n = name ? name : "";
  info = ConstellationInfo();
  if (n == "BPSK") { info.bps = 1; info.kind = 0; }
  else if (n == "QPSK") { info.bps = 2; info.kind = 0; }
  else if (n == "8PSK") { info.bps = 3; info.kind = 0; }
  else if (n == "16QAM") { info.bps = 4; info.kind = 1; info.bits_per_axis = 2; }
  else if (n == "64QAM") { info.bps = 6; info.kind = 1; info.bits_per_axis = 3; }
  else if (n == "256QAM") { info.bps = 8; info.kind = 1; info.bits_per_axis = 4; }
  else return 0;
  return info.bps;
}

int constellation_table(const char* name, std::vector<c64>& dict) {
  ConstellationInfo info;
  if (!constellation_info(name, info)) return 0;
  std::string n = name;
  int M = 1 << info.bps;
  dict.assign(M, c64{0, 0});
  if (n == "BPSK") {
    dict[0] = {-1, 0}; dict[1] = {1, 0};
  } else if (n == "QPSK") {
    dict[0] = {-1, -1}; dict[1] = {-1, 1}; dict[2] = {1, -1}; dict[3] = {1, 1};
  } else if (n == "8PSK") {
    static const int gray_map[8] = {5, 4, 2, 3, 6, 7, 1, 0};       // constellation_func.m:13
    for (int i = 0; i < 8; ++i) {
      double a = (double)gray_map[i] * 2.0 * M_PI / 8.0;           // :14
      dict[i] = {std::cos(a), std::sin(a)};
    }
  } else {
    int ba = info.bits_per_axis, L = 1 << ba;
    for (int idx = 0; idx < M; ++idx) {
      int ci = idx >> ba, cq = idx & (L - 1);
      double il = 2.0 * gray_decode(ci) - (L - 1);
      double ql = -(2.0 * gray_decode(cq) - (L - 1));
      dict[idx] = {il, ql};
    }
  }
  double s = 0;
  for (auto& d : dict) s += d.x * d.x + d.y * d.y;                 // :28
  double norm = std::sqrt(s / M);
  for (auto& d : dict) { d.x /= norm; d.y /= norm; }               // :29
  return info.bps;
}

}  // namespace ofdm

using namespace ofdm;

extern "C" {

int ofdm_version(void) { return 100; }

const char* ofdm_last_error_string(void) { return get_error(); }

int ofdm_init(int device_id) {
  Context& c = ctx();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("ofdm_init: no HIP device visible (%s) -- this library has no CPU fallback",
              e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    return OFDM_ERR_STATE;
  }
  if (device_id < 0) {
    int cur = 0;
    if (c.ready) return OFDM_OK;
    (void)hipGetDevice(&cur);
    device_id = cur;
  }
  OFDM_ARG(device_id < n, "ofdm_init: device %d out of range (%d devices)", device_id, n);
  if (c.ready && c.device == device_id) return OFDM_OK;
  if (c.ready && c.live_plans > 0) {
    // one device per process: live plans hold tables, workspaces and twiddles on c.device -- re-initialising under them
    // would free the caches they launch with and aim their launches at another GPU
    set_error("ofdm_init: the context belongs to device %d with %d live RX plan(s); destroy them (or call ofdm_shutdown) "
              "before switching to device %d -- one device per process", c.device, c.live_plans, device_id);
    return OFDM_ERR_STATE;
  }
  if (c.ready) ofdm_shutdown();
  OFDM_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  OFDM_HIP(hipGetDeviceProperties(&prop, device_id));
  c.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c.device = device_id;
  c.stream = nullptr;
  c.ready = true;
  return OFDM_OK;
}

int ofdm_shutdown(void) {
  Context& c = ctx();
  if (!c.ready) return OFDM_OK;
  (void)hipDeviceSynchronize();
  for (auto& kv : c.twiddles) (void)hipFree(kv.second);
  c.twiddles.clear();
  for (auto& b : c.pool) {
    if (b.p) (void)hipFree(b.p);
    if (b.ev) (void)hipEventDestroy(b.ev);
  }
  c.pool.clear();
  c.ready = false;
  c.device = -1;
  c.live_plans = 0;
  return OFDM_OK;
}

int ofdm_set_stream(void* hip_stream) {
  OFDM_TRY(ensure_init());
  ctx().stream = (hipStream_t)hip_stream;
  return OFDM_OK;
}

int ofdm_synchronize(void) {
  OFDM_TRY(ensure_init());
  OFDM_HIP(hipStreamSynchronize(ctx().stream));
  return OFDM_OK;
}

int ofdm_constellation_func(const char* name, void* dict_out, int* bps_out, int flags) {
  std::vector<c64> d;
  int bps = constellation_table(name, d);
  OFDM_ARG(bps > 0, "constellation_func: unknown constellation '%s'", name ? name : "(null)");
  if (bps_out) *bps_out = bps;
  if (dict_out) {
    if (is_f64(flags)) {
      memcpy(dict_out, d.data(), sizeof(c64) * d.size());
    } else {
      c32* o = (c32*)dict_out;
      for (size_t i = 0; i < d.size(); ++i) { o[i].x = (float)d[i].x; o[i].y = (float)d[i].y; }
    }
  }
  return OFDM_OK;
}

}  // extern "C"
