// Bit-exact integer side of the path: mapping, demapping, Scrambler/DeScrambler, BER_func, MER_func.
#include "demap_core.hpp"

namespace ofdm {

// ---------------------------------------------------------------------------------------------
// mapping -- T5/mapping.m:1-25: groups of bps bits (first bit = MSB), zero padded, table lookup.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void mapping_kernel(const uint8_t* __restrict__ bits, int64_t n_bits, int64_t n_iq,
                               DemapTable<T> tab, cx<T>* __restrict__ iq) {
  const int bps = tab.bps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_iq;
       i += (int64_t)gridDim.x * blockDim.x) {
    int idx = 0;
    const int64_t b0 = i * bps;
    for (int b = 0; b < bps; ++b) {
      const int64_t p = b0 + b;
      const int bit = (p < n_bits) ? (bits[p] != 0) : 0;     // :11 zero padding
      idx = (idx << 1) | bit;                                  // :18 left-msb
    }
    iq[i] = demap_point(tab, idx);                             // :21
  }
}

// demapping -- T5/demapping.m:1-25
template <typename T>
__global__ void demapping_kernel(const cx<T>* __restrict__ iq, int64_t n_iq, int64_t n_bits_out,
                                 DemapTable<T> tab, uint8_t* __restrict__ bits) {
  const int bps = tab.bps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_iq;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int idx = demap_decide(tab, iq[i]);
    const int64_t b0 = i * bps;
    for (int b = 0; b < bps; ++b) {
      const int64_t p = b0 + b;
      if (p < n_bits_out) bits[p] = (uint8_t)((idx >> (bps - 1 - b)) & 1);   // :15 MSB first, :21-23 pad strip
    }
  }
}

// ---------------------------------------------------------------------------------------------
// DeScrambler -- T5/DeScrambler.m:1-28: d[i] = s[i] ^ s[i-13] ^ s[i-14]; history before the frame
// comes from the register: R(j) = s[i-j].  Pure map: one thread per bit.
// ---------------------------------------------------------------------------------------------
__global__ void descramble_kernel(const uint8_t* __restrict__ seq, uint8_t* __restrict__ out,
                                  int64_t frame_len, int64_t n_frames, uint32_t reg_mask) {
  const int64_t total = frame_len * n_frames;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = g % frame_len;
    const uint8_t* s = seq + (g - i);
    // reg_mask bit (j-1) = R(j)
    const int a = (i >= 13) ? (s[i - 13] != 0) : (int)((reg_mask >> (13 - i - 1)) & 1u);
    const int b = (i >= 14) ? (s[i - 14] != 0) : (int)((reg_mask >> (14 - i - 1)) & 1u);
    out[g] = (uint8_t)((s[i] != 0) ^ a ^ b);
  }
}

// ---------------------------------------------------------------------------------------------
// Scrambler -- T5/Scrambler.m:1-28: s[i] = x[i] ^ s[i-13] ^ s[i-14]  (division by 1+x^13+x^14 over
// GF(2)).  One workgroup per frame; the frame is processed in chunks of SCR_CHUNK bits held
// bit-packed in LDS.  Inside a chunk the IIR is unrolled by repeated squaring:
//     1/P = P * P^2 * P^4 * ... / P^(2^k),   P^(2^m) = 1 + x^(13*2^m) + x^(14*2^m)
// and once 13*2^k exceeds the chunk length the remaining denominator only touches the (known)
// history, which is folded into the first 14 input bits.  log2(chunk/13) parallel stages.
// ---------------------------------------------------------------------------------------------
constexpr int SCR_CHUNK = 16384;              // bits per chunk
constexpr int SCR_WORDS = SCR_CHUNK / 32;     // 512 words
constexpr int SCR_THREADS = 256;

__device__ __forceinline__ uint32_t shl_vec(const uint32_t* w, int idx, int sh) {
  // word `idx` of the bit vector shifted towards higher bit positions by `sh` bits
  const int ws = sh >> 5, bs = sh & 31;
  const int a = idx - ws;
  uint32_t lo = (a >= 0) ? w[a] : 0u;
  if (bs == 0) return lo;
  uint32_t hi = (a - 1 >= 0) ? w[a - 1] : 0u;
  return (lo << bs) | (hi >> (32 - bs));
}

__global__ __launch_bounds__(SCR_THREADS) void scramble_kernel(const uint8_t* __restrict__ seq,
                                                               uint8_t* __restrict__ out,
                                                               int64_t frame_len, uint32_t reg_mask) {
  __shared__ uint32_t buf[2][SCR_WORDS];
  __shared__ uint32_t hist_s;                 // bit (j-1) = s[start - j], j = 1..14
  const uint8_t* x = seq + (int64_t)blockIdx.x * frame_len;
  uint8_t* y = out + (int64_t)blockIdx.x * frame_len;
  const int tid = threadIdx.x;
  if (tid == 0) hist_s = reg_mask & 0x3FFFu;
  __syncthreads();
  for (int64_t start = 0; start < frame_len; start += SCR_CHUNK) {
    const int len = (int)((frame_len - start < SCR_CHUNK) ? (frame_len - start) : SCR_CHUNK);
    const uint32_t hist = hist_s;
    // pack: word w holds bits 32w..32w+31 (bit i%32 of word i/32)
    for (int w = tid; w < SCR_WORDS; w += SCR_THREADS) {
      uint32_t v = 0;
      const int base = w * 32;
      if (base < len) {
        const int lim = (len - base < 32) ? (len - base) : 32;
        for (int b = 0; b < lim; ++b) v |= (uint32_t)(x[start + base + b] != 0) << b;
      }
      if (w == 0) {
        // fold history: x'[i] ^= s[i-13] (i<13) ^ s[i-14] (i<14); s[-j] = hist bit (j-1)
        uint32_t f = 0;
        for (int i = 0; i < 14; ++i) {
          uint32_t t = 0;
          if (i < 13) t ^= (hist >> (13 - i - 1)) & 1u;
          t ^= (hist >> (14 - i - 1)) & 1u;
          f |= t << i;
        }
        v ^= f;
      }
      buf[0][w] = v;
    }
    __syncthreads();
    int cur = 0;
    for (int s13 = 13, s14 = 14; s13 < len; s13 <<= 1, s14 <<= 1) {
      for (int w = tid; w < SCR_WORDS; w += SCR_THREADS)
        buf[cur ^ 1][w] = buf[cur][w] ^ shl_vec(buf[cur], w, s13) ^ shl_vec(buf[cur], w, s14);
      __syncthreads();
      cur ^= 1;
    }
    // unpack
    for (int i = tid; i < len; i += SCR_THREADS) y[start + i] = (uint8_t)((buf[cur][i >> 5] >> (i & 31)) & 1u);
    // next history = last 14 outputs of this chunk (older ones from the previous history)
    if (tid == 0) {
      uint32_t h = 0;
      for (int j = 1; j <= 14; ++j) {
        const int p = len - j;
        uint32_t bit = (p >= 0) ? ((buf[cur][p >> 5] >> (p & 31)) & 1u) : ((hist >> (j - len - 1)) & 1u);
        h |= bit << (j - 1);
      }
      hist_s = h;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// BER_func -- T5/BER_func.m:1-7: count of differing bits (integer, exact).
// ---------------------------------------------------------------------------------------------
__global__ void ber_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int64_t n,
                           unsigned long long* __restrict__ count) {
  unsigned int local = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    local += ((a[i] != 0) != (b[i] != 0));
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  __shared__ unsigned int wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) wsum[wid] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
    if (t) atomicAdd(count, t);
  }
}

// ---------------------------------------------------------------------------------------------
// MER_func -- T5/MER_func.m:1-26: nearest point, sum1 = sum |ideal|^2, sum2 = sum |ideal-rx|^2.
// Deterministic: per-block partial sums in double, final sum on the host.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void mer_kernel(const cx<T>* __restrict__ iq, int64_t n, DemapTable<T> tab, double* __restrict__ partial) {
  double s1 = 0, s2 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const cx<T> z = iq[i];
    const cx<T> p = demap_point(tab, demap_decide(tab, z));
    s1 += (double)p.x * p.x + (double)p.y * p.y;
    const double ex = (double)p.x - (double)z.x, ey = (double)p.y - (double)z.y;
    s2 += ex * ex + ey * ey;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_down(s1, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  __shared__ double w1[16], w2[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { w1[wid] = s1; w2[wid] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += w1[w]; b += w2[w]; }
    partial[2 * blockIdx.x] = a;
    partial[2 * blockIdx.x + 1] = b;
  }
}

static unsigned ew_grid(int64_t total, int per_block = 256) {
  int64_t b = (total + per_block - 1) / per_block;
  int64_t cap = (int64_t)ctx().num_cu * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

static uint32_t reg_to_mask(const uint8_t* reg15) {
  uint32_t m = 0;
  for (int j = 0; j < 15; ++j) m |= (uint32_t)(reg15[j] != 0) << j;   // bit (j-1) = R(j)
  return m;
}

template <typename T>
static int build_table(const char* name, DemapTable<T>& t, const char* who) {
  std::vector<c64> dict;
  ConstellationInfo info;
  OFDM_ARG(constellation_info(name, info) > 0, "%s: unknown constellation '%s'", who, name ? name : "(null)");
  constellation_table(name, dict);
  fill_demap_table<T>(dict, info, t);
  return OFDM_OK;
}

// device-pointer entry points shared with other translation units
int descramble_device(const uint8_t* seq, uint8_t* out, int64_t frame_len, int64_t n_frames, uint32_t reg_mask) {
  const int64_t total = frame_len * n_frames;
  if (total == 0) return OFDM_OK;
  hipLaunchKernelGGL(descramble_kernel, dim3(ew_grid(total)), dim3(256), 0, ctx().stream, seq, out, frame_len,
                     n_frames, reg_mask);
  return check_launch("descramble_kernel");
}

int scramble_device(const uint8_t* seq, uint8_t* out, int64_t frame_len, int64_t n_frames, uint32_t reg_mask) {
  if (frame_len == 0 || n_frames == 0) return OFDM_OK;
  OFDM_ARG(n_frames < (1ll << 31), "Scrambler: too many frames");
  hipLaunchKernelGGL(scramble_kernel, dim3((unsigned)n_frames), dim3(SCR_THREADS), 0, ctx().stream, seq, out,
                     frame_len, reg_mask);
  return check_launch("scramble_kernel");
}

}  // namespace ofdm

using namespace ofdm;

extern "C" {

int ofdm_mapping(const uint8_t* bits, int64_t n_bits, const char* constellation, void* iq_out, int* pad_out,
                 int flags) {
  OFDM_TRY(ensure_init());
  ConstellationInfo info;
  OFDM_ARG(constellation_info(constellation, info) > 0, "mapping: unknown constellation '%s'",
           constellation ? constellation : "(null)");
  OFDM_ARG(n_bits >= 0, "mapping: negative length");
  const int64_t rem = n_bits % info.bps;
  const int pad = rem ? (int)(info.bps - rem) : -1;                  // mapping.m:7-12
  const int64_t n_iq = (n_bits + info.bps - 1) / info.bps;
  if (pad_out) *pad_out = pad;
  Stage st(flags);
  const void* db; void* diq;
  OFDM_TRY(st.in(bits, (size_t)n_bits, &db));
  OFDM_TRY(st.out(iq_out, csize(flags) * (size_t)n_iq, &diq));
  if (n_iq > 0) {
    if (is_f64(flags)) {
      DemapTable<double> t;
      OFDM_TRY(build_table(constellation, t, "mapping"));
      hipLaunchKernelGGL(mapping_kernel<double>, dim3(ew_grid(n_iq)), dim3(256), 0, ctx().stream,
                         (const uint8_t*)db, n_bits, n_iq, t, (c64*)diq);
    } else {
      DemapTable<float> t;
      OFDM_TRY(build_table(constellation, t, "mapping"));
      hipLaunchKernelGGL(mapping_kernel<float>, dim3(ew_grid(n_iq)), dim3(256), 0, ctx().stream,
                         (const uint8_t*)db, n_bits, n_iq, t, (c32*)diq);
    }
    OFDM_TRY(check_launch("mapping_kernel"));
  }
  return st.finish();
}

int ofdm_demapping(int pad, const void* iq, int64_t n_iq, const char* constellation, uint8_t* bits_out, int flags) {
  OFDM_TRY(ensure_init());
  ConstellationInfo info;
  OFDM_ARG(constellation_info(constellation, info) > 0, "demapping: unknown constellation '%s'",
           constellation ? constellation : "(null)");
  OFDM_ARG(n_iq >= 0, "demapping: negative length");
  const int64_t nb = n_iq * info.bps - (pad != -1 ? pad : 0);
  OFDM_ARG(nb >= 0, "demapping: pad exceeds the number of bits");
  Stage st(flags);
  const void* diq; void* db;
  OFDM_TRY(st.in(iq, csize(flags) * (size_t)n_iq, &diq));
  OFDM_TRY(st.out(bits_out, (size_t)nb, &db));
  if (n_iq > 0) {
    if (is_f64(flags)) {
      DemapTable<double> t;
      OFDM_TRY(build_table(constellation, t, "demapping"));
      hipLaunchKernelGGL(demapping_kernel<double>, dim3(ew_grid(n_iq)), dim3(256), 0, ctx().stream,
                         (const c64*)diq, n_iq, nb, t, (uint8_t*)db);
    } else {
      DemapTable<float> t;
      OFDM_TRY(build_table(constellation, t, "demapping"));
      hipLaunchKernelGGL(demapping_kernel<float>, dim3(ew_grid(n_iq)), dim3(256), 0, ctx().stream,
                         (const c32*)diq, n_iq, nb, t, (uint8_t*)db);
    }
    OFDM_TRY(check_launch("demapping_kernel"));
  }
  return st.finish();
}

static int scr_common(bool descr, uint8_t* reg_inout, const uint8_t* reg_in, const uint8_t* seq, int64_t frame_len,
                      int64_t n_frames, uint8_t* out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(frame_len >= 0 && n_frames >= 0, "Scrambler: negative length");
  const uint8_t* reg = reg_inout ? reg_inout : reg_in;
  OFDM_ARG(reg != nullptr, "Scrambler: Register is null");
  const uint32_t mask = reg_to_mask(reg);
  const size_t total = (size_t)frame_len * n_frames;
  Stage st(flags);
  const void* ds; void* dout;
  OFDM_TRY(st.in(seq, total, &ds));
  OFDM_TRY(st.out(out, total, &dout));
  if (descr) OFDM_TRY(descramble_device((const uint8_t*)ds, (uint8_t*)dout, frame_len, n_frames, mask));
  else OFDM_TRY(scramble_device((const uint8_t*)ds, (uint8_t*)dout, frame_len, n_frames, mask));
  // final register (second output of the .m functions): R(j) = fed-back bit j steps ago
  uint8_t tail[15];
  const int ntail = (int)(frame_len < 15 ? frame_len : 15);
  if (reg_inout && n_frames == 1 && ntail > 0) {
    const uint8_t* src = descr ? (const uint8_t*)ds : (const uint8_t*)dout;   // feedback = received / scrambled bit
    OFDM_HIP(hipMemcpyAsync(tail, src + (frame_len - ntail), ntail, hipMemcpyDeviceToHost, ctx().stream));
    OFDM_HIP(hipStreamSynchronize(ctx().stream));
  }
  OFDM_TRY(st.finish());
  if (reg_inout && n_frames == 1) {
    uint8_t nr[15];
    for (int j = 1; j <= 15; ++j)
      nr[j - 1] = (j <= ntail) ? (uint8_t)(tail[ntail - j] != 0) : (uint8_t)(reg[j - ntail - 1] != 0);
    memcpy(reg_inout, nr, 15);
  }
  return OFDM_OK;
}

int ofdm_Scrambler(uint8_t* reg15, const uint8_t* seq, int64_t n, uint8_t* out, int flags) {
  return scr_common(false, reg15, nullptr, seq, n, 1, out, flags);
}
int ofdm_DeScrambler(uint8_t* reg15, const uint8_t* seq, int64_t n, uint8_t* out, int flags) {
  return scr_common(true, reg15, nullptr, seq, n, 1, out, flags);
}
int ofdm_Scrambler_frames(const uint8_t* reg15, const uint8_t* seq, int64_t frame_len, int64_t n_frames,
                          uint8_t* out, int flags) {
  return scr_common(false, nullptr, reg15, seq, frame_len, n_frames, out, flags);
}
int ofdm_DeScrambler_frames(const uint8_t* reg15, const uint8_t* seq, int64_t frame_len, int64_t n_frames,
                            uint8_t* out, int flags) {
  return scr_common(true, nullptr, reg15, seq, frame_len, n_frames, out, flags);
}

int ofdm_BER_func(const uint8_t* bit_tx, const uint8_t* bit_rx, int64_t n, int64_t* n_errors_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(n >= 0 && n_errors_out, "BER_func: bad arguments");
  Stage st(flags);
  const void *da, *db; void* dcnt;
  OFDM_TRY(st.in(bit_tx, (size_t)n, &da));
  OFDM_TRY(st.in(bit_rx, (size_t)n, &db));
  unsigned long long host_cnt = 0;
  OFDM_TRY(st.fetch(&host_cnt, sizeof(host_cnt), &dcnt));
  OFDM_HIP(hipMemsetAsync(dcnt, 0, sizeof(unsigned long long), ctx().stream));
  if (n > 0) {
    hipLaunchKernelGGL(ber_kernel, dim3(ew_grid(n, 1024)), dim3(256), 0, ctx().stream, (const uint8_t*)da,
                       (const uint8_t*)db, n, (unsigned long long*)dcnt);
    OFDM_TRY(check_launch("ber_kernel"));
  }
  OFDM_TRY(st.finish());
  *n_errors_out = (int64_t)host_cnt;
  return OFDM_OK;
}

int ofdm_MER_func(const void* iq, int64_t n, const char* constellation, double* mer_db_out, int flags) {
  OFDM_TRY(ensure_init());
  OFDM_ARG(n >= 0 && mer_db_out, "MER_func: bad arguments");
  ConstellationInfo info;
  OFDM_ARG(constellation_info(constellation, info) > 0, "MER_func: unknown constellation '%s'",
           constellation ? constellation : "(null)");
  Stage st(flags);
  const void* diq; void* dpart;
  OFDM_TRY(st.in(iq, csize(flags) * (size_t)n, &diq));
  const unsigned grid = ew_grid(n, 1024);
  std::vector<double> part(2 * (size_t)grid, 0.0);
  OFDM_TRY(st.fetch(part.data(), sizeof(double) * part.size(), &dpart));
  if (is_f64(flags)) {
    DemapTable<double> t;
    OFDM_TRY(build_table(constellation, t, "MER_func"));
    hipLaunchKernelGGL(mer_kernel<double>, dim3(grid), dim3(256), 0, ctx().stream, (const c64*)diq, n, t,
                       (double*)dpart);
  } else {
    DemapTable<float> t;
    OFDM_TRY(build_table(constellation, t, "MER_func"));
    hipLaunchKernelGGL(mer_kernel<float>, dim3(grid), dim3(256), 0, ctx().stream, (const c32*)diq, n, t,
                       (double*)dpart);
  }
  OFDM_TRY(check_launch("mer_kernel"));
  OFDM_TRY(st.finish());
  double s1 = 0, s2 = 0;
  for (unsigned b = 0; b < grid; ++b) { s1 += part[2 * b]; s2 += part[2 * b + 1]; }
  *mer_db_out = 10.0 * std::log10(s1 / s2);     // MER_func.m:25
  return OFDM_OK;
}

}  // extern "C"
