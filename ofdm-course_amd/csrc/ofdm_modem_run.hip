// OFDM_demodulator.m:2-10 / OFDM_modulator.m:2-11 for Nfft = 1024 and 2048 in fp32 with ONE WAVEFRONT PER RUN OF SYMBOLS -- the
// per-function entries the drivers call for every frame (BASELINE config 2 is 100 000 symbols through both).
//
// The cooperative form (modem_wave_kernel, ofdm_modem_wave.hip) gives a symbol to Nfft / 512 wavefronts: a radix-NW exchange
// with two workgroup barriers, then the column collected in LDS behind two more.  Here a symbol never leaves its wavefront (the
// decomposition of the metric kernel, ofdm_chain_wave.hip, with every output kept):
//   lane l holds x[l + 64 j], j < NJ = Nfft / 64 (coalesced 8-byte nontemporal loads)
//   1. NJ-point DFT over j = j0 + NC m in registers (NC = NJ / 8 dft8 over m, constant twiddles W_NJ, eight dft{NC} over j0)
//   2. Z[kj] *= W_Nfft^(l kj)                                   (LDS table, lane-contiguous)
//   3. NC rounds of eight consecutive kj = ka + 8 kb: a 64-point DFT across the lanes as 8 x 8 through two conflict-free
//      wave-private LDS transposes; round kb's registers are the sample class j = kb (mod NC) and are refilled with the next
//      symbol's samples the moment they are in LDS
//   4. bin k = kj + NJ (ka' + 8 q) leaves as 16-byte stores that complete 128-byte lines: an even round's eight outputs wait for
//      the odd round's, neighbouring bins sit in the two halves of a 16-lane row and are paired by one row_ror:8 (partial
//      lines cost the Task-4 demodulator a third of its time)
// Modulator: ifft(X) = conj(fft(conj X)) / Nfft on the same transform; the cyclic prefix rows are the same registers stored a
// second time.  No workgroup barrier after the table fill.
#include <algorithm>

#include "chain_fast_core.hpp"

namespace ofdm {

constexpr int MR_WPB = 4;
constexpr int MR_TR_ELEMS = 576;

// W_M^m = exp(-2 pi i m / M) for M = 16 or 32, m compile-time after unrolling
template <int M>
__device__ __forceinline__ cx<float> mr_wm(int m) {
  constexpr float C[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                          0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                          0.19509032201612826785f, 0.0f};
  m = (m * (32 / M)) & 31;
  const int q = m >> 3, r = m & 7;
  const float c = C[r], s = C[8 - r];
  switch (q) {
    case 0: return mk<float>(c, -s);
    case 1: return mk<float>(-s, -c);
    case 2: return mk<float>(-c, s);
    default: return mk<float>(s, c);
  }
}

template <int NJ, bool MOD>
__global__ __launch_bounds__(64 * MR_WPB, NJ == 32 ? 3 : 4) void modem_run_kernel(const cx<float>* __restrict__ in, cx<float>* __restrict__ out,
                                                                                 const cx<float>* __restrict__ tw, int64_t n_symb,
                                                                                 int t_guard, int spc) {
  using T = float;
  constexpr int N = 64 * NJ, NC = NJ / 8;
  constexpr unsigned OFF_TWB = 8 * 64 * (NJ - 1), OFF_WAVE = OFF_TWB + 8 * 64 * 7;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lane8 = 8u * lane;
  auto twn_at = [&](int row) { return *(const cx<T>*)(smem + 512 * row + lane8); };                 // W_N^(lane (row+1))
  auto twb_at = [&](int row) { return *(const cx<T>*)(smem + OFF_TWB + 512 * row + lane8); };       // W_64^((row+1) (lane&7))
  {
    cx<T>* const tn = (cx<T>*)smem;
    cx<T>* const tb = (cx<T>*)(smem + OFF_TWB);
    for (int i = threadIdx.x; i < (NJ - 1) * 64; i += blockDim.x) tn[i] = tw[((i / 64 + 1) * (i & 63)) & (N - 1)];
    for (int i = threadIdx.x; i < 7 * 64; i += blockDim.x) tb[i] = tw[(i / 64 + 1) * (i & 7) * (N / 64)];
  }
  __syncthreads();                                                     // the only workgroup barrier of the kernel
  const unsigned wbase = OFF_WAVE + (unsigned)wave * 8 * MR_TR_ELEMS;
  cx<T>* const t1w = (cx<T>*)(smem + wbase + lane8);
  cx<T>* const t1r = (cx<T>*)(smem + wbase) + 72 * (lane >> 3) + (lane & 7);
  cx<T>* const t2w = (cx<T>*)(smem + wbase) + 65 * (lane & 7) + 8 * (lane >> 3);
  cx<T>* const t2r = (cx<T>*)(smem + wbase + lane8);
  const bool odd = (lane >> 3) & 1;
  // first bin of the pair this lane stores for (round pair starting at kb - 1, q): even lane (k, k + 1) of the even round,
  // odd lane (k - 1, k) of the odd round, eight bins further on
  const int k_pair = (lane >> 3) + NJ * (lane & 7) + (odd ? 7 : 0);
  const int Lsym = N + t_guard;
  const int in_stride = MOD ? N : Lsym, in_off = MOD ? 0 : t_guard;
  const int64_t n_items = (n_symb + spc - 1) / spc;
  const int64_t n_waves = (int64_t)gridDim.x * MR_WPB;
  auto src_of = [&](int64_t s) { return in + s * in_stride + in_off + lane; };
  cx<T> v[NJ];
  const int64_t it0 = (int64_t)blockIdx.x * MR_WPB + wave;
  if (it0 < n_items) {
    const cx<T>* src = src_of(it0 * spc);
#pragma unroll
    for (int j = 0; j < NJ; ++j) v[j] = nt_load(src + 64 * j);
  }
  for (int64_t it = it0; it < n_items; it += n_waves) {
    const int64_t s0 = it * spc, s1 = std::min<int64_t>(s0 + spc, n_symb);
    for (int64_t s = s0; s < s1; ++s) {
      // what a finished round's registers are refilled with: the next symbol of the run, past its end the first symbol of this
      // wavefront's next run, past the last run nothing
      const int64_t sn = s + 1 < s1 ? s + 1 : (it + n_waves) * spc;
      const bool fetch = sn < n_symb;
      const cx<T>* nsrc = src_of(fetch ? sn : s);
      if constexpr (MOD) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[j].y = -v[j].y;                // conj(X)
      }
      // ---- 1. NJ-point DFT over j = j0 + NC m: Z[ka + 8 kb] ends in v[kb + NC ka]
#pragma unroll
      for (int j0 = 0; j0 < NC; ++j0) {
        dft8<T, false>(v[j0], v[j0 + NC], v[j0 + 2 * NC], v[j0 + 3 * NC], v[j0 + 4 * NC], v[j0 + 5 * NC], v[j0 + 6 * NC], v[j0 + 7 * NC]);
        if (j0 > 0) {
#pragma unroll
          for (int ka = 1; ka < 8; ++ka) v[j0 + NC * ka] = v[j0 + NC * ka] * mr_wm<NJ>(j0 * ka);
        }
      }
#pragma unroll
      for (int ka = 0; ka < 8; ++ka) {
        if constexpr (NC == 4) dft4<T, false>(v[4 * ka], v[4 * ka + 1], v[4 * ka + 2], v[4 * ka + 3]);
        else dft2<T, false>(v[2 * ka], v[2 * ka + 1]);
      }
      cx<T>* const drow = MOD ? out + s * Lsym + t_guard : out + s * N;
      cx<T> ev[8];
#pragma unroll
      for (int kb = 0; kb < NC; ++kb) {
#pragma unroll
        for (int ka = 0; ka < 8; ++ka) {
          const int kj = ka + 8 * kb;
          t1w[72 * ka] = kj == 0 ? v[kb] : v[kb + NC * ka] * twn_at(kj - 1);
        }
        if (fetch) {
#pragma unroll
          for (int m = 0; m < 8; ++m) v[kb + NC * m] = nt_load(nsrc + 64 * (kb + NC * m));
        }
        wave_sync();
        cx<T> u[8];
        lds_read8<8, true>(u, t1r);
        wave_sync();
        dft8<T, false>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]);
#pragma unroll
        for (int t = 1; t < 8; ++t) u[t] = u[t] * twb_at(t - 1);
#pragma unroll
        for (int t = 0; t < 8; ++t) t2w[t] = u[t];
        wave_sync();
        lds_read8<65, true>(u, t2r);
        wave_sync();
        dft8<T, false>(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7]);
        if constexpr (MOD) {
          const T sc = T(1) / (T)N;
#pragma unroll
          for (int q = 0; q < 8; ++q) u[q] = mk<T>(u[q].x * sc, -u[q].y * sc);     // conj(.) / Nfft
        }
        // bin (lane >> 3) + 8 kb + NJ ((lane & 7) + 8 q)
        if ((kb & 1) == 0) {
#pragma unroll
          for (int q = 0; q < 8; ++q) ev[q] = u[q];
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const cx<T> snd = odd ? ev[q] : u[q];
            const cx<T> rcv = mk<T>(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(snd.x), 0x128, 0xF, 0xF, false)),
                                    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(snd.y), 0x128, 0xF, 0xF, false)));
            const int k0 = k_pair + 8 * (kb - 1) + 8 * NJ * q;
            typedef float v4 __attribute__((ext_vector_type(4)));
            v4 o;
            if (odd) { o.x = rcv.x; o.y = rcv.y; o.z = u[q].x; o.w = u[q].y; }
            else { o.x = ev[q].x; o.y = ev[q].y; o.z = rcv.x; o.w = rcv.y; }
            __builtin_nontemporal_store(o, reinterpret_cast<v4*>(drow + k0));
            if constexpr (MOD) {                                       // OFDM_modulator.m:8-10: the last T_guard samples in front
              if (k0 >= N - t_guard) __builtin_nontemporal_store(o, reinterpret_cast<v4*>(drow + k0 - N));
            }
          }
        }
      }
    }
  }
}

// fp32, Nfft 1024 / 2048; the modulator stores pairs of samples behind the guard: T_guard even
bool modem_run_supported(int nfft, int t_guard, bool f64, bool mod) {
  if (f64 || (nfft != 1024 && nfft != 2048) || getenv("OFDM_MODEM_NO_RUN") || getenv("OFDM_MODEM_GENERIC")) return false;
  return !mod || ((t_guard & 1) == 0 && t_guard <= nfft);
}

int modem_run_launch(const void* in, void* out, const void* tw, int nfft, int64_t n_symb, int t_guard, bool mod) {
  if (n_symb == 0) return OFDM_OK;
  const int wps = nfft == 2048 ? 3 : 4;                                // wavefronts per SIMD the kernel is built for
  const int64_t resident = (int64_t)ctx().num_cu * 4 * wps;
  int spc = (int)std::min<int64_t>(64, std::max<int64_t>(1, n_symb / (5 * resident)));
  if (const char* e = getenv("OFDM_MODEM_RUN_SPC")) spc = std::max(1, atoi(e));
  const int64_t items = (n_symb + spc - 1) / spc;
  const unsigned grid = (unsigned)std::min<int64_t>((items + MR_WPB - 1) / MR_WPB, (int64_t)ctx().num_cu * wps);
  const size_t lds = (size_t)8 * 64 * (nfft / 64 - 1) + 8 * 64 * 7 + (size_t)MR_WPB * 8 * MR_TR_ELEMS;
#define MR_LAUNCH(NJV, MODV)                                                                                              \
  hipLaunchKernelGGL((modem_run_kernel<NJV, MODV>), dim3(grid), dim3(64 * MR_WPB), lds, ctx().stream, (const cx<float>*)in, \
                     (cx<float>*)out, (const cx<float>*)tw, n_symb, t_guard, spc)
  if (nfft == 2048) { if (mod) MR_LAUNCH(32, true); else MR_LAUNCH(32, false); }
  else { if (mod) MR_LAUNCH(16, true); else MR_LAUNCH(16, false); }
#undef MR_LAUNCH
  return check_launch("modem_run_kernel");
}

}  // namespace ofdm
