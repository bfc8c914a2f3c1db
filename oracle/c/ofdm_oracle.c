/*
 * ORACLE (C twin) -- TEST INFRASTRUCTURE ONLY.  Not shipped, not on the product path.
 *
 * Plain-C double-precision restatement of the Task-5 RX chain of ladnlav/OFDM-course, function by
 * function as the reference computes it on the CPU ("T5/x.m:a-b" = /root/reference/Task 5/x.m):
 *   OFDM_demodulator   T5/OFDM_demodulator.m:2-10     strip CP + fft per column
 *   OMP_estimate       T5/OMP_estimate.m:1-37         dense S'*r every iteration, LS refit, stop rule
 *   equalize_signal    T5/equalize_signal.m:1-8
 *   get_payload        T5/get_payload.m:2-4
 *   demapping          T5/demapping.m:1-25            full 2^bps distance search, first minimum
 *   BER_func           T5/BER_func.m:1-7              error count
 * It exists (a) as the CPU baseline of bench.py ("port", OpenMP over frames = the reference's
 * "can be switched to parfor" loop, T5/Task5_part2.m:146) and (b) as a second opinion on the numpy
 * oracle (tests/test_oracle_c.py compares the two).  PARITY STATUS: unpinned, like the numpy oracle
 * (no MATLAB, no reference vectors).
 *
 * MATLAB built-ins restated: fft -> iterative radix-2; pinv(A)*y (A full column rank) -> modified
 * Gram-Schmidt QR least squares; a repeated column (rank deficient) -> pinv's equal split.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double re, im; } cplx;

static inline cplx cmul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static inline cplx cmulc(cplx a, cplx b) { cplx r = {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; return r; } /* a*conj(b) */
static inline cplx csub(cplx a, cplx b) { cplx r = {a.re - b.re, a.im - b.im}; return r; }
static inline cplx cadd(cplx a, cplx b) { cplx r = {a.re + b.re, a.im + b.im}; return r; }
static inline cplx cdivc(cplx a, cplx b) {
  double d = b.re * b.re + b.im * b.im;
  cplx r = {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
  return r;
}

/* in-place forward FFT, n = power of two, tw[k] = exp(-2 pi i k / n) for k < n/2 */
static void fft_inplace(cplx* x, int n, const cplx* tw) {
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { cplx t = x[i]; x[i] = x[j]; x[j] = t; }
  }
  for (int len = 2; len <= n; len <<= 1) {
    const int half = len >> 1, step = n / len;
    for (int i = 0; i < n; i += len)
      for (int k = 0; k < half; ++k) {
        cplx u = x[i + k], v = cmul(x[i + k + half], tw[k * step]);
        x[i + k] = cadd(u, v);
        x[i + k + half] = csub(u, v);
      }
  }
}

/* OMP_estimate.m:1-37.  S is [np x K] column-major.  Returns number of picks; index 0-based. */
static int omp_estimate(const cplx* y, const cplx* S, int np, int K, int taps, int* index, cplx* x,
                        cplx* Q /* np*taps */, cplx* R /* taps*taps */, cplx* res, cplx* res_prev, cplx* qty) {
  int n = 0;
  memcpy(res_prev, y, sizeof(cplx) * np);
  for (int it = 0; it < taps; ++it) {
    /* [~,index] = max(abs(S'*r))  (:7,:14) -- first maximum */
    int best = 0;
    double bs = -1.0;
    for (int k = 0; k < K; ++k) {
      cplx acc = {0, 0};
      const cplx* col = S + (size_t)k * np;
      for (int p = 0; p < np; ++p) acc = cadd(acc, cmulc(res_prev[p], col[p]));
      const double s = hypot(acc.re, acc.im);
      if (s > bs) { bs = s; best = k; }
    }
    int dup = -1;
    for (int j = 0; j < n; ++j) if (index[j] == best) dup = j;
    if (dup >= 0) {
      /* pinv([.. a .. a]) * y : minimum-norm solution splits the coefficient; residual unchanged -> break */
      x[dup].re *= 0.5; x[dup].im *= 0.5;
      x[n] = x[dup];
      index[n++] = best;
      break;
    }
    index[n] = best;
    /* x = pinv(A)*y via MGS QR (A = S(:,index), full column rank) */
    cplx* q = Q + (size_t)n * np;
    memcpy(q, S + (size_t)best * np, sizeof(cplx) * np);
    for (int pass = 0; pass < 2; ++pass)            /* re-orthogonalise once */
      for (int j = 0; j < n; ++j) {
        cplx r = {0, 0};
        const cplx* qj = Q + (size_t)j * np;
        for (int p = 0; p < np; ++p) r = cadd(r, cmulc(q[p], qj[p]));        /* qj' * q */
        for (int p = 0; p < np; ++p) q[p] = csub(q[p], cmul(qj[p], r));
        if (pass == 0) R[j * taps + n] = r; else R[j * taps + n] = cadd(R[j * taps + n], r);
      }
    double nn = 0;
    for (int p = 0; p < np; ++p) nn += q[p].re * q[p].re + q[p].im * q[p].im;
    nn = sqrt(nn);
    for (int p = 0; p < np; ++p) { q[p].re /= nn; q[p].im /= nn; }
    R[n * taps + n].re = nn; R[n * taps + n].im = 0;
    cplx qy = {0, 0};
    for (int p = 0; p < np; ++p) qy = cadd(qy, cmulc(y[p], q[p]));            /* q' * y */
    qty[n] = qy;
    ++n;
    for (int i = n - 1; i >= 0; --i) {                                        /* R x = Q'y */
      cplx s = qty[i];
      for (int j = i + 1; j < n; ++j) s = csub(s, cmul(R[i * taps + j], x[j]));
      x[i] = cdivc(s, R[i * taps + i]);
    }
    /* residue = y - A*x  (:11,:18) */
    double dn = 0, pn = 0;
    for (int p = 0; p < np; ++p) {
      cplx acc = y[p];
      for (int j = 0; j < n; ++j) acc = csub(acc, cmul(S[(size_t)index[j] * np + p], x[j]));
      res[p] = acc;
      const cplx d = csub(acc, res_prev[p]);
      dn += d.re * d.re + d.im * d.im;
      pn += res_prev[p].re * res_prev[p].re + res_prev[p].im * res_prev[p].im;
    }
    memcpy(res_prev, res, sizeof(cplx) * np);
    if (it >= 1 && sqrt(dn) / sqrt(pn) < 1e-2) break;                         /* :20-22 */
  }
  return n;
}

/*
 * Full chain over n_frames frames.  Complex arrays are interleaved doubles.
 *   rx        [(nfft+tg)*n_symb x n_frames]
 *   pc1/dc1   1-based pilot / data carriers
 *   pilots    [np] pilot column
 *   dict      [2^bps] constellation table
 *   ref_bits  [n_frames x nd*n_symb*bps] (may be NULL), bits_out same shape (may be NULL)
 *   errors    [n_frames], h_out [n_frames x n_carrier] (may be NULL), index_out [n_frames x taps] 1-based, 0 = unused
 */
int oracle_rx_chain_task5(const double* rx_, int64_t n_frames, int nfft, int tg, int n_symb, int n_carrier,
                          const int32_t* pc1, int np, const int32_t* dc1, int nd, const double* pilots_, int K,
                          int taps, const double* dict_, int bps, const uint8_t* ref_bits, uint8_t* bits_out,
                          int64_t* errors, double* h_out, int32_t* index_out, int n_threads) {
  const cplx* rx = (const cplx*)rx_;
  const cplx* pilots = (const cplx*)pilots_;
  const cplx* dict = (const cplx*)dict_;
  const int M = 1 << bps, L = nfft + tg;
  const int64_t frame_bits = (int64_t)nd * n_symb * bps;
  cplx* tw = (cplx*)malloc(sizeof(cplx) * (nfft / 2));
  for (int k = 0; k < nfft / 2; ++k) { tw[k].re = cos(2.0 * M_PI * k / nfft); tw[k].im = -sin(2.0 * M_PI * k / nfft); }
  /* sensing matrix S = P*F(:,1:K)  (T5/Main_model_Task_5.m:182-190), closed form */
  cplx* S = (cplx*)malloc(sizeof(cplx) * (size_t)np * K);
  for (int k = 0; k < K; ++k)
    for (int p = 0; p < np; ++p) {
      const int64_t ph = ((int64_t)(pc1[p] - 1) * k) % nfft;
      S[(size_t)k * np + p].re = cos(2.0 * M_PI * (double)ph / nfft);
      S[(size_t)k * np + p].im = -sin(2.0 * M_PI * (double)ph / nfft);
    }
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
#pragma omp parallel
  {
    cplx* X = (cplx*)malloc(sizeof(cplx) * (size_t)nfft * n_symb);
    cplx* Y = (cplx*)malloc(sizeof(cplx) * np);
    cplx* Q = (cplx*)malloc(sizeof(cplx) * (size_t)np * taps);
    cplx* R = (cplx*)calloc((size_t)taps * taps, sizeof(cplx));
    cplx* res = (cplx*)malloc(sizeof(cplx) * np);
    cplx* resp = (cplx*)malloc(sizeof(cplx) * np);
    cplx* qty = (cplx*)malloc(sizeof(cplx) * taps);
    cplx* xs = (cplx*)malloc(sizeof(cplx) * (taps + 1));
    int* idx = (int*)malloc(sizeof(int) * (taps + 1));
    cplx* h = (cplx*)malloc(sizeof(cplx) * nfft);
#pragma omp for schedule(dynamic, 4)
    for (int64_t f = 0; f < n_frames; ++f) {
      const cplx* fr = rx + f * (int64_t)L * n_symb;
      for (int s = 0; s < n_symb; ++s) {                          /* OFDM_demodulator */
        memcpy(X + (size_t)s * nfft, fr + (size_t)s * L + tg, sizeof(cplx) * nfft);
        fft_inplace(X + (size_t)s * nfft, nfft, tw);
      }
      for (int p = 0; p < np; ++p) Y[p] = cdivc(X[pc1[p] - 1], pilots[p]);     /* Task5_part2.m:190 */
      const int n = omp_estimate(Y, S, np, K, taps, idx, xs, Q, R, res, resp, qty);
      memset(h, 0, sizeof(cplx) * nfft);
      for (int j = 0; j < n; ++j) h[idx[j]] = xs[j];              /* :31-33 last write wins */
      fft_inplace(h, nfft, tw);                                    /* H_OMP = fft(h)  :36 */
      if (h_out) memcpy(h_out + 2 * f * n_carrier, h, sizeof(cplx) * n_carrier);
      if (index_out) for (int j = 0; j < taps; ++j) index_out[f * taps + j] = j < n ? idx[j] + 1 : 0;
      int64_t err = 0;
      for (int s = 0; s < n_symb; ++s)
        for (int d = 0; d < nd; ++d) {
          const int k = dc1[d] - 1;
          const cplx z = cdivc(X[(size_t)s * nfft + k], h[k]);     /* equalize_signal + get_payload */
          int best = 0;                                            /* demapping.m:7-12 */
          double bd = (z.re - dict[0].re) * (z.re - dict[0].re) + (z.im - dict[0].im) * (z.im - dict[0].im);
          for (int c = 1; c < M; ++c) {
            const double dd = (z.re - dict[c].re) * (z.re - dict[c].re) + (z.im - dict[c].im) * (z.im - dict[c].im);
            if (dd < bd) { bd = dd; best = c; }
          }
          const int64_t b0 = f * frame_bits + ((int64_t)s * nd + d) * bps;
          for (int b = 0; b < bps; ++b) {
            const uint8_t bit = (uint8_t)((best >> (bps - 1 - b)) & 1);
            if (bits_out) bits_out[b0 + b] = bit;
            if (ref_bits) err += (bit != (ref_bits[b0 + b] != 0));
          }
        }
      if (errors) errors[f] = err;
    }
    free(X); free(Y); free(Q); free(R); free(res); free(resp); free(qty); free(xs); free(idx); free(h);
  }
  free(tw); free(S);
  return 0;
}

int oracle_c_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
