/*
 * ofdm_mi355x.h -- C ABI of libofdm_mi355x.so (MI355X / gfx950 native OFDM baseband hot path).
 *
 * Drop-in boundary for the Task 1-5 TX -> channel -> RX path of ladnlav/OFDM-course.
 * The reference has no FFI layer: its boundary is the MATLAB function-call interface
 * (one `function` per .m file).  Each entry point below replaces
 * exactly one reference function and is what a MEX gateway (mex/<name>.cpp, see
 * INTEGRATION.md) binds.  "T5/x.m:a-b" = /root/reference/Task 5/x.m lines a-b.
 *
 * Conventions (all entries):
 *   - arrays are MATLAB column-major; complex data is interleaved (re,im);
 *   - `flags` bit0: OFDM_F64 -> data are double / complex double, else float / complex float;
 *     the kernels compute in that same type (fp64 = parity mode, fp32 = throughput mode);
 *   - `flags` bit1: OFDM_DEVICE -> every *data* pointer is a device (HBM) pointer and the call
 *     is asynchronous on the library stream (ofdm_set_stream); otherwise data pointers are
 *     host pointers, the library stages them through HBM and the call returns synchronised;
 *   - index vectors (`*_carriers`, `pilot_loc`) are HOST int32 arrays holding the reference's
 *     1-based carrier indices; scalar out-params are HOST pointers (a call that returns
 *     scalars synchronises the stream);
 *   - bit vectors are uint8 arrays of 0/1 (one byte per bit), except the fused chain which
 *     uses packed bits (MSB-first inside each byte);
 *   - inputs are never modified; outputs are caller-allocated;
 *   - return value: 0 = ok, <0 = error (invalid argument / shape / HIP failure; text in
 *     ofdm_last_error_string()), >0 = soft condition (documented per function).  Nothing
 *     throws across the boundary.
 */
#ifndef OFDM_MI355X_H
#define OFDM_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFDM_F32 0
#define OFDM_F64 1
#define OFDM_HOST 0
#define OFDM_DEVICE 2

/* status codes */
#define OFDM_OK 0
#define OFDM_ERR_ARG (-1)       /* invalid argument / shape (MATLAB would raise an error) */
#define OFDM_ERR_HIP (-2)       /* HIP runtime failure */
#define OFDM_ERR_STATE (-3)     /* library not initialised / no GPU */
#define OFDM_ERR_UNSUPPORTED (-4)
#define OFDM_SOFT_ACF_FALLBACK 1 /* AutoCorrFunction: plateau not found, TgPosition = 65 (T5/AutoCorrFunction.m:21-24) */

/* ---- lifecycle -------------------------------------------------------------------------- */
int ofdm_init(int device_id);                 /* binds the calling process to one GPU          */
int ofdm_shutdown(void);                      /* frees plan / twiddle caches (MEX: mexAtExit)  */
const char* ofdm_last_error_string(void);
int ofdm_set_stream(void* hip_stream);        /* hipStream_t to launch on (NULL = default)     */
int ofdm_synchronize(void);
int ofdm_version(void);

/* ---- constellation / mapping (bit-exact) ------------------------------------------------- */
/* T5/constellation_func.m:4-35.  name in {BPSK,QPSK,8PSK,16QAM} + extensions {64QAM,256QAM}.
 * dict_out: 2^bps complex (host pointer always). */
int ofdm_constellation_func(const char* name, void* dict_out, int* bps_out, int flags);

/* T5/mapping.m:1-25.  bits[n_bits] -> iq[ceil(n_bits/bps)]; *pad_out = -1 when no padding. */
int ofdm_mapping(const uint8_t* bits, int64_t n_bits, const char* constellation,
                 void* iq_out, int* pad_out, int flags);

/* T5/demapping.m:1-25.  iq[n_iq] -> bits_out[n_iq*bps - max(pad,0)] (hard decision, first min). */
int ofdm_demapping(int pad, const void* iq, int64_t n_iq, const char* constellation,
                   uint8_t* bits_out, int flags);

/* ---- scrambler (bit-exact) --------------------------------------------------------------- */
/* T5/Scrambler.m:1-28 / T5/DeScrambler.m:1-28.  reg[15] (host, uint8 0/1) is read AND updated
 * with the final register (second output of the .m function). */
int ofdm_Scrambler(uint8_t* reg15, const uint8_t* seq, int64_t n, uint8_t* out, int flags);
int ofdm_DeScrambler(uint8_t* reg15, const uint8_t* seq, int64_t n, uint8_t* out, int flags);
/* Per-frame batch used by the drivers (register reset to reg15 for every frame,
 * T5/Main_model_Task_5.m:58-69): seq/out are [frame_len x n_frames] column-major. */
int ofdm_Scrambler_frames(const uint8_t* reg15, const uint8_t* seq, int64_t frame_len,
                          int64_t n_frames, uint8_t* out, int flags);
int ofdm_DeScrambler_frames(const uint8_t* reg15, const uint8_t* seq, int64_t frame_len,
                            int64_t n_frames, uint8_t* out, int flags);

/* ---- carriers / OFDM (I)FFT + CP ---------------------------------------------------------- */
/* T5/OFDM_map_carriers.m:2-9.  payload[n_data*n_symb]; pilot_values[n_pilots*n_symb] or a single
 * complex value when pilot_scalar != 0 (T3/Main_model_Task_3.m:59); out[nfft*n_symb]. */
int ofdm_OFDM_map_carriers(const void* payload, int64_t n_symb, int nfft,
                           const int32_t* data_carriers, int n_data,
                           const int32_t* pilot_carriers, int n_pilots,
                           const void* pilot_values, int pilot_scalar, void* out, int flags);
/* T5/get_payload.m:2-4.  x[nfft*n_symb] -> out[n_data*n_symb]. */
int ofdm_get_payload(const void* x, int nfft, int64_t n_symb,
                     const int32_t* data_carriers, int n_data, void* out, int flags);
/* T5/OFDM_modulator.m:2-11.  x[nfft*n_symb] -> y[(nfft+t_guard)*n_symb] (ifft, 1/nfft, CP). */
int ofdm_OFDM_modulator(const void* x, void* y, int nfft, int64_t n_symb, int t_guard, int flags);
/* T5/OFDM_demodulator.m:2-10.  y[(nfft+t_guard)*n_symb] -> x[nfft*n_symb] (strip CP, fft). */
int ofdm_OFDM_demodulator(const void* y, void* x, int nfft, int64_t n_symb, int t_guard, int flags);

/* ---- channel side ------------------------------------------------------------------------ */
/* T5/get_MP_channel_resp.m:2-19.  taps: host double [n_taps x 2] column-major (delay, amplitude)
 * (+ optional imaginary amplitudes taps_im[n_taps], may be NULL).  h_out[max_delay+1] and
 * H_out[nfft] complex, HOST pointers; *h_len_out = max_delay+1. */
int ofdm_get_MP_channel_resp(const double* taps, const double* taps_im, int n_taps, int nfft,
                             void* h_out, int* h_len_out, void* H_out, int flags);
/* conv(x,h.','full')(1:len) -- T5/Main_model_Task_5.m:126-127.  h is a HOST complex array. */
int ofdm_channel_conv(const void* x, int64_t len, const void* h, int h_len, void* y, int flags);
/* Batch of independent streams (one Monte-Carlo frame each, T5/Task5_part2.m:148-152): x/y are
 * [frame_len x n_frames] column-major, every frame starts from silence. */
int ofdm_channel_conv_frames(const void* x, int64_t frame_len, int64_t n_frames, const void* h, int h_len,
                             void* y, int flags);
/* T5/Noise.m:1-12 with a counter-based generator (Philox4x32-10 + Box-Muller, seed/stream):
 * y = x + sqrt(P/snr/2)*(n_re + i n_im).  *n_var_out = sqrt(NoisePower) (Noise.m:11). */
int ofdm_Noise(double snr_db, const void* x, int64_t len, uint64_t seed, uint32_t stream,
               void* y, double* n_var_out, int flags);
/* One Noise() call per frame (power measured per frame, Philox stream = stream0 + frame). */
int ofdm_Noise_frames(double snr_db, const void* x, int64_t frame_len, int64_t n_frames, uint64_t seed,
                      uint32_t stream0, void* y, int flags);
/* The same with frame f at its own snr_db[f] (HOST array): the SNR sweep of T5/Main_model_Task_5.m:305-307 as a batch. */
int ofdm_Noise_frames_snr(const double* snr_db, const void* x, int64_t frame_len, int64_t n_frames, uint64_t seed,
                          uint32_t stream0, void* y, int flags);
/* T5/add_STO.m:1-10 and T5/add_CFO.m:1-8. */
int ofdm_add_STO(const void* y, int64_t len, int64_t n_sto, void* out, int flags);
int ofdm_add_CFO(const void* y, int64_t len, double cfo, int nfft, void* out, int flags);
/* add_STO then add_CFO of a batch of frames y[frame_len x n_frames], frame f with its own n_sto[f] (int64) and cfo[f]
 * (double) -- the per-run draws of T4/Main_model_Task_4.m:101-110; either array may be NULL (stage off).  The arrays
 * live where `flags` says. */
int ofdm_add_STO_CFO_frames(const void* y, int64_t frame_len, int64_t n_frames, const int64_t* n_sto, const double* cfo,
                            int nfft, void* out, int flags);

/* ---- synchronisation --------------------------------------------------------------------- */
/* T5/AutoCorrFunction.m:1-28.  rho_out[len-width-nfft] complex (may be NULL);
 * returns OFDM_SOFT_ACF_FALLBACK when the catch branch (:21-24) is taken. */
int ofdm_AutoCorrFunction(const void* rx, int64_t len, int width_window, int nfft,
                          void* rho_out, int64_t* tg_position_out, double* freq_offset_out, int flags);
/* T5/remove_IFO.m:1-11. */
int ofdm_remove_IFO(const void* rx, int64_t len, int nfft, void* fixed_out, int* ifo_out, int flags);
/* T5/fine_sync.m:1-45 (variant 0) / T4/fine_sync.m:1-60 (variant 1).  pilot_values[n_pilots*n_symb].
 * tau_out / phase_out (HOST, may be NULL) receive the two internal estimates. */
int ofdm_fine_sync(const void* rx, int nfft, int64_t n_symb, const int32_t* pilot_carriers, int n_pilots,
                   const void* pilot_values, int time_desync, int freq_desync, int variant,
                   void* out, double* tau_out, double* phase_out, int flags);

/* ---- channel estimation / equalisation ---------------------------------------------------- */
/* T5/interpolate.m:1-24.  method: 'l'inear or 's'pline.  h[n_pilots] -> out[n_out]. */
int ofdm_interpolate(const void* h, const int32_t* pilot_loc, int n_pilots, int n_out, char method,
                     void* out, int flags);
/* T5/estimate_channel.m:1-9.  all_carriers[n_all] (1-based query points). */
int ofdm_estimate_channel(const void* rx, int nfft, int64_t n_symb, const int32_t* all_carriers, int n_all,
                          const int32_t* pilot_carriers, int n_pilots, const void* pilot_values,
                          void* h_est_out, void* h_pilots_out, int flags);
/* T5/equalize_signal.m:1-8. */
int ofdm_equalize_signal(const void* x, int nfft, int64_t n_symb, const void* h_est, int n_carrier,
                         void* out, int flags);
/* T5/LS_CE.m:1-34 (symbol 1 only).  y[nfft*n_symb], xp[n_pilots*n_symb] -> h_ls[n_carrier]. */
int ofdm_LS_CE(const void* y, int nfft, int64_t n_symb, const void* xp, const int32_t* pilot_loc,
               int n_pilots, int n_carrier, void* h_ls_out, int flags);
/* T5/MMSE_CE.m:1-39.  h[h_len] = CIR guess, snr_db. */
int ofdm_MMSE_CE(const void* y, int nfft, int64_t n_symb, const void* xp, const int32_t* pilot_loc,
                 int n_pilots, int n_carrier, const void* h, int h_len, double snr_db,
                 void* h_mmse_out, int flags);
/* Sensing matrix of T5/Main_model_Task_5.m:182-190 in closed form: s_out[n_pilots x k]. */
int ofdm_sensing_matrix(const int32_t* pilot_carriers, int n_pilots, int nfft, int k, void* s_out, int flags);
/* T5/MP_estimate.m:1-34.  y[n_pilots], s[n_pilots x k]; H_out[nfft], h_out[nfft],
 * picks_out[dominant_taps] (HOST int32, 1-based, may be NULL). */
int ofdm_MP_estimate(const void* y, const void* s, int n_pilots, int k, int nfft, int dominant_taps,
                     void* H_out, void* h_out, int32_t* picks_out, int flags);
/* T5/OMP_estimate.m:1-37.  index_out[dominant_taps] (HOST, 1-based), *n_index_out = picks made. */
int ofdm_OMP_estimate(const void* y, const void* s, int n_pilots, int k, int nfft, int dominant_taps,
                      double snr_db, void* H_out, void* h_out, int32_t* index_out, int* n_index_out, int flags);

/* ---- metrics ----------------------------------------------------------------------------- */
/* T5/BER_func.m:1-7: *n_errors_out = sum(tx != rx) (HOST).  BER = n_errors / n. */
int ofdm_BER_func(const uint8_t* bit_tx, const uint8_t* bit_rx, int64_t n, int64_t* n_errors_out, int flags);
/* T5/MER_func.m:1-26. */
int ofdm_MER_func(const void* iq, int64_t n, const char* constellation, double* mer_db_out, int flags);

/* ---- PAPR study of Task 2 (T2/Main_model_Task_2.m:69-82) ------------------------------------- */
/* T2/calculatePAPR.m:2-11: *papr_db_out = 10 log10(max|x|^2 / mean|x|^2) (HOST scalar). */
int ofdm_calculatePAPR(const void* x, int64_t n, double* papr_db_out, int flags);
/* T2/calculate_window_PAPR.m:2-15: paprs_out[i] = PAPR of x(i : i+Nfft-1), i < n - nfft + 1 (double whatever the
 * input precision; nothing is written when n < nfft).  O(1) per window (sliding maximum + prefix sums), nfft <= 8192. */
int ofdm_calculate_window_PAPR(const void* x, int64_t n, int nfft, double* paprs_out, int flags);
/* T2/calculateCCDF.m:2-6 ([CCDF, x] = ecdf(values); CCDF = 1 - CCDF): sorted distinct values with the smallest one
 * repeated in front, NaN ignored.  Outputs hold up to n + 1 doubles; *n_out (HOST) = entries written. */
int ofdm_calculateCCDF(const double* papr_values, int64_t n, double* papr_ccdf_out, double* ccdf_out, int64_t* n_out, int flags);

/* ---- fused Task-5 RX chain (the benchmark path; everything stays in HBM) -------------------- */
/* Per frame of n_symb symbols (call order of T5/Task5_part2.m:169-193,:272,:279-303 with the
 * sensing matrix of T5/Main_model_Task_5.m:182-190):
 *   OFDM_demodulator -> Y = RX(pilots,1)./pilot -> OMP_estimate -> equalize_signal(1..n_carrier)
 *   -> get_payload -> demapping -> BER_func against ref bits.
 * The plan owns carrier index tables, the pilot column and the sensing matrix on the device. */
typedef struct ofdm_rx_plan ofdm_rx_plan;
int ofdm_rx_plan_create(ofdm_rx_plan** plan_out, int nfft, int t_guard, int n_symb, int n_carrier,
                        const int32_t* pilot_carriers, int n_pilots,
                        const int32_t* data_carriers, int n_data,
                        const void* pilot_values_col /* [n_pilots] complex, HOST */,
                        int k_atoms, int dominant_taps, const char* constellation, int flags);
int ofdm_rx_plan_destroy(ofdm_rx_plan* plan);
/* rx[(nfft+t_guard)*n_symb*n_frames] (DEVICE or HOST per flags);
 * bits_out: packed decided bits, per frame ceil(n_data*n_symb*bps/8)*... see ofdm_rx_plan_frame_bytes;
 * ref_bits (may be NULL): packed reference bits, same layout;
 * errors_out[n_frames] uint32 bit errors per frame (may be NULL when ref_bits is NULL);
 * h_out (may be NULL): [n_carrier x n_frames] OMP estimate; index_out (may be NULL): [dominant_taps x n_frames]
 * 1-based picks, 0 = unused slot. */
/* MMSE mode of a plan: ofdm_rx_chain_task5 then estimates the channel of every frame with
 * MMSE_CE(Y, Xp, pilot_loc, Nfft, N_carrier, h, SNR) (T5/MMSE_CE.m:1-39, as called per Monte-Carlo run at
 * T5/Task5_part2.m:176-177) instead of OMP_estimate, and index_out is not written.  For a fixed (h, SNR) that
 * estimator is one linear operator on the pilot LS values of symbol 1 (rms delay spread :19-24, Rpp :33-35, the
 * mrdivide of :36 and the spline re-interpolation of :38 folded together); it is built here once, on the host in
 * double, and applied to a whole batch as one complex GEMM (matrix cores in fp32).  h: host array of n_h complex
 * values in the precision of `flags` (OFDM_F32 / OFDM_F64); h = NULL returns the plan to OMP mode.
 * Needs 2..512 pilots; the chain call needs the fast-path geometry (Nfft 512..4096). */
int ofdm_rx_plan_set_mmse(ofdm_rx_plan* plan, const void* h, int64_t n_h, double snr_db, int flags);
/* Task-4 receiver over a batch of frames, each frame one stream of N_symb guarded symbols (T4/Main_model_Task_4.m:278-347
 * per frame, the plan supplying Nfft, T_guard, N_symb, the carrier sets, the pilot column and the constellation):
 *   if time_desync || freq_desync:  AutoCorrFunction(Rx, T_guard, Nfft)                              (:278)
 *   if time_desync:                 add_STO(Rx, TgPosition); add_STO(., -(Nfft+T_guard))             (:292-294)
 *   if freq_desync:                 add_CFO(., -FreqOffset, Nfft); remove_IFO(., Nfft)               (:301-303)
 *   OFDM_demodulator; fine_sync (Task-4 variant) if either flag                                      (:310-314)
 *   if mp_desync:                   estimate_channel + equalize_signal                               (:318-334)
 *   get_payload; demapping                                                                           (:340-347)
 * bits_out: packed demapped bits per frame (layout of ofdm_rx_chain_task5), after the per-frame DeScrambler of :354-364
 * when the plan has one (ofdm_rx_plan_set_descrambler); errors_out: differences to ref_bits (the scrambled bits without
 * a descrambler, the payload bits with it).
 * tg_position_out / freq_offset_out / ifo_out: the per-frame estimates (0 where a stage is off);
 * status_out: 0 ok, 1 = AutoCorrFunction's catch branch (TgPosition 65, :19-24), -1 = remove_IFO found no line above
 * 0.77 (the script would abort at remove_IFO.m:8; the frame is decoded with IFO = 0), -2 = TgPosition beyond AutoCorr.
 * h_out (optional): H_est(1..N_carrier) per frame.  All arrays live where `flags` says.  At most 65535 frames per call. */
int ofdm_rx_chain_task4(ofdm_rx_plan* plan, const void* rx, int64_t n_frames, int time_desync, int freq_desync, int mp_desync,
                        uint8_t* bits_out, const uint8_t* ref_bits, uint32_t* errors_out, int64_t* tg_position_out,
                        double* freq_offset_out, int32_t* ifo_out, int32_t* status_out, void* h_out, int flags);
/* Synthetic RX frames of the plan's geometry, generated on the device (no host payload), per frame:
 *   payload (one Philox4x32-10 draw per QAM symbol, stream = frame0 + f)
 *   -> Scrambler with the register reset per frame (T5/Main_model_Task_5.m:55-69; scr_reg15 = HOST uint8[15], NULL = off)
 *   -> mapping (T5/mapping.m) -> OFDM_map_carriers with the plan's pilot column on every symbol -> OFDM_modulator
 *   -> channel stages.  noise_first != 0 is the REFERENCE order (T5/Main_model_Task_5.m:106-127, T4/Main_model_Task_4.m:
 *      94-110,:257-267, T5/Task5_part2.m:134,:152):  Noise(snr_db) -> add_STO -> add_CFO -> conv(x, h) truncated per frame;
 *      noise_first == 0 (the order ofdm_tx_frames has always used):  add_STO -> add_CFO -> conv -> Noise, i.e. the SNR is
 *      set on the channel's output.  h = host array of h_len complex taps in the precision of `flags`, NULL = no channel;
 *      noise_on == 0 = no Noise stage.
 *   STO / CFO per frame (T4/Main_model_Task_4.m:101-110): mode 0 = off, 1 = sto_value / cfo_value for every frame,
 *      2 = drawn per frame from Philox counter (0, 0, frame0 + f, 2): Time_Delay = word0 mod (Nfft + T_guard + 1)
 *      [randi([0, Nfft+T_Guard])], Freq_Shift = (word1 mod 31) + ((word2 + 0.5) 2^-32 - 0.5) [randi([0,30]) + (rand-0.5)].
 * rx_out: [frame_samples x n_frames]; ref_bits_out: the packed PAYLOAD (input) bits [n_frames][frame_bytes] in the layout
 * ofdm_rx_chain_task5 compares against (with the Scrambler on, compare after ofdm_rx_plan_set_descrambler);
 * bits_out (optional): the payload, one byte per bit [n_frames][frame_bits]; sc_ref_bits_out (optional, Scrambler on): the
 * packed SCRAMBLED bits, what the demapper decides; sto_out / cfo_out (optional): the per-frame draws (int64 / double).
 * All outputs live where `flags` says.  Results depend on (seed, frame0 + f) only, not on how frames are batched. */
int ofdm_tx_frames_ex(ofdm_rx_plan* plan, const void* h, int h_len, double snr_db, int noise_on, uint64_t seed,
                      int64_t frame0, int64_t n_frames, const uint8_t* scr_reg15, int sto_mode, int64_t sto_value,
                      int cfo_mode, double cfo_value, int noise_first, void* rx_out, uint8_t* ref_bits_out,
                      uint8_t* bits_out, uint8_t* sc_ref_bits_out, int64_t* sto_out, double* cfo_out, int flags);
/* = ofdm_tx_frames_ex without Scrambler / STO / CFO and with noise_first = 0 (conv -> Noise; NOT the order of the
 * reference's drivers, which add the noise before the channel -- use ofdm_tx_frames_ex(noise_first = 1) for that). */
int ofdm_tx_frames(ofdm_rx_plan* plan, const void* h, int h_len, double snr_db, int noise_on, uint64_t seed,
                   int64_t frame0, int64_t n_frames, void* rx_out, uint8_t* ref_bits_out, uint8_t* bits_out, int flags);
/* Per-frame DeScrambler inside ofdm_rx_chain_task5 / ofdm_rx_chain_task4 (T5/DeScrambler.m:1-16 with
 * the register reset for every frame, T5/Main_model_Task_5.m:257-274, T4/Main_model_Task_4.m:354-364): the demapped bits
 * of a frame go through d[i] = s[i] ^ s[i-13] ^ s[i-14], s[-m] = reg15[m-1], before they are written to bits_out and
 * compared with ref_bits (which then hold the TX's INPUT bits): inside the pack stage of the wave-per-frame symbol kernel
 * (the metric geometry), as one more pass over the packed words on every other path.  ofdm_task5_part2_tile refuses a plan
 * with a DeScrambler (the study runs unscrambled, T5/Task5_part2.m:104-114).  reg15: HOST uint8[15]; NULL = off. */
int ofdm_rx_plan_set_descrambler(ofdm_rx_plan* plan, const uint8_t* reg15);
int64_t ofdm_rx_plan_frame_bytes(const ofdm_rx_plan* plan);   /* packed bytes per frame (4-byte multiple) */
/* One tile of the Monte-Carlo study of T5/Task5_part2.m:148-205, :269-304: n_frames channel realisations jj of the plan's
 * pilot scenario kk, all four estimators on every realisation, nothing but sums returned.
 *   tx_noised[(nfft+t_guard)*n_symb]: the scenario's noisy TX stream (:130-134; DEVICE or HOST per flags);
 *   tap_delay[n_frames][n_ch_taps] (0-based sample delays), tap_amp[n_frames][n_ch_taps] (interleaved complex double):
 *   the realisation's channel (what :150-155 hand to conv, :160-166) -- HOST arrays;
 *   ref_bits: the scenario's packed payload bits, ONE frame (layout of ofdm_rx_chain_task5; all realisations share the TX).
 * Per realisation: conv + truncate, OFDM_demodulator, LS_CE (:174), MMSE_CE with h = the true CIR and snr_db (:176-177),
 * MP_estimate and OMP_estimate with dominant_taps = the plan's (:192-193; the plan's K = the dictionary's columns),
 * NMSE of each against fft(h) on carriers 1..N_carrier (:202-205), equalize_signal / get_payload / demapping / BER_func
 * for each (:269-304).
 *   nmse_out[4][n_frames] (double), errors_out[4][n_frames] (uint32 bit errors): rows LS, MMSE, MP, OMP.
 * The MP projections S^H * residue run as one real GEMM per iteration on the matrix cores in fp32 when Np is a multiple of
 * 16 (dense dictionaries of random pilot masks, :58-64).  At most 65535 realisations and 64 channel taps per call. */
int ofdm_task5_part2_tile(ofdm_rx_plan* plan, const void* tx_noised, const int32_t* tap_delay, const double* tap_amp,
                          int n_ch_taps, int64_t n_frames, double snr_db, const uint8_t* ref_bits, double* nmse_out,
                          uint32_t* errors_out, int flags);
/* One tile of the MSE(SNR) sweep of T5/Main_model_Task_5.m:303-346: the frames of the tile are its n_points SNR values.
 * Per point i:  Noise(snr_db[i], Tx) (:307; Philox stream stream0 + i, key = seed) -> conv(Rx, h) truncated (:308-309) ->
 * OFDM_demodulator (:311) -> LS_CE (:313) -> MMSE_CE with h = ifft(H_est_LS) and snr_db[i] (:314-315) -> MP_estimate and
 * OMP_estimate on Y = RX(pilotCarriers, 1) ./ pilot column with dominant_taps = the plan's (:328-331; the plan's K = the
 * dictionary's columns, :318-320) -> the four mean squared errors against fft(h) on carriers 1..N_carrier (:334-344).
 *   tx[(nfft+t_guard)*n_symb]: the clean TX stream (DEVICE or HOST per flags); tap_delay[n_ch_taps] (0-based sample delays) /
 *   tap_amp[n_ch_taps] (interleaved complex double): the channel, HOST; snr_db[n_points]: HOST;
 *   mse_out[4][n_points] (double, where `flags` says): rows LS, MMSE, MP, OMP.
 * The plan may have no data carriers (comb = 1, the script as committed: pilots on every carrier).  Everything stays on the
 * device: the 61-point sweep of the script is one call.  At most 65535 points and 64 channel taps per call. */
int ofdm_task5_mse_tile(ofdm_rx_plan* plan, const void* tx, const int32_t* tap_delay, const double* tap_amp, int n_ch_taps,
                        const double* snr_db, int64_t n_points, uint64_t seed, uint32_t stream0, double* mse_out, int flags);
/* Measurement aid: with timing enabled every ofdm_rx_chain_task5 call brackets its launches with HIP
 * events on the launch stream; ms3 = {symbol-1 kernel, OMP kernel, symbols kernel} of the last call
 * (comb pilot layouts run the first two as one launch and report {symbol-1 + OMP kernel, 0, symbols
 * kernel}; the generic single-kernel path reports {0, 0, total}). */
int ofdm_rx_plan_set_timing(ofdm_rx_plan* plan, int enable);
int ofdm_rx_plan_last_kernel_ms(ofdm_rx_plan* plan, float* ms3);
/* The same for ofdm_rx_chain_task4: ms5 = {AutoCorrFunction stage, remove_IFO stage, OFDM_demodulator, fine_sync +
 * estimate_channel, equalise + demap (+ DeScrambler)} of the last call (T4/Main_model_Task_4.m:278, :301-303, :308-310,
 * :313-318, :334-364). */
int ofdm_rx_plan_last_task4_ms(ofdm_rx_plan* plan, float* ms5);
int ofdm_rx_chain_task5(ofdm_rx_plan* plan, const void* rx, int64_t n_frames,
                        uint8_t* bits_out, const uint8_t* ref_bits, uint32_t* errors_out,
                        void* h_out, int32_t* index_out, int flags);

#ifdef __cplusplus
}
#endif
#endif /* OFDM_MI355X_H */
