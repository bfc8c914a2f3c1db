"""Replay of `Task 5/Main_model_Task_5.m`: LS / MMSE / MP / OMP channel estimates on one frame (:6-300) and
their MSE(SNR) sweep (:303-359)."""
from __future__ import annotations

import numpy as np

from . import common as c

CHANNEL_TAPS = np.array([[0, 1.0], [4, .8], [10, .6], [15, .4], [21, .2], [25, .1]])   # T5/Main_model_Task_5.m:112-119


def _ifft_row(lib, x):
    """`ifft(H_est_LS_l)` of :179 -- an OFDM_modulator call without guard is the same transform."""
    x = np.asarray(x).ravel()
    return np.asarray(lib.OFDM_modulator(x.reshape(-1, 1), 0)).ravel()


def _estimates(lib, Xr, pilotValues, pilotCarriers, Nfft, N_carrier, amp_col, S, n_taps, SNR_dB, H_freq):
    """:178-205 / :313-345 -- the four estimators on one demodulated frame and their MSE against H_freq."""
    H_est_LS_l = lib.LS_CE(Xr, pilotValues, pilotCarriers, N_carrier)                          # :178
    h_t_mmse = _ifft_row(lib, H_est_LS_l)                                                      # :179
    H_est_MMSE = lib.MMSE_CE(Xr, pilotValues, pilotCarriers, Nfft, N_carrier, h_t_mmse, SNR_dB)   # :180
    Y = np.asarray(lib.get_payload(np.asarray(Xr)[:, :1], pilotCarriers)).ravel() / amp_col    # :191
    H_est_MP, _ = lib.MP_estimate(Y, S, Nfft, n_taps)                                          # :193
    H_est_OMP, _, index = lib.OMP_estimate(Y, S, Nfft, n_taps, SNR_dB)                         # :194
    H = {"LS": H_est_LS_l, "MMSE": H_est_MMSE, "MP": H_est_MP, "OMP": H_est_OMP}
    return H, {k: c.mse_row(H_freq, v, N_carrier) for k, v in H.items()}, np.asarray(index)    # :196-205


def run(lib=None, Nfft=4096, N_carrier=1024, Amount_OFDM_Frames=2, Amount_ODFM_SpF=7, comb=1,
        Constellation="16QAM", SNR_dB=20, noise_desync=1, mp_desync=1, SNRs=None, channel_taps=None, seed=1,
        batched=None, precision="fp64", rank=0, world=1):
    """T5/Main_model_Task_5.m.  comb = 1 (as committed, :13) sends pilots only; comb > 1 also decodes a payload.

    batched: run the MSE(SNR) sweep of :303-346 as device-resident tiles (`lib.task5_mse_tile`: noise, channel, demodulator
    and the four estimators of all SNR points in one call) instead of point by point; default = whenever `lib` has that
    entry (the HIP library does, the oracle adapter of the tests does not).  precision: of the batched tile.
    rank / world: the SNR points are independent; every rank takes a contiguous block of them (one tile call) and fills
    its own columns of `MSEs` (the others stay 0: one SUM all-reduce assembles the table)."""
    lib = lib or c.default_lib()
    T_Guard = Nfft // 8
    N_symb = Amount_OFDM_Frames * Amount_ODFM_SpF
    allCarriers, pilotCarriers, dataCarriers = c.layout_comb(Nfft, N_carrier, comb)              # :17-35
    dict_, bps = lib.constellation_func(Constellation)
    amp_pilots = 4 / 3 * np.max(np.abs(dict_))                                                  # :42-43
    pilotValues = np.full((len(pilotCarriers), N_symb), amp_pilots, dtype=np.complex128)         # :44-46
    taps = CHANNEL_TAPS if channel_taps is None else np.asarray(channel_taps)
    input_bits = pad = None
    if comb != 1:
        Size_Buffer = N_symb * len(dataCarriers) * bps
        input_bits = c.synthetic_bits(Size_Buffer, seed)                                        # :51
        sc_bits = c.scramble_per_frame(lib, "Scrambler", input_bits, Amount_OFDM_Frames)        # :58-69
        TX_IQ, pad = lib.mapping(sc_bits, Constellation)                                        # :72
        X = lib.OFDM_map_carriers(TX_IQ, N_symb, Nfft, dataCarriers, pilotCarriers, pilotValues)   # :75
    else:
        X = np.zeros((Nfft, N_symb), dtype=np.complex128)                                       # :78-80
        X[pilotCarriers.astype(int) - 1, :] = pilotValues
    Tx = np.asarray(lib.OFDM_modulator(X, T_Guard)).ravel(order="F")                            # :83-85

    H_tau, H_freq = lib.get_MP_channel_resp(taps, Nfft)                                         # :123
    K = int(np.ceil(N_carrier / comb))                                                          # :184
    S = lib.sensing_matrix(pilotCarriers, Nfft, K)                                              # :182-190 closed form

    Rx = Tx
    if noise_desync:
        Rx, _ = lib.Noise(SNR_dB, Rx, seed=seed, stream=0)                                      # :108
    if mp_desync:
        Rx = c.conv_truncate(lib, Rx, H_tau)                                                    # :126-127
    rx = np.asarray(Rx).reshape((Nfft + T_Guard, N_symb), order="F")                            # :140
    Xr = lib.OFDM_demodulator(rx, T_Guard)                                                      # :142
    res = {"driver": "Task 5/Main_model_Task_5.m", "comb": comb, "SNR_dB": SNR_dB}
    if mp_desync:
        H_est, _ = lib.estimate_channel(Xr, allCarriers, pilotCarriers, pilotValues)            # :163
        H, mse, index = _estimates(lib, Xr, pilotValues, pilotCarriers, Nfft, N_carrier, amp_pilots, S,
                                   taps.shape[0], SNR_dB, H_freq)
        res.update(MSE=mse, OMP_index=index, _H=H, _H_est=np.asarray(H_est))
        Xr = lib.equalize_signal(Xr, H_est, N_carrier)                                          # :244
    if comb != 1:
        RX_IQ = np.asarray(lib.get_payload(Xr, dataCarriers)).ravel(order="F")                  # :247-248
        output_bits = np.asarray(lib.demapping(pad, RX_IQ, Constellation)).ravel()              # :254
        dsc_bits = c.scramble_per_frame(lib, "DeScrambler", output_bits, Amount_OFDM_Frames)    # :257-271
        BER = float(lib.BER_func(input_bits, dsc_bits))                                         # :274
        res.update(BER=BER, passed=bool(BER < 0.2), MER_dB=float(lib.MER_func(RX_IQ, Constellation)),   # :275-283
                   _dsc_bits=dsc_bits)

    SNRs = np.arange(0, 30.5, 0.5) if SNRs is None else np.asarray(SNRs, dtype=float)           # :303
    MSEs = np.zeros((4, len(SNRs)))                                                             # :304 LS, MMSE, MP, OMP
    n_pts = len(SNRs)
    mine = list(range(rank * n_pts // world, (rank + 1) * n_pts // world))      # a contiguous block per rank: one tile call
    use_tile = (batched if batched is not None else hasattr(lib, "task5_mse_tile")) and len(mine) > 0
    if use_tile:
        # :305-345 for all of this rank's SNR points in one call; point i draws its noise from stream 1 + i as below
        plan = lib.RxPlan(Nfft, T_Guard, N_symb, N_carrier, pilotCarriers, dataCarriers, pilotValues[:, 0], K, taps.shape[0],
                          Constellation, precision=precision)
        cdt = np.complex128 if precision == "fp64" else np.complex64
        for i0 in range(0, len(mine), 4096):
            r = mine[i0:i0 + 4096]
            MSEs[:, r] = np.asarray(lib.task5_mse_tile(plan, np.asarray(Tx).astype(cdt), taps, SNRs[r], seed=seed,
                                                       stream0=1 + r[0]))
        plan.close()
    else:
        for i in mine:                                                                          # :305
            snr = SNRs[i]
            Rx_i, _ = lib.Noise(float(snr), Tx, seed=seed, stream=1 + i)                        # :307
            Rx_i = c.conv_truncate(lib, Rx_i, H_tau)                                            # :308-309
            Xi = lib.OFDM_demodulator(np.asarray(Rx_i).reshape((Nfft + T_Guard, N_symb), order="F"), T_Guard)   # :310-311
            _, mse, _ = _estimates(lib, Xi, pilotValues, pilotCarriers, Nfft, N_carrier, amp_pilots, S,
                                   taps.shape[0], float(snr), H_freq)
            MSEs[:, i] = [mse["LS"], mse["MMSE"], mse["MP"], mse["OMP"]]                        # :341-344
    res["sweep"] = {"SNRs": SNRs, "estimators": ["LS", "MMSE", "MP", "OMP"], "MSEs": MSEs}
    return res


if __name__ == "__main__":
    c.cli(run, __doc__)
