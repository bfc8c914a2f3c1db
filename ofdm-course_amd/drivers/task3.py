"""Replay of `Task 3/Main_model_Task_3.m`: single impaired run (:6-190) + BER(SNR) sweep over the four
constellations (:192-279)."""
from __future__ import annotations

import numpy as np

from . import common as c

CHANNEL_TAPS = np.array([[0, 1.0], [2, 0.4], [4, 0.01]])                          # T3/Main_model_Task_3.m:116-120


def _tx(lib, Constellation, N_symb, Nfft, T_Guard, dataCarriers, pilotCarriers, Amount_OFDM_Frames, seed):
    """:27-66 / :199-232 -- bits -> per-frame Scrambler -> mapping -> OFDM_map_carriers -> OFDM_modulator."""
    dict_, bps = lib.constellation_func(Constellation)
    Size_Buffer = N_symb * len(dataCarriers) * bps
    input_bits = c.synthetic_bits(Size_Buffer, seed)
    sc_bits = c.scramble_per_frame(lib, "Scrambler", input_bits, Amount_OFDM_Frames)
    TX_IQ, pad = lib.mapping(sc_bits, Constellation)
    amp_pilots = 4 / 3 * np.max(np.abs(dict_))                                    # :53-54
    X = lib.OFDM_map_carriers(TX_IQ, N_symb, Nfft, dataCarriers, pilotCarriers, amp_pilots)
    Tx_OFDM_Signal = np.asarray(lib.OFDM_modulator(X, T_Guard)).ravel(order="F")
    return input_bits, pad, Tx_OFDM_Signal


def _rx(lib, Rx_OFDM_Signal, Nfft, T_Guard, N_symb, dataCarriers, pad, Constellation, Amount_OFDM_Frames):
    """:128-165 / :241-262 -- reshape -> OFDM_demodulator -> get_payload -> demapping -> per-frame DeScrambler."""
    rx = np.asarray(Rx_OFDM_Signal).reshape((Nfft + T_Guard, N_symb), order="F")
    X = lib.OFDM_demodulator(rx, T_Guard)
    RX_IQ = np.asarray(lib.get_payload(X, dataCarriers)).ravel(order="F")
    output_bits = np.asarray(lib.demapping(pad, RX_IQ, Constellation)).ravel()
    return RX_IQ, c.scramble_per_frame(lib, "DeScrambler", output_bits, Amount_OFDM_Frames)


def run(lib=None, Nfft=1024, N_carrier=400, Amount_OFDM_Frames=10, Amount_ODFM_SpF=5, Percent_pilot=15,
        Constellation="16QAM", noise_desync=0, time_desync=0, freq_desync=0, mp_desync=1, SNR_dB=25,
        Time_Delay=37, Freq_Shift=100, SNRs=None, Constellations=("BPSK", "QPSK", "8PSK", "16QAM"), seed=1):
    """T3/Main_model_Task_3.m.  Flags default to the committed values (:81-84: only the multipath is on)."""
    lib = lib or c.default_lib()
    T_Guard = Nfft // 8
    N_symb = Amount_OFDM_Frames * Amount_ODFM_SpF
    _, pilotCarriers, dataCarriers = c.layout_percent(Nfft, N_carrier, Percent_pilot, tail=2)          # :16-24
    input_bits, pad, Tx = _tx(lib, Constellation, N_symb, Nfft, T_Guard, dataCarriers, pilotCarriers,
                              Amount_OFDM_Frames, seed)
    Rx = Tx                                                                        # :86
    if noise_desync:
        Rx, _ = lib.Noise(SNR_dB, Rx, seed=seed, stream=0)                         # :91-93
    if time_desync:
        Rx = lib.add_STO(np.asarray(Rx).ravel(), Time_Delay)                       # :96-100
    if freq_desync:
        Rx = lib.add_CFO(np.asarray(Rx).ravel(), Freq_Shift, Nfft)                 # :103-107
    if mp_desync:
        H_tau, _ = lib.get_MP_channel_resp(CHANNEL_TAPS, Nfft)                     # :121
        Rx = c.conv_truncate(lib, Rx, H_tau)                                       # :123-124
    RX_IQ, dsc_bits = _rx(lib, Rx, Nfft, T_Guard, N_symb, dataCarriers, pad, Constellation, Amount_OFDM_Frames)
    res = {"driver": "Task 3/Main_model_Task_3.m", "passed": bool(np.array_equal(input_bits, dsc_bits)),   # :177
           "BER": float(lib.BER_func(input_bits, dsc_bits)),                       # :185
           "MER_dB": float(lib.MER_func(RX_IQ, Constellation)),                    # :186
           "SNR_dB": SNR_dB, "_dsc_bits": dsc_bits, "_RX_IQ": RX_IQ}

    SNRs = np.arange(0, 30.5, 0.5) if SNRs is None else np.asarray(SNRs, dtype=float)   # :192
    BERs = np.zeros((len(Constellations), len(SNRs)))
    for ci, const in enumerate(Constellations):                                    # :196
        in_bits, pad_c, Tx_c = _tx(lib, const, N_symb, Nfft, T_Guard, dataCarriers, pilotCarriers,
                                   Amount_OFDM_Frames, seed)
        for i, snr in enumerate(SNRs):                                             # :237
            Rx_c, _ = lib.Noise(float(snr), Tx_c, seed=seed, stream=1 + ci * len(SNRs) + i)   # :239
            _, dsc = _rx(lib, Rx_c, Nfft, T_Guard, N_symb, dataCarriers, pad_c, const, Amount_OFDM_Frames)
            BERs[ci, i] = lib.BER_func(in_bits, dsc)                               # :264
    res["sweep"] = {"SNRs": SNRs, "Constellations": list(Constellations), "BERs": BERs}
    return res


if __name__ == "__main__":
    c.cli(run, __doc__)
