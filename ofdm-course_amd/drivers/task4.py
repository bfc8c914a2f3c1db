"""Replay of `Task 4/Main_model_Task_4.m`: STO / CFO / multipath, coarse + fine synchronisation, spline
channel estimate, equalise, BER / MER (BASELINE config 3 shape)."""
from __future__ import annotations

import warnings

import numpy as np

from . import common as c

CHANNEL_TAPS = np.array([[0, 1.0], [4, 0.6], [10, 0.3]])                           # T4/Main_model_Task_4.m:257-261


def run(lib=None, Nfft=1024, N_carrier=400, Amount_OFDM_Frames=10, Amount_ODFM_SpF=5, Percent_pilot=15,
        Constellation="16QAM", noise_desync=0, time_desync=0, freq_desync=0, mp_desync=0, SNR_dB=25,
        Time_Delay=None, Freq_Shift=None, channel_taps=None, seed=1):
    """T4/Main_model_Task_4.m:6-376.  Flags default to the committed values (:81-87, all off).  `Time_Delay` /
    `Freq_Shift` default to seeded draws with the ranges of :101 / :108."""
    lib = lib or c.default_lib()
    T_Guard = Nfft // 8
    N_symb = Amount_OFDM_Frames * Amount_ODFM_SpF
    allCarriers, pilotCarriers, dataCarriers = c.layout_percent(Nfft, N_carrier, Percent_pilot, tail=2)   # :14-21
    dict_, bps = lib.constellation_func(Constellation)
    amp_pilots = 4 / 3 * np.max(np.abs(dict_))                                     # :26-27
    pilotValues = c.alternating_pilots(amp_pilots, len(pilotCarriers), N_symb)     # :28-31
    Size_Buffer = N_symb * len(dataCarriers) * bps
    input_bits = c.synthetic_bits(Size_Buffer, seed)                               # :39
    sc_bits = c.scramble_per_frame(lib, "Scrambler", input_bits, Amount_OFDM_Frames)   # :48-58
    TX_IQ, pad = lib.mapping(sc_bits, Constellation)                               # :60
    X = lib.OFDM_map_carriers(TX_IQ, N_symb, Nfft, dataCarriers, pilotCarriers, pilotValues)   # :63
    Rx = np.asarray(lib.OFDM_modulator(X, T_Guard)).ravel(order="F")               # :66-68, :89

    rng = np.random.Generator(np.random.PCG64([seed, 4]))
    info = {}
    if noise_desync:
        Rx, _ = lib.Noise(SNR_dB, Rx, seed=seed, stream=0)                         # :95
    if time_desync:
        if Time_Delay is None:
            Time_Delay = int(rng.integers(0, Nfft + T_Guard + 1))                  # :101
        Rx = lib.add_STO(np.asarray(Rx).ravel(), Time_Delay)                       # :103
        info["Time_Delay"] = int(Time_Delay)
    if freq_desync:
        if Freq_Shift is None:
            Freq_Shift = float(rng.integers(0, 31)) + (rng.random() - 0.5)         # :108
        Rx = lib.add_CFO(np.asarray(Rx).ravel(), Freq_Shift, Nfft)                 # :110
        info["Freq_Shift"] = float(Freq_Shift)
    if mp_desync:
        taps = CHANNEL_TAPS if channel_taps is None else np.asarray(channel_taps)
        H_tau, _ = lib.get_MP_channel_resp(taps, Nfft)                             # :262
        Rx = c.conv_truncate(lib, Rx, H_tau)                                       # :263-264

    if time_desync or freq_desync:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            _, TgPosition, FreqOffset = lib.AutoCorrFunction(np.asarray(Rx).ravel(), T_Guard, Nfft)   # :278
        info.update(TgPosition=int(TgPosition), FreqOffset=float(FreqOffset), acf_fallback=bool(w))
        if time_desync:
            Rx = lib.add_STO(np.asarray(Rx).ravel(), TgPosition)                   # :292
            Rx = lib.add_STO(np.asarray(Rx).ravel(), -(Nfft + T_Guard))            # :294
    if freq_desync:
        Rx = lib.add_CFO(np.asarray(Rx).ravel(), -FreqOffset, Nfft)                # :301
        Rx, e_IFO = lib.remove_IFO(np.asarray(Rx).ravel(), Nfft)                   # :303
        info["e_IFO"] = float(e_IFO)

    rx = np.asarray(Rx).reshape((Nfft + T_Guard, N_symb), order="F")               # :308
    Xr = lib.OFDM_demodulator(rx, T_Guard)                                         # :310
    if time_desync or freq_desync:
        Xr = lib.fine_sync(Xr, pilotCarriers, pilotValues, time_desync, freq_desync, variant="T4")   # :314
    if mp_desync:
        H_est, Hest_at_pilots = lib.estimate_channel(Xr, allCarriers, pilotCarriers, pilotValues)    # :318
        Xr = lib.equalize_signal(Xr, H_est, N_carrier)                             # :334
        info["_H_est"] = np.asarray(H_est)
    RX_IQ = np.asarray(lib.get_payload(Xr, dataCarriers)).ravel(order="F")         # :340-341
    output_bits = np.asarray(lib.demapping(pad, RX_IQ, Constellation)).ravel()     # :347
    dsc_bits = c.scramble_per_frame(lib, "DeScrambler", output_bits, Amount_OFDM_Frames)   # :354-364
    BER = float(lib.BER_func(input_bits, dsc_bits))                                # :366
    MER = float(lib.MER_func(RX_IQ, Constellation))                                # :374
    return {"driver": "Task 4/Main_model_Task_4.m", "passed": bool(BER < 0.2), "BER": BER, "MER_dB": MER,   # :367
            "SNR_dB": SNR_dB, **info, "_dsc_bits": dsc_bits, "_RX_IQ": RX_IQ, "_input_bits": input_bits}


if __name__ == "__main__":
    c.cli(run, __doc__)
