"""Replay of `Task 2/Main_model_Task_2.m`: scrambled and plain loop-back, and the PAPR / CCDF study of :69-97
(whole-signal PAPR, sliding-window PAPR, CCDF curves of the plain and the scrambled signal; the plot itself stays
with the caller)."""
from __future__ import annotations

import numpy as np

from . import common as c


def run(lib=None, Nfft=1024, N_carrier=400, Amount_OFDM_Frames=10, Amount_ODFM_SpF=5, Percent_pilot=1,
        Constellation="16QAM", seed=1, input_bits=None):
    """T2/Main_model_Task_2.m:6-162.  `input_bits`: the payload `file_reader` would return (:32; default = seeded
    synthetic bits); at least Size_Buffer bits, the first Size_Buffer are used like `input_bits(1:Size_Buffer)`."""
    lib = lib or c.default_lib()
    T_Guard = Nfft // 8
    N_symb = Amount_OFDM_Frames * Amount_ODFM_SpF
    allCarriers, pilotCarriers, dataCarriers = c.layout_percent(Nfft, N_carrier, Percent_pilot, tail=2)   # :16-24
    dict_, bps = lib.constellation_func(Constellation)                             # :28
    Size_Buffer = N_symb * len(dataCarriers) * bps
    input_bits = c.payload_bits(Size_Buffer, seed, input_bits)                     # :32
    sc_bits = c.scramble_per_frame(lib, "Scrambler", input_bits, Amount_OFDM_Frames)   # :36-51
    TX_IQ, _ = lib.mapping(input_bits, Constellation)                              # :53
    sc_TX_IQ, pad = lib.mapping(sc_bits, Constellation)                            # :54
    amp_pilots = 2 * np.max(np.abs(dict_))                                         # :57-58
    out, papr = {}, {}
    for tag, iq in (("plain", TX_IQ), ("scrambled", sc_TX_IQ)):
        X = lib.OFDM_map_carriers(iq, N_symb, Nfft, dataCarriers, pilotCarriers, amp_pilots)   # :60-61
        tx = np.asarray(lib.OFDM_modulator(X, T_Guard)).ravel(order="F")           # :64-68
        PAPRs = lib.calculate_window_PAPR(tx, Nfft)                                # :78 / :81
        PAPR_ccdf, CCDF = lib.calculateCCDF(PAPRs)                                 # :79 / :82
        papr[tag] = {"PAPR_dB": float(lib.calculatePAPR(tx)),                      # :72-73
                     "PAPR_ccdf": np.asarray(PAPR_ccdf), "CCDF": np.asarray(CCDF), "_PAPRs": np.asarray(PAPRs)}
        rx = tx.reshape((Nfft + T_Guard, N_symb), order="F")                       # :103-108
        Xr = lib.OFDM_demodulator(rx, T_Guard)                                     # :111 / :116
        RX_IQ = np.asarray(lib.get_payload(Xr, dataCarriers)).ravel(order="F")     # :113-119
        out[tag] = np.asarray(lib.demapping(pad, RX_IQ, Constellation)).ravel()    # :122-123
    dsc_bits = c.scramble_per_frame(lib, "DeScrambler", out["scrambled"], Amount_OFDM_Frames)   # :126-138
    return {"driver": "Task 2/Main_model_Task_2.m",
            "passed": bool(np.array_equal(out["plain"], input_bits)),              # :140
            "passed_scrambled": bool(np.array_equal(dsc_bits, input_bits)),        # :152
            "BER": float(lib.BER_func(input_bits, out["plain"])),
            "BER_scrambled": float(lib.BER_func(input_bits, dsc_bits)),
            "ones_fraction_plain": float(np.mean(input_bits)), "ones_fraction_scrambled": float(np.mean(sc_bits)),
            "papr": papr,
            "_sc_bits": sc_bits, "_dsc_bits": dsc_bits, "_input_bits": input_bits}


if __name__ == "__main__":
    c.cli(run, __doc__)
