"""Replay of `Task 1/Main_model.m` (plumbing loop-back, SURVEY.md 3.5 / BASELINE config 1)."""
from __future__ import annotations

import numpy as np

from . import common as c


def run(lib=None, Nfft=1024, N_carrier=400, Amount_OFDM_Frames=10, Amount_ODFM_SpF=5, Percent_pilot=25,
        Constellation="16QAM", nSTO=0, SNR_dB=None, seed=1):
    """T1/Main_model.m:6-108.  `SNR_dB=None` keeps the channel commented out as committed (:58-59)."""
    lib = lib or c.default_lib()
    T_Guard = Nfft // 8                                                            # :8
    N_symb = Amount_OFDM_Frames * Amount_ODFM_SpF                                  # :11
    allCarriers, pilotCarriers, dataCarriers = c.layout_percent(Nfft, N_carrier, Percent_pilot, tail=2)   # :16-24
    dict_, bps = lib.constellation_func(Constellation)                             # :30
    Size_Buffer = N_symb * len(dataCarriers) * bps                                 # :33
    input_bits = c.synthetic_bits(Size_Buffer, seed)                               # :34 (file_reader stand-in)
    TX_IQ, pad = lib.mapping(input_bits, Constellation)                            # :37
    amp_pilots = 2 * np.max(np.abs(dict_))                                         # :40-41
    OFDM_mapped_carriers = lib.OFDM_map_carriers(TX_IQ, N_symb, Nfft, dataCarriers, pilotCarriers, amp_pilots)  # :43
    Tx_OFDM_Signal_matrix = lib.OFDM_modulator(OFDM_mapped_carriers, T_Guard)      # :46
    Tx_OFDM_Signal = np.asarray(Tx_OFDM_Signal_matrix).ravel(order="F")            # :48
    if nSTO:
        Tx_OFDM_Signal = np.asarray(lib.add_STO(Tx_OFDM_Signal, nSTO)).ravel()     # :51-55
    Rx_OFDM_Signal = Tx_OFDM_Signal                                                # :69
    if SNR_dB is not None:
        Rx_OFDM_Signal, _ = lib.Noise(SNR_dB, Tx_OFDM_Signal, seed=seed, stream=0)  # :58-59
    Rx_OFDM_Signal = np.asarray(Rx_OFDM_Signal).reshape((Nfft + T_Guard, N_symb), order="F")   # :72
    RX_OFDM_mapped_carriers = lib.OFDM_demodulator(Rx_OFDM_Signal, T_Guard)        # :75
    RX_IQ = np.asarray(lib.get_payload(RX_OFDM_mapped_carriers, dataCarriers)).ravel(order="F")  # :86-88
    output_bits = np.asarray(lib.demapping(pad, RX_IQ, Constellation)).ravel()     # :94
    passed = bool(np.array_equal(output_bits, input_bits))                         # :99
    BER = lib.BER_func(input_bits, output_bits)                                    # :101 / :107
    return {"driver": "Task 1/Main_model.m", "passed": passed, "BER": float(BER), "pad": int(pad),
            "amount_pilots": len(pilotCarriers), "amount_data_carriers": len(dataCarriers),
            "_output_bits": output_bits, "_RX_IQ": RX_IQ, "_input_bits": input_bits}


if __name__ == "__main__":
    c.cli(run, __doc__)
