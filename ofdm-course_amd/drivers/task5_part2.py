"""Replay of `Task 5/Task5_part2.m`: NMSE and BER of LS / MMSE / MP / OMP equalisation against the pilot
count, Monte-Carlo over channel draws (SURVEY.md 3.3 -- the shape of the north-star metric).

`lteFadingChannel` is a closed-source toolbox call; `common.fading_taps` draws a static tap-delay line
from the same public delay-profile table instead (documented deviation).  The (kk, jj) pairs are
independent: `rank` / `world` deal them round-robin like `sweep.tiles_for_rank`, and the per-scenario sums
returned under "_sums" add up across ranks (one all-reduce).
"""
from __future__ import annotations

import numpy as np

from . import common as c

ESTIMATORS = ("LS", "MMSE", "MP", "OMP")


def scenario_combs(N_carrier=1024, lo=4, hi=256):
    """:13-17 -- the combs with distinct pilot counts floor(N_carrier/comb), first comb of each count."""
    combs = np.arange(lo, hi + 1)
    amounts = N_carrier // combs
    _, ia = np.unique(amounts, return_index=True)                                   # first occurrence, like unique()
    combs = combs[np.sort(ia)]
    return combs, N_carrier // combs


def random_pilot_layout(Nfft, N_carrier, amount_pilots, seed):
    """:58-64 -- `sort(randperm(N_carrier, Np))` and `pilot_step = pilotCarriers(3) - pilotCarriers(2)`; a step of 1
    switches to the 100 % rule like the regular mask (:67-75).  The draw is PCG64(seed) (MATLAB's `randperm` stream
    is not restated: any sorted subset without repetition is a valid mask).  Returns the layout and pilot_step."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pilotCarriers = np.sort(rng.permutation(N_carrier)[:int(amount_pilots)] + 1).astype(np.float64)
    pilot_step = int(pilotCarriers[2] - pilotCarriers[1])
    if pilot_step == 1:
        return c.layout_percent(Nfft, N_carrier, 100, tail=1) + (1,)
    allCarriers = np.arange(1, Nfft + 1, dtype=np.float64)
    dataCarriers = allCarriers[:N_carrier][~np.isin(allCarriers[:N_carrier], pilotCarriers)]
    return allCarriers, pilotCarriers, dataCarriers, pilot_step


def run(lib=None, Nfft=4096, N_carrier=1024, Amount_OFDM_Frames=2, Amount_ODFM_SpF=7, Constellation="16QAM",
        monteCarloRuns=100, SNR_dB=20, combs=None, DelayProfile="EPA", SamplingRate=4e7, seed=5, rank=0, world=1,
        reg_pilot=1, Nps=None, batched=None, precision="fp64"):
    """T5/Task5_part2.m:4-320.  reg_pilot = 1 (:12): one scenario per comb; reg_pilot = 0: one scenario per pilot
    count `Nps` (:21, default = the regular study's counts) on a random mask, dictionary = all Nfft delays (:181-184).

    batched: run this rank's realisations of a scenario as ONE device-resident tile (`lib.task5_part2_tile`: all four
    estimators, NMSE sums and BER counters on the device) instead of call by call; default = whenever the library has
    that entry (the HIP library does, the oracle adapter of the tests does not).  precision: of the batched tile."""
    lib = lib or c.default_lib()
    T_Guard = Nfft // 8
    N_symb = Amount_OFDM_Frames * Amount_ODFM_SpF
    if combs is None:
        combs, _ = scenario_combs(N_carrier)
    combs = np.asarray(combs, dtype=int)
    if not reg_pilot:
        Nps = np.asarray(N_carrier // combs if Nps is None else Nps, dtype=int)
        combs = np.zeros(len(Nps), dtype=int)                                       # scenario axis = pilot counts (:42-44)
    rng = np.random.Generator(np.random.PCG64(seed))                                # :23 rng(5)
    channel_Seeds = rng.integers(1, 2 ** 16 + 1, size=(len(combs), monteCarloRuns))  # :24
    dict_, bps = lib.constellation_func(Constellation)
    amp_pilots = 2 * np.max(np.abs(dict_))                                          # :84-85
    nmse_sum = np.zeros((4, len(combs)))
    err_sum = np.zeros((4, len(combs)), dtype=np.int64)
    bit_sum = np.zeros(len(combs), dtype=np.int64)
    runs = np.zeros(len(combs), dtype=np.int64)
    for kk, comb in enumerate(combs):                                               # :46
        if reg_pilot:
            allCarriers, pilotCarriers, dataCarriers = c.layout_comb(Nfft, N_carrier, int(comb))    # :48-79
            K = int(np.ceil(Nfft / comb))                                           # :183
        else:
            allCarriers, pilotCarriers, dataCarriers, _ = random_pilot_layout(Nfft, N_carrier, Nps[kk], [seed, 7, kk])
            K = Nfft                                                                # :181 F = dftmtx(Nfft), all columns
        pilotValues = c.alternating_pilots(amp_pilots, len(pilotCarriers), N_symb)  # :86-91
        Size_Buffer = N_symb * len(dataCarriers) * bps
        input_bits = c.synthetic_bits(Size_Buffer, [seed, kk])                      # :96 (scrambler commented out, :99-115)
        TX_IQ, pad = lib.mapping(input_bits, Constellation)                         # :119
        X = lib.OFDM_map_carriers(TX_IQ, N_symb, Nfft, dataCarriers, pilotCarriers, pilotValues)   # :123
        Tx = np.asarray(lib.OFDM_modulator(X, T_Guard)).ravel(order="F")            # :130-132
        Tx_noised, _ = lib.Noise(SNR_dB, Tx, seed=seed, stream=kk)                  # :134 (noise BEFORE the channel)
        mine = [jj for jj in range(monteCarloRuns) if (kk * monteCarloRuns + jj) % world == rank]
        # the tile needs a payload (the 100 % pilot rule of :67-75 leaves none), two pilots and a dictionary whose batch-OMP state
        # fits the LDS (K <= 2048 atoms); every other scenario runs call by call on the same library
        tile_ok = len(dataCarriers) > 0 and len(pilotCarriers) >= 2 and K <= 2048
        use_tile = (batched if batched is not None else hasattr(lib, "task5_part2_tile")) and len(mine) > 0 and tile_ok
        if use_tile:
            # :148-304 for all of this rank's realisations of scenario kk in one call
            from .. import frames as fr
            taps_l = [c.fading_taps(DelayProfile, SamplingRate, channel_Seeds[kk, jj]) for jj in mine]   # :150-152
            plan = lib.RxPlan(Nfft, T_Guard, N_symb, N_carrier, pilotCarriers, dataCarriers, pilotValues[:, 0], K,
                              taps_l[0].shape[0], Constellation, precision=precision)
            cdt = np.complex128 if precision == "fp64" else np.complex64
            out = lib.task5_part2_tile(plan, np.asarray(Tx_noised).astype(cdt), taps_l, SNR_dB,
                                       fr.pack_bits(np.asarray(input_bits)[None, :])[0])
            nmse_sum[:, kk] += np.asarray(out["nmse"]).sum(axis=1)
            err_sum[:, kk] += np.asarray(out["errors"]).astype(np.int64).sum(axis=1)
            bit_sum[kk] += input_bits.size * len(mine)
            runs[kk] += len(mine)
            plan.close()
            continue
        S = lib.sensing_matrix(pilotCarriers, Nfft, K)                              # :181-189 closed form
        for jj in range(monteCarloRuns):                                            # :148
            if (kk * monteCarloRuns + jj) % world != rank:
                continue
            taps = c.fading_taps(DelayProfile, SamplingRate, channel_Seeds[kk, jj])  # :150-152
            h_t, H_f = lib.get_MP_channel_resp(taps, Nfft)                          # :154-155
            rx = c.conv_truncate(lib, Tx_noised, h_t)
            rx = rx.reshape((Nfft + T_Guard, N_symb), order="F")                    # :169
            Xr = lib.OFDM_demodulator(rx, T_Guard)                                  # :172
            h_full = np.zeros(N_carrier, dtype=np.complex128)
            h_full[:min(len(h_t), N_carrier)] = np.asarray(h_t)[:N_carrier]         # :176 h_t(1:N_carrier).'
            H = {"LS": lib.LS_CE(Xr, pilotValues, pilotCarriers, N_carrier),                         # :174
                 "MMSE": lib.MMSE_CE(Xr, pilotValues, pilotCarriers, Nfft, N_carrier, h_full, SNR_dB)}   # :177
            Y = np.asarray(lib.get_payload(np.asarray(Xr)[:, :1], pilotCarriers)).ravel() / pilotValues[:, 0]   # :190
            n_paths = taps.shape[0]                                                 # length(info.PathSampleDelays)
            H["MP"], _ = lib.MP_estimate(Y, S, Nfft, n_paths)                       # :192
            H["OMP"], _, _ = lib.OMP_estimate(Y, S, Nfft, n_paths, SNR_dB)          # :193
            for e, name in enumerate(ESTIMATORS):
                nmse_sum[e, kk] += c.mse_row(H_f, H[name], N_carrier)               # :202-205
                eq = lib.equalize_signal(Xr, H[name], N_carrier)                    # :269-272
                RX_IQ = np.asarray(lib.get_payload(eq, dataCarriers)).ravel(order="F")   # :281-282
                output_bits = lib.demapping(pad, RX_IQ, Constellation)              # :284
                err_sum[e, kk] += lib.BER_func(input_bits, np.asarray(output_bits).ravel(), return_count=True)   # :286
            bit_sum[kk] += input_bits.size
            runs[kk] += 1
    safe = np.maximum(runs, 1)
    return {"driver": "Task 5/Task5_part2.m", "reg_pilot": int(bool(reg_pilot)), "combs": combs,
            "amounts_pilots": N_carrier // combs if reg_pilot else Nps,
            "estimators": list(ESTIMATORS), "monteCarloRuns": monteCarloRuns, "SNR_dB": SNR_dB,
            "NMSEs": nmse_sum / safe, "BERs": err_sum / np.maximum(bit_sum, 1),     # :309-318 (mean over jj)
            "_sums": {"nmse": nmse_sum, "errors": err_sum, "bits": bit_sum, "runs": runs}}


if __name__ == "__main__":
    c.cli(run, __doc__)
