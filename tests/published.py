"""The numbers the reference itself publishes for this path, as data, plus the
assertions that tie a driver replay to them.  Shared by tests/test_oracle_published.py (CPU oracle) and
tests/test_gpu_published.py (the same replays on the HIP library).  TEST INFRASTRUCTURE.

Every value is read off a README table or graph of /root/reference, so the tolerances have two parts: the
reading error of a curve (about 10 %) and the spread of ONE noise realisation per SNR point (the reference
draws each point once, `Main_model_Task_3.m:239`, `Main_model_Task_5.m:307`; MATLAB's `normrnd` stream is
not reproducible here, so the replays use their own Philox realisations).
"""
from __future__ import annotations

import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# Task 3/graphs/ber(snr).png (Task 3/README.md:57-60): BER after the descrambler, 50 symbols x 332 data carriers
BER_POINTS = {
    "BPSK": {0.0: 0.043, 2.0: 0.0095, 4.0: 0.0013},
    "QPSK": {0.0: 0.17, 2.0: 0.08, 4.0: 0.023, 6.0: 0.0036},
    "8PSK": {0.0: 0.34, 5.0: 0.135, 10.0: 0.010},
    "16QAM": {0.0: 0.43, 5.0: 0.26, 10.0: 0.054, 12.0: 0.016, 14.0: 0.0022},
}
# Task 5/graphs/mse(snr), comb1.png (Task 5/README.md:32-39)
MSE_POINTS = {
    "LS": {0.0: 0.70, 5.0: 0.198, 10.0: 0.067, 30.0: 0.0007},
    "MP": {0.0: 0.022, 10.0: 0.025, 30.0: 0.024},
    "OMP": {0.0: 0.007, 10.0: 0.003, 30.0: 0.003},
}
# the MMSE curve of the same graph: NOT what the committed MMSE_CE.m produces
MSE_MMSE_PUBLISHED = {0.0: 0.175, 2.0: 0.095, 5.0: 0.044, 10.0: 0.017}
# Task 4/README.md:181-183
MER_TABLE_PUBLISHED = {"linear": 60.0, "cubic": 108.0, "spline": 130.0}
# Task 2/README.md:54, :70-71 (image payload)
PAPR_PUBLISHED = {"plain_dB": (22.0, 23.0), "scrambled_dB": 10.0, "ccdf_0p02_plain": 22.0, "ccdf_0p02_scrambled": 10.0}


def eagle_bits():
    """The reference's payload: `file_reader('eagle.tiff', ...)` (fixture made by tests/golden/make_eagle_bits.py)."""
    g = np.load(os.path.join(HERE, "golden", "eagle_bits.npz"))
    return np.unpackbits(g["packed"])[: int(g["n_bits"])]


def ber_tolerance(p, n_bits):
    """Accepted ratio band for a BER read off the log plot: 15 % reading error plus 4 sigma of the count, with the
    descrambler tripling every channel error (DeScrambler.m:8-13: errors come in correlated triples, so the count has
    a third of the independent events)."""
    sigma = np.sqrt(max(p, 1e-12) * 3.0 / n_bits)
    return 0.15 * p + 4.0 * sigma


def check_ber_sweep(sweep, n_bits_per_bps):
    """sweep = drivers.task3.run(...)["sweep"] computed on SNRs containing every key of BER_POINTS."""
    snrs = list(np.asarray(sweep["SNRs"], dtype=float))
    report = []
    for ci, name in enumerate(sweep["Constellations"]):
        bps = {"BPSK": 1, "QPSK": 2, "8PSK": 3, "16QAM": 4}[name]
        for snr, want in BER_POINTS[name].items():
            got = float(sweep["BERs"][ci, snrs.index(snr)])
            tol = ber_tolerance(want, n_bits_per_bps * bps)
            report.append((name, snr, want, got, tol))
            assert abs(got - want) <= tol, (name, snr, want, got, tol)
    return report


def check_mse_sweep(sweep):
    """sweep = drivers.task5.run(...)["sweep"]; LS, MP and OMP against the published curve.  LS is one noise draw
    of channel-coloured noise over 1024 carriers (the graph itself wiggles by +-12 % between neighbouring points); MP / OMP sit on
    their noiseless floors (0.02373 / 0.002916, tests/test_oracle_kat.py) with wiggles of the size the graph shows."""
    snrs = list(np.asarray(sweep["SNRs"], dtype=float))
    row = {n: i for i, n in enumerate(sweep["estimators"])}
    for snr, want in MSE_POINTS["LS"].items():
        got = float(sweep["MSEs"][row["LS"], snrs.index(snr)])
        assert abs(got - want) <= 0.20 * want + 1e-4, ("LS", snr, want, got)
    for snr, want in MSE_POINTS["MP"].items():
        got = float(sweep["MSEs"][row["MP"], snrs.index(snr)])
        assert abs(got - want) <= 0.008, ("MP", snr, want, got)
    for snr, want in MSE_POINTS["OMP"].items():
        got = float(sweep["MSEs"][row["OMP"], snrs.index(snr)])
        assert abs(got - want) <= (0.012 if snr < 5 else 0.002), ("OMP", snr, want, got)


def check_papr(papr):
    """papr = drivers.task2.run(..., input_bits=eagle_bits())["papr"]."""
    lo, hi = PAPR_PUBLISHED["plain_dB"]
    assert lo - 1.0 <= papr["plain"]["PAPR_dB"] <= hi + 1.0, papr["plain"]["PAPR_dB"]
    assert abs(papr["scrambled"]["PAPR_dB"] - PAPR_PUBLISHED["scrambled_dB"]) <= 2.0, papr["scrambled"]["PAPR_dB"]
    for tag, want in (("plain", PAPR_PUBLISHED["ccdf_0p02_plain"]), ("scrambled", PAPR_PUBLISHED["ccdf_0p02_scrambled"])):
        x, c = np.asarray(papr[tag]["PAPR_ccdf"]), np.asarray(papr[tag]["CCDF"])
        at = float(x[np.argmin(np.abs(c - 0.02))])
        assert abs(at - want) <= 1.0, (tag, at, want)
