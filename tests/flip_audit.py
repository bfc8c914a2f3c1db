"""SURVEY.md section 8c rule for fp32 demapper parity: "decisions may differ only for points within 1e-4 of a decision
boundary -- count and report them".  TEST INFRASTRUCTURE (flips were counted, not audited).

demapping.m:7-12 decides the constellation point nearest to the equalised IQ (first minimum on ties).  For every QAM symbol
whose decided bits differ between the device (fp32) and the oracle (float64), `decision_flip_audit` takes the ORACLE's
equalised IQ z of that symbol and requires it to lie within `band` x (level spacing) of the boundary between the two
decisions, i.e. of the perpendicular bisector of D[want], D[got]:

        0 <= (|z - D[got]|^2 - |z - D[want]|^2) / (2 |D[got] - D[want]|)  <  band * d_min,

d_min = the smallest distance between two constellation points (the level spacing of a square QAM).  A differing decision
that is not such a near-tie fails the test, however few there are.  Returns (flipped symbols, largest relative distance)."""
import numpy as np


def _codes(bits, bps):
    b = np.asarray(bits, dtype=np.int64).reshape(-1, bps)
    return b @ (1 << np.arange(bps - 1, -1, -1))


def decision_flip_audit(oracle, got_bits, want_bits, iq, Constellation, band=1e-4, what=""):
    D, bps = oracle.constellation_func(Constellation)
    D = np.asarray(D, dtype=np.complex128)
    got, want = _codes(got_bits, bps), _codes(want_bits, bps)
    z = np.asarray(iq, dtype=np.complex128).ravel()
    assert got.size == want.size == z.size, (got.size, want.size, z.size)
    diff = np.flatnonzero(got != want)
    if diff.size == 0:
        return 0, 0.0
    dd = np.abs(D[:, None] - D[None, :])
    dmin = float(np.min(dd[dd > 0]))
    zg, dg, dw = z[diff], D[got[diff]], D[want[diff]]
    dist = (np.abs(zg - dg) ** 2 - np.abs(zg - dw) ** 2) / (2 * np.abs(dg - dw)) / dmin
    worst = int(np.argmax(dist))
    assert np.all(dist >= -1e-12), f"{what}: the oracle's own decision is not the nearest point (symbol {diff[np.argmin(dist)]})"
    assert dist[worst] < band, (
        f"{what}: {diff.size} differing decisions; symbol {diff[worst]} decided {got[diff][worst]} instead of {want[diff][worst]} "
        f"with the oracle's IQ {zg[worst]:.6f} at {dist[worst]:.3g} level spacings from the boundary -- not a near-tie (band {band:g})")
    return int(diff.size), float(dist[worst])
