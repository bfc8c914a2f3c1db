#!/usr/bin/env python3
"""Host model of rx_symbols_wave_kernel's transform (ofdm-course_amd/csrc/ofdm_chain_wave.hip): the index algebra of
the 2048 = 32 x 64 decomposition against numpy's FFT, and both LDS transposes against the per-instruction banking of
MI355X_MICROARCH.md (ds_write_b64: four groups of 16 contiguous lanes over 32 four-byte banks; ds_read_b64: two
32-lane halves over 64 banks).  Run: python tools/lds_bank_check.py"""
import numpy as np

N = 2048
lane = np.arange(64)


def conflicts_write_b64(elem):          # elem[lane] = 8-byte element index
    worst = 1
    for g in range(4):
        banks = {}
        for l in range(16 * g, 16 * g + 16):
            for d in (0, 1):
                b = (2 * elem[l] + d) % 32
                banks.setdefault(b, set()).add(2 * elem[l] + d)
        worst = max(worst, max(len(s) for s in banks.values()))
    return worst


def conflicts_read_b64(elem):
    worst = 1
    for h in range(2):
        banks = {}
        for l in range(32 * h, 32 * h + 32):
            for d in (0, 1):
                b = (2 * elem[l] + d) % 64
                banks.setdefault(b, set()).add(2 * elem[l] + d)
        worst = max(worst, max(len(s) for s in banks.values()))
    return worst


def main():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    X = np.fft.fft(x)
    v = np.array([x[lane + 64 * j] for j in range(32)])            # v[j][lane]
    # 1. 32-point DFT over j (natural order result Z[kj][lane])
    Z = np.fft.fft(v, axis=0)
    # 2. twiddle
    kj = np.arange(32)[:, None]
    Z = Z * np.exp(-2j * np.pi * (kj * lane[None, :]) / N)
    out = np.zeros(N, complex)
    worst = {"T1w": 1, "T1r": 1, "T2w": 1, "T2r": 1}
    for r in range(4):
        tr = np.full(576, np.nan, complex)
        for c in range(8):                                         # T1 write: element 72 c + lane
            e = 72 * c + lane
            worst["T1w"] = max(worst["T1w"], conflicts_write_b64(e))
            tr[e] = Z[8 * r + c]
        u = np.zeros((8, 64), complex)
        for e_ in range(8):                                        # T1 read by lane 8 c' + l0'
            e = 72 * (lane >> 3) + (lane & 7) + 8 * e_
            worst["T1r"] = max(worst["T1r"], conflicts_read_b64(e))
            u[e_] = tr[e]
        u = np.fft.fft(u, axis=0)                                  # radix-8 over l1 -> ka
        ka = np.arange(8)[:, None]
        u = u * np.exp(-2j * np.pi * ka * (lane & 7)[None, :] / 64)
        tr[:] = np.nan
        for t in range(8):                                         # T2 write: 65 l0' + 8 c' + ka
            e = 65 * (lane & 7) + 8 * (lane >> 3) + t
            worst["T2w"] = max(worst["T2w"], conflicts_write_b64(e))
            tr[e] = u[t]
        w = np.zeros((8, 64), complex)
        for e_ in range(8):                                        # T2 read by lane 8 c'' + ka'': 65 e + lane
            e = 65 * e_ + lane
            worst["T2r"] = max(worst["T2r"], conflicts_read_b64(e))
            w[e_] = tr[e]
        w = np.fft.fft(w, axis=0)                                  # radix-8 over l0 -> kb
        for kb in range(8):
            k = 32 * ((lane & 7) + 8 * kb) + 8 * r + (lane >> 3)
            out[k] = w[kb]
    err = np.max(np.abs(out - X)) / np.max(np.abs(X))
    print("max rel error of the decomposition vs numpy fft:", err)
    print("worst-case ways per instruction:", worst)
    assert err < 1e-12 and all(v == 1 for v in worst.values())
    print("OK: index algebra exact, both transposes conflict-free")


if __name__ == "__main__":
    main()
