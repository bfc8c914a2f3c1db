#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded inputs + expected outputs of the ORACLE (the CPU
restatement of the .m reference) for every function on the hot path.

The reference is MATLAB and ships no vectors, so these fixtures pin the
oracle against regressions and give the GPU parity tests fixed data; they are not MATLAB output.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ofdm_oracle as o  # noqa: E402

TAPS6 = np.array([[0, 1], [4, .8], [10, .6], [15, .4], [21, .2], [25, .1]])


def crandn(rng, *shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2)


def main():
    rng = np.random.default_rng(20240501)
    # ---- bit side
    bits = rng.integers(0, 2, 6 * 200).astype(np.uint8)
    g = dict(bits=bits, register=np.array(o.DEFAULT_REGISTER, np.uint8))
    g["scrambled"], g["scr_reg"] = o.Scrambler(o.DEFAULT_REGISTER, bits)
    g["descrambled"], g["dsc_reg"] = o.DeScrambler(o.DEFAULT_REGISTER, g["scrambled"])
    for name in ["BPSK", "QPSK", "8PSK", "16QAM", "64QAM", "256QAM"]:
        D, bps = o.constellation_func(name)
        iq, _ = o.mapping(bits[: 40 * bps], name)
        noisy = iq + 0.3 * crandn(rng, iq.size)
        g[f"dict_{name}"] = D
        g[f"map_{name}"] = iq
        g[f"noisy_{name}"] = noisy
        g[f"demap_{name}"] = o.demapping(-1, noisy, name)
    np.savez_compressed(os.path.join(HERE, "bits.npz"), **g)

    # ---- modem + channel + sync (T4 shape, Nfft 256).  remove_IFO's absolute 0.77 threshold is fragile
    #      (a payload-dependent leakage bin can trip it), so take the first sub-seed whose chain decodes
    #      symbols 2..S error-free -- the fixture should exercise the working regime of T4.
    nfft, nc, tg, ns = 256, 100, 32, 8
    pc, dc = o.pilot_layout_percent(nfft, nc, 15, 2)
    D, bps = o.constellation_func("16QAM")
    amp = 4 / 3 * np.max(np.abs(D))
    pv = np.repeat(np.where(np.arange(len(pc)) % 2 == 0, amp, -amp).astype(complex)[:, None], ns, axis=1)
    h, H = o.get_MP_channel_resp(np.array([[0, 1], [3, .5], [7, .25]]), nfft)
    for sub in range(20):
        r2 = np.random.default_rng([20240501, sub])
        pay_bits = r2.integers(0, 2, len(dc) * ns * bps).astype(np.uint8)
        iq, _ = o.mapping(pay_bits, "16QAM")
        X = o.OFDM_map_carriers(iq, ns, nfft, dc, pc, pv)
        tx = o.OFDM_modulator(X, tg)
        rx = o.apply_channel(o.add_CFO(o.add_STO(tx.ravel(order="F"), 21), 3.3, nfft), h)
        nr, ni = o.awgn_philox(rx.size, 77, sub)
        rx, nvar = o.Noise(30.0, rx, nr, ni)
        rho, pos, fo, ok = o.AutoCorrFunction(rx, tg, nfft)
        rx1 = o.add_CFO(o.add_STO(o.add_STO(rx, pos), -(nfft + tg)), -fo, nfft)
        rx2, ifo = o.remove_IFO(rx1, nfft)
        R = o.OFDM_demodulator(rx2.reshape((nfft + tg, ns), order="F"), tg)
        Rs, tau, ph = o.fine_sync(R, pc, pv, 1, 1)
        Hest, Hp = o.estimate_channel(Rs, np.arange(1, nfft + 1.0), pc, pv)
        eq = o.equalize_signal(Rs, Hest, nc)
        out_bits = o.demapping(-1, o.get_payload(eq, dc).ravel(order="F"), "16QAM")
        per = len(dc) * bps
        if ok and ifo == 3 and np.isfinite(tau) and not np.any(out_bits[per:] != pay_bits[per:]):
            break
    else:
        raise SystemExit("no sub-seed gave a decodable T4 chain")
    print("t4 chain: sub-seed", sub, "pos", pos, "fo", fo, "ifo", ifo, "tau*N", tau * nfft, "BER", np.mean(out_bits != pay_bits))
    np.savez_compressed(os.path.join(HERE, "t4_chain.npz"), pc=pc, dc=dc, pv=pv, pay_bits=pay_bits, X=X, tx=tx,
                        h=h, H=H, rx=rx, nvar=nvar, rho=rho, pos=pos, fo=fo, ifo=ifo, R=R, Rs=Rs, tau=tau, ph=ph,
                        Hest=Hest, Hp=Hp, eq=eq, out_bits=out_bits, nfft=nfft, nc=nc, tg=tg, ns=ns, noise_stream=sub)

    # ---- estimators (Nfft 1024, N_carrier 256, comb 4)
    nfft, nc, comb, ns = 1024, 256, 4, 3
    pc, dc = o.pilot_layout_comb(nc, comb)
    X = np.zeros((nfft, ns), complex)
    pvv = np.full((len(pc), ns), 1.6 + 0j)
    X[(pc - 1).astype(int)] = pvv
    X[(dc - 1).astype(int)] = crandn(rng, len(dc), ns)
    h, H = o.get_MP_channel_resp(TAPS6, nfft)
    rx = o.apply_channel(o.OFDM_modulator(X, nfft // 8).ravel(order="F"), h)
    rx, _ = o.Noise(20.0, rx, rng=rng)
    R = o.OFDM_demodulator(rx.reshape((nfft + nfft // 8, ns), order="F"), nfft // 8)
    Hls = o.LS_CE(R, pvv, pc, nc)
    Hmmse, tau_rms = o.MMSE_CE(R, pvv, pc, nfft, nc, np.fft.ifft(Hls), 20.0)
    K = int(np.ceil(nc / comb))
    S = o.sensing_matrix(pc, nfft, K)
    Y = R[(pc - 1).astype(int), 0] / pvv[:, 0]
    Hmp, hmp, kp = o.MP_estimate(Y, S, nfft, 6)
    Homp, homp, idx = o.OMP_estimate(Y, S, nfft, 6)
    lin = o.interpolate(Y[:10], pc[:10], 40, "linear")
    spl = o.interpolate(Y[:10], pc[:10], 40, "spline")
    np.savez_compressed(os.path.join(HERE, "estimators.npz"), pc=pc, dc=dc, pv=pvv, R=R, H=H, Hls=Hls, Hmmse=Hmmse,
                        tau_rms=tau_rms, S=S, Y=Y, Hmp=Hmp, hmp=hmp, kp=kp, Homp=Homp, homp=homp, idx=idx,
                        lin=lin, spl=spl, nfft=nfft, nc=nc)

    # ---- fused Task-5 chain (Nfft 512, 16QAM, 3 frames of 4 symbols)
    nfft, nc, comb, ns, F = 512, 128, 4, 4, 3
    tg = nfft // 8
    pc, dc = o.pilot_layout_comb(nc, comb)
    D, bps = o.constellation_func("16QAM")
    amp = 2 * np.max(np.abs(D))
    pvc = np.where(np.arange(len(pc)) % 2 == 0, amp, -amp).astype(complex)
    h, _ = o.get_MP_channel_resp(np.array([[0, 1], [3, .6], [7, .3]]), nfft)
    cbits = rng.integers(0, 2, (F, len(dc) * ns * bps)).astype(np.uint8)
    rxf = np.zeros(((nfft + tg) * ns, F), complex)
    for f in range(F):
        iq, _ = o.mapping(cbits[f], "16QAM")
        Xf = o.OFDM_map_carriers(iq, ns, nfft, dc, pc, np.repeat(pvc[:, None], ns, axis=1))
        y = o.apply_channel(o.OFDM_modulator(Xf, tg).ravel(order="F"), h)
        rxf[:, f], _ = o.Noise(22.0, y, rng=rng)
    ref = o.rx_chain_task5(rxf, nfft, tg, nc, pc, dc, pvc, nc // comb, 3, "16QAM", ref_bits=cbits)
    np.savez_compressed(os.path.join(HERE, "task5_chain.npz"), pc=pc, dc=dc, pilots=pvc, rx=rxf, bits_tx=cbits,
                        bits_rx=ref["bits"], errors=ref["errors"], H=ref["H"],
                        index=np.array([list(i) + [0] * (3 - len(i)) for i in ref["index"]]),
                        nfft=nfft, nc=nc, tg=tg, ns=ns, taps=3)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


def papr_fixture():
    """Task 2 PAPR study: an OFDM stream (Nfft 256, 16QAM, 12 symbols) -> whole-signal PAPR, window PAPR, CCDF."""
    rng = np.random.default_rng(20240502)
    nfft, nc, tg, ns = 256, 100, 32, 12
    pc, dc = o.pilot_layout_percent(nfft, nc, 15, 2)
    D, bps = o.constellation_func("16QAM")
    iq, _ = o.mapping(rng.integers(0, 2, len(dc) * ns * bps), "16QAM")
    tx = o.OFDM_modulator(o.OFDM_map_carriers(iq, ns, nfft, dc, pc, 2 * np.max(np.abs(D))), tg).ravel(order="F")
    paprs = o.calculate_window_PAPR(tx, nfft)
    x, c = o.calculateCCDF(np.round(paprs, 3))                     # rounded: ties, like a histogram of the curve
    np.savez_compressed(os.path.join(HERE, "papr.npz"), tx=tx, nfft=nfft, papr=o.calculatePAPR(tx), paprs=paprs,
                        ccdf_in=np.round(paprs, 3), ccdf_x=x, ccdf=c)


if __name__ == "__main__":
    if "--papr-only" in sys.argv:
        papr_fixture()
    else:
        main()
        papr_fixture()
