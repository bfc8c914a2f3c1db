"""CPU: the oracle (numpy restatement of the .m reference) against analytic known-answer tests
derived from the reference code.  The reference ships no tests / vectors,
so these KATs plus the committed golden fixtures are what pins the oracle ("parity unpinned")."""
import warnings

import numpy as np
import pytest

from oracle import ofdm_oracle as o

TAPS6 = np.array([[0, 1], [4, .8], [10, .6], [15, .4], [21, .2], [25, .1]])
CONSTS = ["BPSK", "QPSK", "8PSK", "16QAM", "64QAM", "256QAM"]


@pytest.mark.parametrize("name", CONSTS)
def test_constellation_unit_power_and_loopback(name):
    D, bps = o.constellation_func(name)
    assert len(D) == 2 ** bps and abs(np.mean(np.abs(D) ** 2) - 1) < 1e-14       # KAT (8)
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2, 999 * bps)
    iq, pad = o.mapping(bits, name)
    assert pad == -1 and np.array_equal(o.demapping(pad, iq, name), bits)         # KAT (1)


def test_16qam_table_is_the_reference_literal():
    D, _ = o.constellation_func("16QAM")
    lit = np.array([-3 + 3j, -3 + 1j, -3 - 3j, -3 - 1j, -1 + 3j, -1 + 1j, -1 - 3j, -1 - 1j,
                    3 + 3j, 3 + 1j, 3 - 3j, 3 - 1j, 1 + 3j, 1 + 1j, 1 - 3j, 1 - 1j]) / np.sqrt(10)
    assert np.max(np.abs(D - lit)) < 1e-15
    # the 64/256-QAM extension reduces to the literal rule on its 2-bit sub-grid
    assert np.allclose(o._square_qam(2) / np.sqrt(10), lit)


def test_mapping_padding_rules():
    iq, pad = o.mapping(np.ones((7, 1)), "16QAM")
    assert pad == 1 and len(iq) == 2
    with pytest.raises(ValueError):
        o.mapping(np.ones((1, 7)), "16QAM")                                       # mapping.m:11 vertcat error
    assert len(o.demapping(1, iq, "16QAM")) == 7


def test_scrambler_kat_and_inverse():
    sc, reg = o.Scrambler(o.DEFAULT_REGISTER, np.zeros(48))
    assert "".join(map(str, sc)) == "000001111110110000100000110100011000010111001010"   # KAT (2)
    rng = np.random.default_rng(1)
    x = rng.integers(0, 2, 5000)
    s, r = o.Scrambler(o.DEFAULT_REGISTER, x)
    s2, r2 = o.Scrambler_fast(o.DEFAULT_REGISTER, x)
    assert np.array_equal(s, s2) and np.array_equal(r, r2)
    d, _ = o.DeScrambler(o.DEFAULT_REGISTER, s)
    d2, _ = o.DeScrambler_fast(o.DEFAULT_REGISTER, s)
    assert np.array_equal(d, x) and np.array_equal(d2, x)
    s_err = s.copy(); s_err[100] ^= 1
    assert np.count_nonzero(o.DeScrambler(o.DEFAULT_REGISTER, s_err)[0] != x) == 3   # self-synchronising


def test_mod_demod_roundtrip_and_cp():
    rng = np.random.default_rng(2)
    X = rng.standard_normal((256, 5)) + 1j * rng.standard_normal((256, 5))
    y = o.OFDM_modulator(X, 32)
    assert y.shape == (288, 5) and np.array_equal(y[:32], y[256:])
    assert np.max(np.abs(o.OFDM_demodulator(y, 32) - X)) < 1e-12                  # KAT (3)


def test_pilot_layouts_of_the_drivers():
    pc, dc = o.pilot_layout_percent(1024, 400, 25, 1)
    assert (len(pc), len(dc)) == (101, 299)                                       # T1/Main_model.m:14-24
    pc, dc = o.pilot_layout_percent(1024, 400, 15, 2)
    assert (len(pc), len(dc)) == (68, 332)                                        # T4/Main_model_Task_4.m:14-24
    pc, dc = o.pilot_layout_comb(512, 4)
    assert (len(pc), len(dc)) == (128, 384) and pc[1] - pc[0] == 4


def test_multipath_is_a_per_carrier_gain():
    """KAT (4): max delay < Tg -> demod of every symbol equals X .* H_freq."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((512, 4)) + 1j * rng.standard_normal((512, 4))
    h, H = o.get_MP_channel_resp(TAPS6, 512)
    rx = o.apply_channel(o.OFDM_modulator(X, 64).ravel(order="F"), h)
    R = o.OFDM_demodulator(rx.reshape((576, 4), order="F"), 64)
    assert np.max(np.abs(R - X * H[:, None])) < 1e-12


def test_mp_omp_published_floors():
    """KAT (5): noiseless Main_model_Task_5 set-up -> the high-SNR floors of mse(snr), comb1.png."""
    nfft, nc = 4096, 1024
    _, H = o.get_MP_channel_resp(TAPS6, nfft)
    S = o.sensing_matrix(np.arange(1, nc + 1.0), nfft, nc)
    Hmp, _, kp = o.MP_estimate(H[:nc], S, nfft, 6)
    Homp, _, idx = o.OMP_estimate(H[:nc], S, nfft, 6)
    assert list(kp) == [1, 5, 11, 16, 21, 6] and list(idx) == [1, 5, 11, 16, 21, 27]
    mse = lambda A: np.mean(np.abs(H[:nc] - A[:nc]) ** 2)
    assert abs(mse(Hmp) - 0.02373) < 1e-5 and abs(mse(Homp) - 0.002916) < 1e-6


def test_ls_mmse_kats():
    nfft, nc = 4096, 1024
    _, H = o.get_MP_channel_resp(TAPS6, nfft)
    pc, _ = o.pilot_layout_comb(nc, 4)
    Y, Xp = H.reshape(-1, 1), np.ones((256, 1))
    Hls = o.LS_CE(Y, Xp, pc, nc)
    assert abs(np.mean(np.abs(Hls - H[:nc]) ** 2) - 3.94e-9) < 2e-11
    Hm, tau = o.MMSE_CE(Y, Xp, pc, nfft, nc, np.fft.ifft(Hls), 20)
    assert abs(tau - 107.65) < 0.01 and abs(np.mean(np.abs(Hm - H[:nc]) ** 2) - 2.27e-4) < 1e-6


def test_spline_matches_scipy_not_a_knot():
    from scipy.interpolate import CubicSpline
    rng = np.random.default_rng(4)
    x = np.array([1, 5, 9, 14, 20, 33.0])
    y = rng.standard_normal(6) + 1j * rng.standard_normal(6)
    xq = np.arange(-3, 40.0)
    for n in (2, 3, 4, 6):
        assert np.max(np.abs(o.interp1_spline(x[:n], y[:n], xq) - CubicSpline(x[:n], y[:n], bc_type="not-a-knot")(xq))) < 1e-12


def test_coarse_sync_kat():
    """KAT (6)/(7): STO 37 + CFO 7.24 noiseless: plateau at the CP start, FFO and IFO recovered."""
    rng = np.random.default_rng(1)
    nfft, tg, ns = 1024, 128, 12
    pc, dc = o.pilot_layout_percent(nfft, 400, 15, 2)
    D, bps = o.constellation_func("16QAM")
    amp = 4 / 3 * np.max(np.abs(D))
    pv = np.repeat(np.where(np.arange(len(pc)) % 2 == 0, amp, -amp).astype(complex)[:, None], ns, axis=1)
    iq, _ = o.mapping(rng.integers(0, 2, len(dc) * ns * bps), "16QAM")
    tx = o.OFDM_modulator(o.OFDM_map_carriers(iq, ns, nfft, dc, pc, pv), tg).ravel(order="F")
    rx = o.add_CFO(o.add_STO(tx, 37), 7.24, nfft)
    rho, pos, fo, ok = o.AutoCorrFunction(rx, tg, nfft)
    assert ok and 1116 - 8 <= pos <= 1116 and abs(fo - 0.24) < 1e-3
    rx2 = o.add_CFO(o.add_STO(o.add_STO(rx, pos), -(nfft + tg)), -fo, nfft)
    _, ifo = o.remove_IFO(rx2, nfft)
    assert ifo == 7
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert o.AutoCorrFunction(rng.standard_normal(4000) + 0j, tg, nfft)[1] == 65     # catch branch


def test_philox_reference_vector():
    """Philox4x32-10 known answers (Random123 kat_vectors: zero counter/key, and the pi-digits vector)."""
    r = o.philox4x32_10(np.zeros((1, 4), np.uint32), np.zeros(2, np.uint32))[0]
    assert [hex(int(v)) for v in r] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    r = o.philox4x32_10(np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32),
                        np.array([0xa4093822, 0x299f31d0], np.uint32))[0]
    assert [hex(int(v)) for v in r] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_papr_kat():
    """Task 2 PAPR study: closed-form cases.  A constant-envelope signal has 0 dB; one sample of amplitude A among
    N-1 unit samples has 10 log10(A^2 N / (A^2 + N - 1)); the window version is the scalar one window by window
    (`calculate_window_PAPR.m:8-14`); ecdf of [3 1 2 2] is x = [1 1 2 3], F = [0 .25 .75 1] (`calculateCCDF.m:4-5`)."""
    assert o.calculatePAPR(np.exp(1j * np.arange(64))) == pytest.approx(0.0, abs=1e-12)
    x = np.ones(100, complex)
    x[17] = 5j
    assert o.calculatePAPR(x) == pytest.approx(10 * np.log10(25 * 100 / (25 + 99)), abs=1e-12)
    rng = np.random.default_rng(2)
    y = rng.standard_normal(400) + 1j * rng.standard_normal(400)
    w = o.calculate_window_PAPR(y, 64)
    assert w.shape == (337,)
    for i in (0, 1, 100, 336):
        assert w[i] == pytest.approx(o.calculatePAPR(y[i:i + 64]), abs=1e-12)
    assert o.calculate_window_PAPR(y, 401).size == 0
    xs, c = o.calculateCCDF([3.0, 1.0, 2.0, 2.0, np.nan])
    assert np.array_equal(xs, [1, 1, 2, 3]) and np.array_equal(c, [1, 0.75, 0.25, 0])


def _fine_sync_tau_literal(rx_signal, pilotCarriers, pilotValues, t4):
    """T4/fine_sync.m:4-35 / T5/fine_sync.m:4-20 statement by statement with MATLAB's growing-array semantics (the
    check for oracle.fine_sync's vectorised form): returns tau."""
    txv = np.asarray(pilotValues).ravel(order="F")
    rxv = np.asarray(rx_signal)[np.asarray(pilotCarriers, int) - 1, :].ravel(order="F")
    deltak = pilotCarriers[1] - pilotCarriers[0]
    taus = [0.0] * (np.asarray(pilotValues).shape[0] if t4 else txv.size)          # :8
    for i in range(2, txv.size + 1):                                                # :25 / :10
        q_k_1 = txv[i - 2] * np.conj(rxv[i - 2])
        q_k = txv[i - 1] * np.conj(rxv[i - 1])
        v = float(np.angle(q_k * np.conj(q_k_1)) / (2 * np.pi * deltak))
        if i - 1 > len(taus):
            taus.append(v)                                                          # MATLAB grows the row
        else:
            taus[i - 2] = v
    taus = np.array(taus)
    diffs = np.diff(taus)
    mask = np.concatenate([[False], (np.abs(diffs) < 1e-3) & (diffs != 0) if t4 else np.abs(diffs) < 1e-3])
    return float(np.mean(taus[mask][len(pilotCarriers):]))


def test_fine_sync_t4_has_no_trailing_zero():
    """Residual delay of 0.2 samples (tau = 2e-4 < the 1e-3 mask): T5's trailing taus(end) = 0 is averaged into tau,
    T4's grown array has no such entry."""
    rng = np.random.default_rng(11)
    nfft, ns = 256, 3
    pc = np.arange(1, 101, 6)
    pv = np.repeat(np.where(np.arange(len(pc)) % 2 == 0, 1.5, -1.5).astype(complex)[:, None], ns, axis=1)
    X = (rng.standard_normal((nfft, ns)) + 1j * rng.standard_normal((nfft, ns)))
    X[pc - 1, :] = pv
    X *= np.exp(-2j * np.pi * 2e-4 * np.arange(nfft))[:, None]
    X += 1e-6 * (rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape))
    t4 = o.fine_sync(X, pc, pv, 1, 0, variant="T4")[1]
    t5 = o.fine_sync(X, pc, pv, 1, 0, variant="T5")[1]
    assert abs(t4 - _fine_sync_tau_literal(X, pc, pv, True)) < 1e-15
    assert abs(t5 - _fine_sync_tau_literal(X, pc, pv, False)) < 1e-15
    assert abs(t4 - 2e-4) < 2e-7 and 1e-6 < abs(t4 - t5) < 2e-5          # the spurious 0 pulls T5 down
