"""CPU: pins the oracle (and through it every GPU parity test) to the numbers the reference itself publishes --
BASELINE.md section 2 / SURVEY.md section 6: the BER(SNR) graph of Task 3, the MSE(SNR) graph of Task 5, the
NMSE(SNR) graph and the interpolation-type MER table of Task 4, the PAPR numbers of Task 2 (on the reference's own
image payload).  The replays are `ofdm_course_amd.drivers.taskN.run(lib=OracleLib)`; the GPU twin
(tests/test_gpu_published.py) runs the same replays on the HIP library.

Two published results are NOT what the committed .m files compute; both are reproduced by a named one-line
variant, asserted here next to the committed behaviour (analysis: DESIGN.md section 0):
  * the MMSE curve of `Task 5/graphs/mse(snr), comb1.png` needs `df = 1/Nfft` on MMSE_CE.m:25 (the form in that
    line's comment) instead of the committed `df = 1/N_carrier`;
  * the MER table of Task 4/README.md:181-183 (60 / 108 / 130 dB) needs pilot period 2 on a uniform grid (the
    README's own figure 23 is "pilot period = 2") and interp1's 'cubic' = cubic convolution (MATLAB >= R2020b),
    not the committed Percent_pilot = 15 (period 6).
"""
import numpy as np
import pytest

import published as pub
from oracle_lib import OracleLib


@pytest.fixture(scope="module")
def olib(oracle):
    return OracleLib(oracle)


@pytest.fixture(scope="module")
def drivers():
    from ofdm_course_amd import drivers as d
    return d


# ---------------------------------------------------------------------------------------- Task 3: BER(SNR)
def test_task3_ber_snr_points(drivers, olib):
    """Task 3/README.md:57-60, graphs/ber(snr).png <- Main_model_Task_3.m:192-268 (AWGN only, 4 constellations)."""
    snrs = sorted({s for pts in pub.BER_POINTS.values() for s in pts})
    r = drivers.task3.run(olib, SNRs=snrs)
    rep = pub.check_ber_sweep(r["sweep"], n_bits_per_bps=50 * 332)
    assert len(rep) == 15


# ---------------------------------------------------------------------------------------- Task 5: MSE(SNR)
def test_task5_mse_snr_ls_mp_omp(drivers, olib):
    """Task 5/README.md:32-39, graphs/mse(snr), comb1.png <- Main_model_Task_5.m:303-346."""
    snrs = sorted({s for pts in pub.MSE_POINTS.values() for s in pts})
    r = drivers.task5.run(olib, SNRs=snrs)
    pub.check_mse_sweep(r["sweep"])


def _task5_frames(oracle, olib, snr, seed, N_symb=14):
    """Main_model_Task_5.m:307-312 on the committed set-up: pilots-only TX, Noise, 6-tap channel, demodulator."""
    from ofdm_course_amd.drivers import task5
    Nfft, Nc, Tg = 4096, 1024, 512
    D, _ = oracle.constellation_func("16QAM")
    amp = 4 / 3 * np.max(np.abs(D))
    pil = np.arange(1, Nc + 1, dtype=np.float64)
    X = np.zeros((Nfft, N_symb), complex)
    X[:Nc, :] = amp
    tx = oracle.OFDM_modulator(X, Tg).ravel(order="F")
    h, H = oracle.get_MP_channel_resp(task5.CHANNEL_TAPS, Nfft)
    rx, _ = olib.Noise(snr, tx, seed=seed, stream=7)
    rx = oracle.apply_channel(rx, h)
    Xr = oracle.OFDM_demodulator(rx.reshape((Nfft + Tg, N_symb), order="F"), Tg)
    return Xr, np.full((Nc, N_symb), amp, complex), pil, H[:Nc], Nfft, Nc


def _mmse_mse(oracle, olib, snr, seeds, df=None):
    out = []
    for sd in seeds:
        Xr, pv, pil, H, Nfft, Nc = _task5_frames(oracle, olib, snr, sd)
        H_ls = oracle.LS_CE(Xr, pv, pil, Nc)                                  # :313
        h_t = np.fft.ifft(H_ls)                                               # :314
        H_mmse, _ = oracle.MMSE_CE(Xr, pv, pil, Nfft, Nc, h_t, snr, df=df)    # :315
        out.append((np.mean(np.abs(H - H_ls) ** 2), np.mean(np.abs(H - H_mmse) ** 2)))
    return np.mean(out, axis=0)


def test_mmse_committed_file_does_not_give_the_published_curve(oracle, olib):
    """MMSE_CE.m as committed (`df = 1/N_carrier`, :25): 0.42-0.46 at 0 dB and 0.045-0.051 at 10 dB -- a factor
    2.4-2.8 above the published 0.175 / 0.017.  Recorded so that nobody 'fixes' the kernel towards the graph."""
    for snr, want in ((0.0, 0.175), (10.0, 0.017)):
        ls, mm = _mmse_mse(oracle, olib, snr, seeds=(1, 2, 3))
        assert 2.2 < mm / want < 3.2, (snr, mm, want)
        assert 0.60 < mm / ls < 0.75, (snr, mm, ls)               # committed MMSE only shaves a third off LS


@pytest.mark.xfail(strict=True, reason="published MMSE curve comes from df = 1/Nfft, not the committed df = 1/N_carrier")
def test_mmse_published_curve_kat_committed_file(oracle, olib):
    for snr, want in pub.MSE_MMSE_PUBLISHED.items():
        _, mm = _mmse_mse(oracle, olib, snr, seeds=(1,))
        assert abs(mm - want) <= 0.25 * want, (snr, mm, want)


def test_mmse_published_curve_is_the_df_1_over_nfft_variant(oracle, olib):
    """With `df = 1/Nfft` (the textbook form in the comment of MMSE_CE.m:25) the same replay lands on the published
    curve at every point read off the graph: MMSE ~ LS/4 (0.172 / 0.107 / 0.051 / 0.015 here, 0.175 / 0.095 /
    0.044 / 0.017 published; one realisation per point in the graph, 3 averaged here)."""
    for snr, want in pub.MSE_MMSE_PUBLISHED.items():
        ls, mm = _mmse_mse(oracle, olib, snr, seeds=(1, 2, 3), df=1.0 / 4096)
        assert abs(mm - want) <= 0.20 * want, (snr, mm, want)
        assert 0.19 < mm / ls < 0.28, (snr, mm, ls)


# ---------------------------------------------------------------------------------------- Task 4: channel estimate
def _task4_link(oracle, olib, pilotCarriers, N_symb, snr=None, seed=1, last_data=None):
    """Main_model_Task_4.m:39-68 + :257-264 + :308-310 with only the multipath (and optionally AWGN) switched on."""
    from ofdm_course_amd.drivers import common as c, task4
    Nfft, Nc, Tg = 1024, 400, 128
    allc = np.arange(1, Nfft + 1, dtype=np.float64)
    data = allc[:Nc][~np.isin(allc[:Nc], pilotCarriers)]
    if last_data is not None:
        data = data[data <= last_data]
    D, bps = oracle.constellation_func("16QAM")
    pv = c.alternating_pilots(4 / 3 * np.max(np.abs(D)), len(pilotCarriers), N_symb)
    bits = c.synthetic_bits(N_symb * len(data) * bps, seed)
    iq, _ = oracle.mapping(bits, "16QAM")
    X = oracle.OFDM_map_carriers(iq, N_symb, Nfft, data, pilotCarriers, pv)
    tx = oracle.OFDM_modulator(X, Tg).ravel(order="F")
    h, H = oracle.get_MP_channel_resp(task4.CHANNEL_TAPS, Nfft)
    rx = tx
    if snr is not None:
        rx, _ = olib.Noise(snr, rx, seed=seed, stream=3)
    rx = oracle.apply_channel(rx, h)
    Xr = oracle.OFDM_demodulator(rx.reshape((Nfft + Tg, N_symb), order="F"), Tg)
    return Xr, allc, data, pv, H, Nc


def test_task4_nmse_snr_of_estimate_channel(drivers, oracle, olib):
    """Task 4/graphs/nmse(snr).png (README.md:185-191, pilot_step = 4) <- the commented sweep of
    Main_model_Task_4.m:205-239: estimate_channel's MSE over 1..N_carrier.  Published (read off): 5.5e-3 / 1.75e-3 /
    4e-4 / 1.3e-4 at 0 / 5 / 10 / 15 dB, one realisation per point."""
    _, pil, _ = drivers.common.layout_percent(1024, 400, 25, tail=2)
    assert pil[1] - pil[0] == 4
    for snr, want in ((0.0, 5.5e-3), (5.0, 1.75e-3), (10.0, 4.0e-4), (15.0, 1.3e-4)):
        v = []
        for sd in (1, 2, 3):
            Xr, allc, _, pv, H, Nc = _task4_link(oracle, olib, pil, 50, snr=snr, seed=sd)
            He, _ = oracle.estimate_channel(Xr, allc, pil, pv)
            v.append(np.mean(np.abs(H[:Nc] - He[:Nc]) ** 2))
        assert abs(np.mean(v) - want) <= 0.30 * want, (snr, np.mean(v), want)


def _mer_by_method(oracle, olib, pil, last_data=None):
    Xr, allc, data, pv, _, Nc = _task4_link(oracle, olib, pil, 10, last_data=last_data)
    out = {}
    for m in ("linear", "cubic", "spline"):
        try:
            He, _ = oracle.estimate_channel(Xr, allc[:Nc], pil, pv, method=m)
        except ValueError:
            out[m] = None                                       # 'cubic' on a non-uniform grid
            continue
        He = np.concatenate([He, np.ones(1024 - Nc)])
        eq = oracle.equalize_signal(Xr, He, Nc)
        out[m] = float(oracle.MER_func(oracle.get_payload(eq, data).ravel(order="F"), "16QAM"))
    return out


def test_task4_mer_table_committed_parameters(drivers, oracle, olib):
    """Percent_pilot = 15 (period 6, 68 pilots, last pilot appended at 400: Main_model_Task_4.m:14-21), taps 0/4/10:
    linear 42.7 dB, spline 94.2 dB -- not the README's 60 / 130; 'cubic' cannot be evaluated at all (knots not
    uniform: 397 -> 400)."""
    _, pil, _ = drivers.common.layout_percent(1024, 400, 15, tail=2)
    got = _mer_by_method(oracle, olib, pil)
    assert abs(got["linear"] - 42.7) < 0.5 and abs(got["spline"] - 94.2) < 0.7 and got["cubic"] is None, got


@pytest.mark.xfail(strict=True, reason="README table was made with pilot period 2, not the committed Percent_pilot = 15")
def test_task4_mer_table_kat_committed_parameters(drivers, oracle, olib):
    _, pil, _ = drivers.common.layout_percent(1024, 400, 15, tail=2)
    got = _mer_by_method(oracle, olib, pil)
    assert abs(got["linear"] - pub.MER_TABLE_PUBLISHED["linear"]) < 3 and abs(got["spline"] - pub.MER_TABLE_PUBLISHED["spline"]) < 3


def test_task4_mer_table_is_pilot_period_2(oracle, olib):
    """Pilots 1:2:399 (uniform; data carriers between them), same 3-tap channel, noiseless: linear 59.8, cubic
    convolution 107.1, spline 129.2 dB -- the README's 60 / 108 / 130 to within 1 dB.  (With the committed rule and
    Percent_pilot = 50 the appended pilot 400 makes the grid non-uniform: linear 59.8, spline 122.6, no 'cubic'.)"""
    pil = np.arange(1, 400, 2, dtype=np.float64)
    got = _mer_by_method(oracle, olib, pil, last_data=399)
    for m, want in pub.MER_TABLE_PUBLISHED.items():
        assert abs(got[m] - want) < 1.5, (m, got[m], want)


def test_v5cubic_is_exact_on_quadratics_and_nan_outside(oracle):
    x = np.arange(1.0, 20.0, 3.0)
    f = lambda t: 0.3 * t * t - 2 * t + 1
    q = np.linspace(1, 19, 55)
    assert np.allclose(oracle.interp1_v5cubic(x, f(x), q), f(q), atol=1e-12)
    assert np.isnan(oracle.interp1_v5cubic(x, f(x), [0.5, 19.5])).all()
    with pytest.raises(ValueError):
        oracle.interp1_v5cubic([1, 2, 4, 5], [0, 1, 2, 3], [1.5])


# ---------------------------------------------------------------------------------------- Task 2: PAPR on the image
def test_task2_papr_on_the_reference_payload(drivers, olib):
    """Task 2/README.md:54, :70-71: whole-signal PAPR 22-23 dB plain vs ~10 dB scrambled; sliding-window PAPR exceeded
    with probability 0.02: ~22 dB vs ~10 dB.  Payload = file_reader('eagle.tiff') (fixture).  Here: 22.3 / 11.8 dB and
    21.7 / 10.6 dB."""
    bits = pub.eagle_bits()
    assert bits.size == 129600 and abs(bits.mean() - 0.3370) < 1e-4
    assert "".join(map(str, bits[:64])) == "1110001111111111111101111000011110000000111000000000000000000000"
    r = drivers.task2.run(olib, input_bits=bits)
    assert r["passed"] and r["passed_scrambled"]
    pub.check_papr(r["papr"])
