"""Driver replays on the HIP library against the same replay on the CPU oracle:
identical seeds, identical call order, host arrays in float64 -> fp64 kernels.  Bits and integer outputs
must be identical; tables (MER, MSE, NMSE, frequency offsets) agree to the fp64 tolerances of DESIGN.md 4."""
import numpy as np
import pytest

from oracle_lib import OracleLib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def olib(oracle):
    return OracleLib(oracle)


@pytest.fixture(scope="module")
def drivers(ofdm):
    from ofdm_course_amd import drivers as d
    return d


def test_task1_reference_size(drivers, ofdm, olib):
    g = drivers.task1.run(ofdm)                                        # Nfft 1024, 16QAM, 50 symbols (as committed)
    o = drivers.task1.run(olib)
    assert g["passed"] and o["passed"] and g["BER"] == 0.0
    assert np.array_equal(g["_output_bits"], o["_output_bits"])
    np.testing.assert_allclose(g["_RX_IQ"], o["_RX_IQ"], atol=1e-12)


def test_task1_noisy(drivers, ofdm, olib):
    g = drivers.task1.run(ofdm, SNR_dB=2, Constellation="QPSK")
    o = drivers.task1.run(olib, SNR_dB=2, Constellation="QPSK")
    assert np.array_equal(g["_output_bits"], o["_output_bits"]) and g["BER"] == o["BER"] > 0


def test_task2(drivers, ofdm, olib):
    g = drivers.task2.run(ofdm)
    o = drivers.task2.run(olib)
    assert g["passed"] and g["passed_scrambled"]
    assert np.array_equal(g["_sc_bits"], o["_sc_bits"]) and np.array_equal(g["_dsc_bits"], o["_dsc_bits"])


def test_task3_run_and_sweep(drivers, ofdm, olib):
    kw = dict(noise_desync=1, SNRs=np.arange(0, 31, 3.0))
    g = drivers.task3.run(ofdm, **kw)
    o = drivers.task3.run(olib, **kw)
    assert np.array_equal(g["_dsc_bits"], o["_dsc_bits"]) and g["BER"] == o["BER"]
    assert abs(g["MER_dB"] - o["MER_dB"]) < 1e-8
    assert np.array_equal(g["sweep"]["BERs"], o["sweep"]["BERs"])
    assert g["sweep"]["BERs"][3, 0] > 0.05 and g["sweep"]["BERs"][0, -1] == 0.0


def test_task3_sto_cfo_flags(drivers, ofdm, olib):
    kw = dict(time_desync=1, freq_desync=1, mp_desync=0, SNRs=[20.0], Constellations=("QPSK",))
    g = drivers.task3.run(ofdm, **kw)
    o = drivers.task3.run(olib, **kw)
    assert g["BER"] == o["BER"] and abs(g["MER_dB"] - o["MER_dB"]) < 1e-7


@pytest.mark.parametrize("sub", [0, 3, 4])
def test_task4_full_sync(drivers, ofdm, olib, sub):
    kw = dict(noise_desync=1, time_desync=1, freq_desync=1, mp_desync=1, SNR_dB=28, seed=10 + sub)
    g = drivers.task4.run(ofdm, **kw)
    o = drivers.task4.run(olib, **kw)
    assert g["Time_Delay"] == o["Time_Delay"] and g["Freq_Shift"] == o["Freq_Shift"]
    assert g["TgPosition"] == o["TgPosition"] and g["acf_fallback"] == o["acf_fallback"]
    assert abs(g["FreqOffset"] - o["FreqOffset"]) < 1e-9
    if np.isfinite(o["e_IFO"]):
        assert g["e_IFO"] == o["e_IFO"]
        assert abs(g["BER"] - o["BER"]) <= 3 / g["_input_bits"].size
        if np.isfinite(o["MER_dB"]):
            assert abs(g["MER_dB"] - o["MER_dB"]) < 1e-6 * max(1.0, abs(o["MER_dB"]))
    else:
        assert not np.isfinite(g["e_IFO"])


def test_task4_committed_flags(drivers, ofdm, olib):
    g = drivers.task4.run(ofdm)                                        # all impairments off (:81-87)
    assert g["passed"] and g["BER"] == 0.0


def test_task5_estimator_tables(drivers, ofdm, olib):
    kw = dict(SNRs=[0.0, 10.0, 20.0, 30.0])                            # Nfft 4096, N_carrier 1024, comb 1 (as committed)
    g = drivers.task5.run(ofdm, **kw)
    o = drivers.task5.run(olib, **kw)
    assert np.array_equal(g["OMP_index"], o["OMP_index"])
    for k in ("LS", "MMSE", "MP", "OMP"):
        assert g["MSE"][k] == pytest.approx(o["MSE"][k], rel=1e-6), k
    np.testing.assert_allclose(g["sweep"]["MSEs"], o["sweep"]["MSEs"], rtol=1e-6)
    np.testing.assert_allclose(g["_H_est"], o["_H_est"], rtol=1e-10, atol=1e-10)   # spline extrapolates to ~1e8 beyond N_carrier


@pytest.mark.parametrize("kw", [dict(),                                                  # as committed: Nfft 4096, comb 1, Np = K = 1024
                                dict(Nfft=2048, N_carrier=512, comb=4, Constellation="64QAM"),
                                dict(Nfft=1024, N_carrier=256, comb=8, Constellation="QPSK")])   # (the single-frame part needs a power-of-two N_carrier: its ifft is an OFDM_modulator call)
def test_task5_sweep_tile_equals_oracle_and_call_by_call(drivers, ofdm, olib, kw):
    """ofdm_task5_mse_tile: the MSE(SNR) sweep of Main_model_Task_5.m:303-346 as ONE device-resident
    call -- Noise per point, conv, demodulator, LS / MMSE(h = ifft(H_LS)) / MP / OMP, four errors -- against the OracleLib replay
    and the point-by-point form on the per-function entries (fp64: 1e-9), sharded over two ranks, and in fp32."""
    snrs = np.array([0.0, 3.5, 10.0, 17.0, 20.0, 30.0])
    t = drivers.task5.run(ofdm, SNRs=snrs, batched=True, **kw)["sweep"]["MSEs"]
    p = drivers.task5.run(ofdm, SNRs=snrs, batched=False, **kw)["sweep"]["MSEs"]
    o = drivers.task5.run(olib, SNRs=snrs, **kw)["sweep"]["MSEs"]
    np.testing.assert_allclose(t, o, rtol=1e-9)
    np.testing.assert_allclose(t, p, rtol=1e-9)
    parts = [drivers.task5.run(ofdm, SNRs=snrs, batched=True, rank=r, world=2, **kw)["sweep"]["MSEs"] for r in range(2)]
    assert np.array_equal(parts[0] + parts[1], t) and np.all(parts[0][:, 3:] == 0) and np.all(parts[1][:, :3] == 0)
    f32 = drivers.task5.run(ofdm, SNRs=snrs, batched=True, precision="fp32", **kw)["sweep"]["MSEs"]
    np.testing.assert_allclose(f32, o, rtol=2e-3)


def test_task5_comb4_payload(drivers, ofdm, olib):
    kw = dict(Nfft=2048, N_carrier=512, comb=4, Constellation="64QAM", SNR_dB=26, SNRs=[20.0])
    g = drivers.task5.run(ofdm, **kw)
    o = drivers.task5.run(olib, **kw)
    assert np.array_equal(g["_dsc_bits"], o["_dsc_bits"]) and g["BER"] == o["BER"]
    assert abs(g["MER_dB"] - o["MER_dB"]) < 1e-7
    np.testing.assert_allclose(g["sweep"]["MSEs"], o["sweep"]["MSEs"], rtol=1e-6)


def test_task5_part2_subset(drivers, ofdm, olib):
    kw = dict(combs=[4, 16, 64], monteCarloRuns=3)                     # Nfft 4096 / N_carrier 1024 / EPA @ 40 MHz
    g = drivers.task5_part2.run(ofdm, **kw)
    o = drivers.task5_part2.run(olib, **kw)
    assert np.array_equal(g["_sums"]["errors"], o["_sums"]["errors"])
    np.testing.assert_allclose(g["NMSEs"], o["NMSEs"], rtol=1e-6)
    assert np.array_equal(g["amounts_pilots"], [256, 64, 16])


def test_task5_part2_random_pilots(drivers, ofdm, olib):
    """reg_pilot = 0 (T5/Task5_part2.m:57-64): random pilot masks, dictionary of all Nfft delays (:181-184)."""
    kw = dict(Nfft=1024, N_carrier=256, Nps=[24, 64], reg_pilot=0, monteCarloRuns=2)
    g = drivers.task5_part2.run(ofdm, **kw)
    o = drivers.task5_part2.run(olib, **kw)
    assert np.array_equal(g["_sums"]["errors"], o["_sums"]["errors"])
    np.testing.assert_allclose(g["NMSEs"], o["NMSEs"], rtol=1e-6)
    assert np.array_equal(g["amounts_pilots"], [24, 64]) and g["reg_pilot"] == 0


def test_task2_papr_tables(drivers, ofdm, olib):
    """T2/Main_model_Task_2.m:69-82 through the driver replay."""
    kw = dict(Nfft=512, N_carrier=200, Amount_OFDM_Frames=3, Amount_ODFM_SpF=4)
    g = drivers.task2.run(ofdm, **kw)
    o = drivers.task2.run(olib, **kw)
    for tag in ("plain", "scrambled"):
        assert abs(g["papr"][tag]["PAPR_dB"] - o["papr"][tag]["PAPR_dB"]) < 1e-9
        np.testing.assert_allclose(g["papr"][tag]["_PAPRs"], o["papr"][tag]["_PAPRs"], rtol=0, atol=1e-9)
        assert g["papr"][tag]["CCDF"][0] == 1.0 and g["papr"][tag]["CCDF"][-1] == 0.0
        assert np.all(np.diff(g["papr"][tag]["PAPR_ccdf"][1:]) > 0)
