"""GPU parity: OFDM_modulator / OFDM_demodulator / map_carriers / get_payload / equalize_signal
against the oracle, through the C ABI (numpy = host-pointer flavour, torch = device flavour)."""
import numpy as np
import pytest

from conftest import crandn, rel_l2

pytestmark = pytest.mark.gpu

SIZES = [64, 128, 256, 512, 1024, 2048, 4096, 8192]
# tolerances of SURVEY.md section 8c: rel-L2 <= 1e-13*log2(N) (fp64), 1e-6*log2(N) (fp32)
TOL = {np.complex128: 1e-13, np.complex64: 1e-6}


@pytest.mark.parametrize("nfft", SIZES)
@pytest.mark.parametrize("dt", [np.complex128, np.complex64])
def test_demodulator_matches_oracle(ofdm, oracle, nfft, dt):
    rng = np.random.default_rng(nfft)
    tg = nfft // 8
    ns = 37 if nfft <= 1024 else 5          # ragged vs. transforms-per-workgroup
    y = crandn(rng, nfft + tg, ns).astype(dt)
    got = ofdm.OFDM_demodulator(y, tg)
    want = oracle.OFDM_demodulator(y.astype(np.complex128), tg)
    assert got.dtype == dt and got.shape == (nfft, ns)
    assert rel_l2(got, want) <= TOL[dt] * np.log2(nfft)


@pytest.mark.parametrize("nfft", SIZES)
@pytest.mark.parametrize("dt", [np.complex128, np.complex64])
def test_modulator_matches_oracle_and_cp_is_exact(ofdm, oracle, nfft, dt):
    rng = np.random.default_rng(nfft + 1)
    tg = nfft // 8
    ns = 19 if nfft <= 1024 else 3
    x = crandn(rng, nfft, ns).astype(dt)
    got = ofdm.OFDM_modulator(x, tg)
    want = oracle.OFDM_modulator(x.astype(np.complex128), tg)
    assert got.shape == (nfft + tg, ns)
    assert rel_l2(got, want) <= TOL[dt] * np.log2(nfft)
    # CP indexing is exact: rows 1..Tg are bitwise copies of the last Tg rows (OFDM_modulator.m:8-9)
    assert np.array_equal(got[:tg], got[nfft:])


@pytest.mark.parametrize("nfft", [64, 1024, 2048, 8192])
def test_mod_demod_roundtrip(ofdm, nfft):
    rng = np.random.default_rng(7)
    x = crandn(rng, nfft, 9)
    back = ofdm.OFDM_demodulator(ofdm.OFDM_modulator(x, nfft // 8), nfft // 8)
    assert rel_l2(back, x) <= 1e-12
    x32 = x.astype(np.complex64)
    back32 = ofdm.OFDM_demodulator(ofdm.OFDM_modulator(x32, nfft // 8), nfft // 8)
    assert rel_l2(back32, x32) <= 1e-5


def test_guard_edge_cases(ofdm, oracle):
    rng = np.random.default_rng(3)
    x = crandn(rng, 256, 4)
    for tg in (0, 1, 255, 256):
        y = ofdm.OFDM_modulator(x, tg)
        assert rel_l2(y, oracle.OFDM_modulator(x, tg)) < 1e-13
        assert rel_l2(ofdm.OFDM_demodulator(y, tg), x) < 1e-12
    # empty batch
    assert ofdm.OFDM_demodulator(np.zeros((72, 0), np.complex128), 8).shape == (64, 0)
    with pytest.raises(ofdm.OfdmError):
        ofdm.OFDM_demodulator(crandn(rng, 100 + 10, 2), 10)     # Nfft = 100 unsupported


def test_demodulator_linearity_full_size(ofdm):
    """Size-independent property at BASELINE config-2 scale (Nfft=1024, 20k symbols)."""
    rng = np.random.default_rng(11)
    a = crandn(rng, 1152, 20000).astype(np.complex64)
    b = crandn(rng, 1152, 20000).astype(np.complex64)
    fa, fb = ofdm.OFDM_demodulator(a, 128), ofdm.OFDM_demodulator(b, 128)
    fab = ofdm.OFDM_demodulator((a + 2j * b).astype(np.complex64), 128)
    assert rel_l2(fab, fa + 2j * fb) < 5e-6
    # Parseval per symbol
    e_t = np.sum(np.abs(a[128:].astype(np.complex128)) ** 2, axis=0)
    e_f = np.sum(np.abs(fa.astype(np.complex128)) ** 2, axis=0) / 1024
    assert np.max(np.abs(e_t - e_f) / e_t) < 1e-5


def test_device_pointer_flavour(ofdm, oracle):
    import torch
    rng = np.random.default_rng(5)
    y = crandn(rng, 2048 + 256, 6).astype(np.complex64)
    yt = torch.from_numpy(np.ascontiguousarray(y.T)).cuda().t()      # column-major on the device
    got = ofdm.OFDM_demodulator(yt, 256)
    assert got.is_cuda and tuple(got.shape) == (2048, 6)
    torch.cuda.synchronize()
    assert rel_l2(got.cpu().numpy(), oracle.OFDM_demodulator(y.astype(np.complex128), 256)) < 1.1e-5
    # a non-default torch stream is honoured
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        got2 = ofdm.OFDM_demodulator(yt, 256)
    s.synchronize()
    assert torch.equal(got, got2)


def test_map_carriers_and_payload_are_index_exact(ofdm, oracle):
    rng = np.random.default_rng(9)
    nfft, nc, ns = 1024, 400, 7
    pc, dc = oracle.pilot_layout_percent(nfft, nc, 15, 2)
    assert len(pc) == 68 and len(dc) == 332          # T4/Main_model_Task_4.m:14-24
    pay = crandn(rng, 1, len(dc) * ns)
    pv = crandn(rng, len(pc), ns)
    got = ofdm.OFDM_map_carriers(pay, ns, nfft, dc, pc, pv)
    want = oracle.OFDM_map_carriers(pay, ns, nfft, dc, pc, pv)
    assert np.array_equal(got, want)
    assert np.array_equal(ofdm.get_payload(got, dc), oracle.get_payload(want, dc))
    # scalar pilot broadcast (T3/Main_model_Task_3.m:59) and pilot-wins-on-overlap (:6,:8)
    got_s = ofdm.OFDM_map_carriers(pay, ns, nfft, dc, pc, np.array([1.5 + 0j]))
    assert np.array_equal(got_s, oracle.OFDM_map_carriers(pay, ns, nfft, dc, pc, 1.5 + 0j))
    dc_over = dc.copy(); dc_over[0] = pc[3]
    assert np.array_equal(ofdm.OFDM_map_carriers(pay, ns, nfft, dc_over, pc, pv),
                          oracle.OFDM_map_carriers(pay, ns, nfft, dc_over, pc, pv))
    with pytest.raises(ofdm.OfdmError):
        ofdm.OFDM_map_carriers(pay[:, :-1], ns, nfft, dc, pc, pv)
    with pytest.raises(ofdm.OfdmError):
        ofdm.get_payload(got, np.array([0.0, 5.0]))            # index 0 is a MATLAB error


@pytest.mark.parametrize("dt,tol", [(np.complex128, 1e-14), (np.complex64, 1e-6)])
def test_equalize_signal(ofdm, oracle, dt, tol):
    rng = np.random.default_rng(2)
    x = crandn(rng, 512, 11).astype(dt)
    h = (crandn(rng, 300) + 1.5).astype(dt)
    got = ofdm.equalize_signal(x, h, 200)
    want = oracle.equalize_signal(x.astype(np.complex128), h.astype(np.complex128), 200)
    assert rel_l2(got, want) < tol
    assert np.all(got[200:] == 0)
    with pytest.raises(ofdm.OfdmError):
        ofdm.equalize_signal(x, h[:100], 200)


@pytest.mark.parametrize("nfft", [1024, 2048])
def test_wave_per_run_modem_fp32(ofdm, oracle, monkeypatch, nfft):
    """modem_run_kernel (fp32, Nfft 1024 / 2048: one wavefront per run of symbols, whole-line stores, CP rows stored from the
    same registers) for runs of 1, 3 and 64 symbols over a ragged symbol count, guards of 0, 2, Nfft/8 and Nfft - 2, against the
    oracle and bit for bit against itself across run lengths; an odd guard takes the cooperative modulator (same results)."""
    rng = np.random.default_rng(nfft)
    ns = 83
    for tg in (0, 2, nfft // 8, nfft - 2, 7):
        x = crandn(rng, nfft, ns).astype(np.complex64)
        y = crandn(rng, nfft + tg, ns).astype(np.complex64)
        want_m = oracle.OFDM_modulator(x.astype(np.complex128), tg)
        want_d = oracle.OFDM_demodulator(y.astype(np.complex128), tg)
        first = None
        for spc in ("1", "3", "64"):
            monkeypatch.setenv("OFDM_MODEM_RUN_SPC", spc)
            gm, gd = ofdm.OFDM_modulator(x, tg), ofdm.OFDM_demodulator(y, tg)
            assert rel_l2(gm, want_m) <= 1e-6 * np.log2(nfft) and rel_l2(gd, want_d) <= 1e-6 * np.log2(nfft)
            assert np.array_equal(gm[:tg], gm[nfft:])                      # CP rows are bitwise copies (OFDM_modulator.m:8-9)
            if first is None:
                first = (gm.copy(), gd.copy())
            else:
                assert np.array_equal(gm, first[0]) and np.array_equal(gd, first[1])
        monkeypatch.delenv("OFDM_MODEM_RUN_SPC")
        monkeypatch.setenv("OFDM_MODEM_NO_RUN", "1")                        # the cooperative kernels on the same data
        gm2, gd2 = ofdm.OFDM_modulator(x, tg), ofdm.OFDM_demodulator(y, tg)
        monkeypatch.delenv("OFDM_MODEM_NO_RUN")
        assert rel_l2(first[0], gm2) < 2e-6 and rel_l2(first[1], gd2) < 2e-6
