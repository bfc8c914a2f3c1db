"""GPU parity: interpolate / estimate_channel / LS_CE / MMSE_CE / sensing matrix / MP / OMP."""
import numpy as np
import pytest

from conftest import crandn, rel_l2

pytestmark = pytest.mark.gpu

TAPS6 = np.array([[0, 1], [4, .8], [10, .6], [15, .4], [21, .2], [25, .1]])


@pytest.mark.parametrize("method", ["spline", "linear"])
@pytest.mark.parametrize("loc,N", [([1, 5, 9, 13, 17, 21, 25, 29], 32), ([3, 7, 12, 20, 31], 40),
                                   ([1, 40], 40), ([2, 9, 30], 33), (list(range(1, 257, 4)), 256)])
def test_interpolate(ofdm, oracle, method, loc, N):
    rng = np.random.default_rng(len(loc))
    H = crandn(rng, len(loc))
    got = ofdm.interpolate(H, np.array(loc, float), N, method)
    want = oracle.interpolate(H, np.array(loc, float), N, method)
    assert rel_l2(got, want) < 1e-10
    got32 = ofdm.interpolate(H.astype(np.complex64), np.array(loc, float), N, method)
    assert rel_l2(got32, want) < 1e-4


def test_spline_operator_against_scipy(ofdm):
    from scipy.interpolate import CubicSpline
    rng = np.random.default_rng(0)
    x = np.array([1, 4, 9, 10, 17, 30, 31, 45.0])
    y = crandn(rng, len(x))
    got = ofdm.interpolate(y, x, 45, "spline")
    want = CubicSpline(x, y, bc_type="not-a-knot")(np.arange(1, 46.0))
    assert rel_l2(got, want) < 1e-11


def _chan_setup(oracle, nfft=4096, nc=1024, comb=4, ns=14, snr=None, seed=0):
    rng = np.random.default_rng(seed)
    pc, dc = oracle.pilot_layout_comb(nc, comb)
    X = np.zeros((nfft, ns), complex)
    amp = 1.7
    pv = np.full((len(pc), ns), amp, complex)
    X[(pc - 1).astype(int)] = pv
    X[(dc - 1).astype(int)] = crandn(rng, len(dc), ns)
    tx = oracle.OFDM_modulator(X, nfft // 8).ravel(order="F")
    h, H = oracle.get_MP_channel_resp(TAPS6, nfft)
    rx = oracle.apply_channel(tx, h)
    if snr is not None:
        rx, _ = oracle.Noise(snr, rx, rng=rng)
    R = oracle.OFDM_demodulator(rx.reshape((nfft + nfft // 8, ns), order="F"), nfft // 8)
    return dict(R=R, pc=pc, dc=dc, pv=pv, H=H, nfft=nfft, nc=nc, amp=amp)


def test_estimate_channel_and_equalize(ofdm, oracle):
    s = _chan_setup(oracle, nfft=1024, nc=400, comb=6, ns=10, snr=30)
    allc = np.arange(1, 1025.0)
    H, Hp = ofdm.estimate_channel(s["R"], allc, s["pc"], s["pv"])
    Hw, Hpw = oracle.estimate_channel(s["R"], allc, s["pc"], s["pv"])
    assert rel_l2(Hp, Hpw) < 1e-13
    assert rel_l2(H[:400], Hw[:400]) < 1e-10          # beyond N_carrier the spline extrapolates wildly
    assert np.max(np.abs(H - Hw) / (np.abs(Hw) + 1)) < 1e-9
    H32, _ = ofdm.estimate_channel(s["R"].astype(np.complex64), allc, s["pc"], s["pv"].astype(np.complex64))
    assert rel_l2(H32[:400], Hw[:400]) < 1e-4
    eq = ofdm.equalize_signal(s["R"], H, 400)
    assert rel_l2(eq, oracle.equalize_signal(s["R"], Hw, 400)) < 1e-9


def test_ls_ce_kat(ofdm, oracle):
    """Survey KAT (10): noiseless Y = H, comb 4, 256 unit pilots: LS MSE 3.94e-9."""
    nfft, nc = 4096, 1024
    _, H = oracle.get_MP_channel_resp(TAPS6, nfft)
    pc, _ = oracle.pilot_layout_comb(nc, 4)
    Y = H.reshape(-1, 1)
    Xp = np.ones((256, 1), complex)
    got = ofdm.LS_CE(Y, Xp, pc, nc)
    want = oracle.LS_CE(Y, Xp, pc, nc)
    assert rel_l2(got, want) < 1e-11
    mse = np.mean(np.abs(got - H[:nc]) ** 2)
    assert abs(mse - 3.94e-9) < 0.02e-9
    got32 = ofdm.LS_CE(Y.astype(np.complex64), Xp.astype(np.complex64), pc, nc)
    assert rel_l2(got32, want) < 1e-4


def test_ls_ce_uses_first_symbol_only(ofdm, oracle):
    s = _chan_setup(oracle, snr=20, seed=3)
    got = ofdm.LS_CE(s["R"], s["pv"], s["pc"], s["nc"])
    want = oracle.LS_CE(s["R"], s["pv"], s["pc"], s["nc"])
    assert rel_l2(got, want) < 1e-11
    R2 = s["R"].copy(); R2[:, 1:] = 0
    assert np.array_equal(ofdm.LS_CE(R2, s["pv"], s["pc"], s["nc"]), got)


@pytest.mark.parametrize("snr", [0.0, 20.0])
def test_mmse_ce(ofdm, oracle, snr):
    s = _chan_setup(oracle, snr=20, seed=4)
    Hls = oracle.LS_CE(s["R"], s["pv"], s["pc"], s["nc"])
    h = np.fft.ifft(Hls)                                    # T5/Main_model_Task_5.m:179
    got = ofdm.MMSE_CE(s["R"], s["pv"], s["pc"], s["nfft"], s["nc"], h, snr)
    want, _ = oracle.MMSE_CE(s["R"], s["pv"], s["pc"], s["nfft"], s["nc"], h, snr)
    assert rel_l2(got, want) < 1e-9
    got32 = ofdm.MMSE_CE(s["R"].astype(np.complex64), s["pv"].astype(np.complex64), s["pc"], s["nfft"], s["nc"],
                         h.astype(np.complex64), snr)
    assert rel_l2(got32, want) < 1e-4


def test_mmse_kat(ofdm, oracle):
    """Survey KAT (10): noiseless, h = ifft(H_LS), SNR arg 20 dB -> MSE 2.27e-4 (tau_rms 107.65)."""
    nfft, nc = 4096, 1024
    _, H = oracle.get_MP_channel_resp(TAPS6, nfft)
    pc, _ = oracle.pilot_layout_comb(nc, 4)
    Y = H.reshape(-1, 1)
    Xp = np.ones((256, 1), complex)
    Hls = ofdm.LS_CE(Y, Xp, pc, nc)
    got = ofdm.MMSE_CE(Y, Xp, pc, nfft, nc, np.fft.ifft(Hls), 20)
    assert abs(np.mean(np.abs(got - H[:nc]) ** 2) - 2.27e-4) < 0.01e-4


def test_sensing_matrix(ofdm, oracle):
    pc, _ = oracle.pilot_layout_comb(512, 4)
    S = ofdm.sensing_matrix(pc, 2048, 128)
    assert np.max(np.abs(S - oracle.sensing_matrix(pc, 2048, 128))) < 1e-15
    S32 = ofdm.sensing_matrix(pc, 2048, 128, precision="fp32")
    assert S32.dtype == np.complex64 and np.max(np.abs(S32 - S)) < 1e-7


def test_mp_omp_default_setup_kat(ofdm, oracle):
    """Survey KAT (5): noiseless Main_model_Task_5 set-up (Nfft 4096, all 1024 carriers pilots):
    MP picks [1,5,11,16,21,6] MSE 0.02373; OMP picks [1,5,11,16,21,27] MSE 0.002916."""
    nfft, nc = 4096, 1024
    _, H = oracle.get_MP_channel_resp(TAPS6, nfft)
    pc = np.arange(1, nc + 1.0)
    S = ofdm.sensing_matrix(pc, nfft, nc)
    Y = H[:nc].copy()
    Hmp, hmp, kp = ofdm.MP_estimate(Y, S, nfft, 6, return_picks=True)
    Homp, homp, idx = ofdm.OMP_estimate(Y, S, nfft, 6, 20)
    assert list(kp) == [1, 5, 11, 16, 21, 6] and list(idx) == [1, 5, 11, 16, 21, 27]
    mse = lambda A: np.mean(np.abs(H[:nc] - A[:nc]) ** 2)
    assert abs(mse(Hmp) - 0.02373) < 1e-5 and abs(mse(Homp) - 0.002916) < 1e-6
    Hmp_w, hmp_w, _ = oracle.MP_estimate(Y, S, nfft, 6)
    Homp_w, homp_w, _ = oracle.OMP_estimate(Y, S, nfft, 6)
    assert rel_l2(Hmp, Hmp_w) < 1e-9 and rel_l2(Homp, Homp_w) < 1e-9
    assert rel_l2(hmp, hmp_w) < 1e-9 and rel_l2(homp, homp_w) < 1e-9


@pytest.mark.parametrize("dt,tol", [(np.complex128, 1e-9), (np.complex64, 1e-4)])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_mp_omp_noisy_metric_config(ofdm, oracle, dt, tol, seed):
    """Config M pilots (Np = K = 128) at 20 dB."""
    nfft, nc, comb = 2048, 512, 4
    rng = np.random.default_rng(seed)
    _, H = oracle.get_MP_channel_resp(TAPS6, nfft)
    pc, _ = oracle.pilot_layout_comb(nc, comb)
    S = oracle.sensing_matrix(pc, nfft, 128)
    Y = H[(pc - 1).astype(int)] + 0.15 * crandn(rng, len(pc))
    Homp_w, h_w, idx_w = oracle.OMP_estimate(Y.astype(dt).astype(np.complex128), S, nfft, 6)
    Homp, h, idx = ofdm.OMP_estimate(Y.astype(dt), S.astype(dt), nfft, 6, 20)
    assert list(idx) == list(idx_w)
    assert rel_l2(Homp, Homp_w) < tol and rel_l2(h, h_w) < tol
    Hmp_w, hm_w, kp_w = oracle.MP_estimate(Y.astype(dt).astype(np.complex128), S, nfft, 6)
    Hmp, hm, kp = ofdm.MP_estimate(Y.astype(dt), S.astype(dt), nfft, 6, return_picks=True)
    assert list(kp) == list(kp_w) and rel_l2(Hmp, Hmp_w) < tol


def test_omp_early_stop_and_errors(ofdm, oracle):
    """Stopping rule (OMP_estimate.m:20-22): y = two atoms + a component orthogonal to every atom
    (possible because Np > K).  After the two true picks the residual is that component, the third
    atom changes it by ~eps, the relative change is < 1e-2 and the loop breaks with 3 picks."""
    nfft = 512
    rng = np.random.default_rng(0)
    pc = np.arange(1, 129.0)
    S = oracle.sensing_matrix(pc, nfft, 32)
    w = crandn(rng, 128)
    w = w - S @ np.linalg.lstsq(S, w, rcond=None)[0]          # w is orthogonal to span(S)
    y = S[:, 3] * (1 + 0.5j) + S[:, 20] * 0.7 + w
    Hw, hw, idx_w = oracle.OMP_estimate(y, S, nfft, 6)
    H, h, idx = ofdm.OMP_estimate(y, S, nfft, 6)
    assert len(idx_w) == 3 and len(idx) == 3 and sorted(idx[:2]) == [4, 21] == sorted(idx_w[:2])
    assert np.max(np.abs(h - hw)) < 1e-9
    with pytest.raises(ofdm.OfdmError):
        ofdm.MP_estimate(y, oracle.sensing_matrix(pc, nfft, 100), nfft, 3)   # K < Np (MP_estimate.m:10)
