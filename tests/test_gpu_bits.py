"""GPU parity, bit-exact side: constellation tables, mapping, demapping, Scrambler, DeScrambler,
BER_func (+ MER_func tolerance)."""
import numpy as np
import pytest

from conftest import crandn

pytestmark = pytest.mark.gpu

CONSTELLATIONS = ["BPSK", "QPSK", "8PSK", "16QAM", "64QAM", "256QAM"]


@pytest.mark.parametrize("name", CONSTELLATIONS)
def test_constellation_tables(ofdm, oracle, name):
    d, bps = ofdm.constellation_func(name)
    dw, bw = oracle.constellation_func(name)
    assert bps == bw and d.shape == dw.shape
    assert np.max(np.abs(d - dw)) < 1e-15
    assert abs(np.mean(np.abs(d) ** 2) - 1) < 1e-14
    with pytest.raises(ofdm.OfdmError):
        ofdm.constellation_func("32QAM")


@pytest.mark.parametrize("name", CONSTELLATIONS)
def test_mapping_bit_exact(ofdm, oracle, name):
    rng = np.random.default_rng(1)
    _, bps = oracle.constellation_func(name)
    bits = rng.integers(0, 2, 5000 * bps)
    iq, pad = ofdm.mapping(bits, name)
    iqw, padw = oracle.mapping(bits, name)
    assert pad == padw == -1
    assert np.max(np.abs(iq - iqw)) < 1e-15
    # index-exact: the mapped points demap to the same indices
    assert np.array_equal(oracle.demapping(-1, iq, name), bits)
    # padding path for column input; row input needing padding is an error (mapping.m:11)
    if bps > 1:
        col = bits[: 5 * bps + 1].reshape(-1, 1)
        iq2, pad2 = ofdm.mapping(col, name)
        iq2w, pad2w = oracle.mapping(col, name)
        assert pad2 == pad2w == bps - 1 and np.max(np.abs(iq2 - iq2w)) < 1e-15
        with pytest.raises(ofdm.OfdmError):
            ofdm.mapping(bits[: 5 * bps + 1].reshape(1, -1), name)


@pytest.mark.parametrize("name", CONSTELLATIONS)
@pytest.mark.parametrize("dt", [np.complex128, np.complex64])
def test_demapping_bit_exact(ofdm, oracle, name, dt):
    rng = np.random.default_rng(2)
    D, bps = oracle.constellation_func(name)
    n = 20000
    iq = (D[rng.integers(0, len(D), n)] + 0.35 * crandn(rng, n)).astype(dt)
    got = ofdm.demapping(-1, iq, name)
    want = oracle.demapping(-1, iq.astype(np.complex128), name)
    if dt == np.complex128:
        assert np.array_equal(got, want)
    else:
        # fp32 mode: decisions may differ only for points within 1e-4 of a decision boundary
        diff_sym = np.unique(np.nonzero(got != want)[0] // bps)
        d = np.abs(iq.astype(np.complex128)[diff_sym, None] - D[None, :])
        d.sort(axis=1)
        assert np.all(d[:, 1] - d[:, 0] < 1e-4)
        assert diff_sym.size <= 3
    # pad stripping (demapping.m:21-23)
    if dt == np.complex128 and bps > 3:
        assert np.array_equal(ofdm.demapping(3, iq[:50], name), want[: 50 * bps - 3])


@pytest.mark.parametrize("name", ["BPSK", "QPSK", "16QAM", "64QAM", "256QAM"])
def test_demapping_ties_first_minimum_wins(ofdm, oracle, name):
    """Exact ties (blanked carriers = 0+0i, points on decision boundaries).  8PSK is left out: its
    table comes from exp(1i*k*pi/4), whose rounded points are not exactly equidistant from 0, so
    the winner there depends on the last ulp of the host libm (parity unpinned)."""
    D, bps = oracle.constellation_func(name)
    lv = np.unique(np.round(D.real, 12))
    mids = (lv[:-1] + lv[1:]) / 2 if len(lv) > 1 else np.array([0.0])
    pts = [0j] + [complex(m, 0) for m in mids] + [complex(0, m) for m in mids] + [complex(m, m) for m in mids]
    iq = np.array(pts, dtype=np.complex128)
    assert np.array_equal(ofdm.demapping(-1, iq, name), oracle.demapping(-1, iq, name))
    iq32 = np.zeros(17, dtype=np.complex64)
    assert np.array_equal(ofdm.demapping(-1, iq32, name), oracle.demapping(-1, iq32.astype(np.complex128), name))


def test_scrambler_kat_and_oracle(ofdm, oracle):
    reg = oracle.DEFAULT_REGISTER
    sc, r = ofdm.Scrambler(reg, np.zeros(48))
    assert "".join(map(str, sc)) == "000001111110110000100000110100011000010111001010"
    rng = np.random.default_rng(3)
    for n in (1, 5, 13, 14, 15, 16, 31, 1000, 16384, 16385, 40000):
        x = rng.integers(0, 2, n)
        sc, r = ofdm.Scrambler(reg, x)
        scw, rw = (oracle.Scrambler if n <= 1000 else oracle.Scrambler_fast)(reg, x)
        assert np.array_equal(sc, scw), n
        assert np.array_equal(r, rw), n
        d, r2 = ofdm.DeScrambler(reg, sc)
        dw, r2w = oracle.DeScrambler_fast(reg, sc)
        assert np.array_equal(d, x) and np.array_equal(d, dw) and np.array_equal(r2, r2w), n
    # other initial registers
    for seed in range(4):
        rg = rng.integers(0, 2, 15)
        x = rng.integers(0, 2, 777)
        sc, r = ofdm.Scrambler(rg, x)
        scw, rw = oracle.Scrambler(rg, x)
        assert np.array_equal(sc, scw) and np.array_equal(r, rw)
    assert ofdm.Scrambler(reg, np.zeros(0))[0].size == 0


def test_scrambler_frames_reset_per_frame(ofdm, oracle):
    rng = np.random.default_rng(4)
    flen, nfr = 7 * 332 * 4, 9                     # T4 frame: SpF * data carriers * bps
    x = rng.integers(0, 2, (flen, nfr))
    sc = ofdm.Scrambler_frames(oracle.DEFAULT_REGISTER, x)
    for f in range(nfr):
        assert np.array_equal(sc[:, f], oracle.Scrambler_fast(oracle.DEFAULT_REGISTER, x[:, f])[0])
    back = ofdm.DeScrambler_frames(oracle.DEFAULT_REGISTER, sc)
    assert np.array_equal(back, x)
    # self-synchronising: one channel bit error -> exactly three output bit errors
    sc2 = sc.copy(); sc2[1000, 2] ^= 1
    back2 = ofdm.DeScrambler_frames(oracle.DEFAULT_REGISTER, sc2)
    assert np.count_nonzero(back2 != x) == 3


def test_scrambler_roundtrip_full_size(ofdm, oracle):
    """BASELINE config-4 scale: 2000 frames x 14 symbols x 768 carriers x 6 bits (round trip)."""
    rng = np.random.default_rng(6)
    flen, nfr = 14 * 768 * 6, 2000
    x = rng.integers(0, 2, (flen, nfr)).astype(np.uint8)
    sc = ofdm.Scrambler_frames(oracle.DEFAULT_REGISTER, x)
    assert np.array_equal(ofdm.DeScrambler_frames(oracle.DEFAULT_REGISTER, sc), x)
    assert np.array_equal(sc[:, 1234], oracle.Scrambler_fast(oracle.DEFAULT_REGISTER, x[:, 1234])[0])
    assert ofdm.BER_func(x, x) == 0.0


def test_ber_func(ofdm, oracle):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2, 1_000_003)
    b = a.copy()
    flip = rng.choice(a.size, 12345, replace=False)
    b[flip] ^= 1
    assert ofdm.BER_func(a, b, return_count=True) == 12345
    assert ofdm.BER_func(a, b) == oracle.BER_func(a, b)
    with pytest.raises(ofdm.OfdmError):
        ofdm.BER_func(a, b[:-1])


@pytest.mark.parametrize("name", ["QPSK", "16QAM", "64QAM"])
def test_mer_func(ofdm, oracle, name):
    rng = np.random.default_rng(8)
    D, _ = oracle.constellation_func(name)
    iq = D[rng.integers(0, len(D), 30000)] + 0.05 * crandn(rng, 30000)
    assert abs(ofdm.MER_func(iq, name) - oracle.MER_func(iq, name)) < 1e-9
    assert abs(ofdm.MER_func(iq.astype(np.complex64), name) - oracle.MER_func(iq, name)) < 1e-3
