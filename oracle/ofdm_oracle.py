"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not shipped, not measured, never on the product path.

CPU restatement (numpy / float64) of the reference's Task 1-5 OFDM hot path
(`ladnlav/OFDM-course`, MATLAB).  One Python function per reference `.m`
file, same argument order, 1-based index vectors accepted exactly as the
MATLAB drivers pass them.  Each function cites the reference file:line it
follows ("T5/..." = `/root/reference/Task 5/...`).

PARITY STATUS: **parity unpinned against MATLAB itself, pinned to what the
reference publishes**.  The reference is MATLAB source; neither MATLAB nor
Octave exists in this pipeline and the reference ships no tests, golden
vectors or .mat fixtures, so no output of the reference
could ever be generated here.  What pins this restatement:
(a) every number the reference's READMEs / graphs publish that the path can
reach -- BER(SNR) of Task 3 (15 points), MSE(SNR) of LS / MP / OMP and, through
a one-line `df` variant, MMSE (Task 5), NMSE(SNR) of estimate_channel and the
MER-by-interpolation table (Task 4), PAPR / CCDF of the plain and scrambled image
payload (Task 2) -- at the graphs' reading error (tests/test_oracle_published.py);
(b) the noiseless MP / OMP floors and picks of `Task 5/graphs/mse(snr),
comb1.png` (0.02373 / 0.002916; tests/test_oracle_kat.py); (c) analytic
known-answer tests derived from the code (same file); (d) frozen fixtures
tests/golden/*.npz.  Bit-level agreement with MATLAB's fft / interp1 / pinv
rests on their documented definitions.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module.

Third-party arithmetic restated here (MATLAB built-ins, version unpinned,
>= R2021b implied by `int2bit`): fft/ifft -> numpy.fft (same scaling),
conv -> numpy.convolve, interp1(...,'spline') -> own not-a-knot cubic spline
(cross-checked against scipy.interpolate.CubicSpline in the tests),
pinv -> numpy.linalg.pinv with MATLAB's tolerance max(size)*eps(norm),
A/B -> solve(B.T, A.T).T, bi2de 'left-msb' / int2bit -> MSB-first weights,
dftmtx(N) -> exp(-2*pi*i*j*k/N), circshift(R,1) -> right rotate.
"""
from __future__ import annotations

import math
import warnings

import numpy as np

# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------


def _idx0(v):
    """1-based MATLAB index vector (doubles) -> 0-based int64 numpy index."""
    a = np.asarray(v)
    r = np.rint(a).astype(np.int64)
    if a.dtype.kind == "f" and not np.array_equal(r, a):
        raise ValueError("index vector must hold integers")
    return r.ravel() - 1


def _angle0(z):
    """numpy.angle with angle(0) := 0 irrespective of signed zeros.

    Deviation from MATLAB documented in DESIGN.md (signed-zero hazard,
    T5/fine_sync.m:14,:35): atan2(+-0, -0) = +-pi would pass the 1e-3 masks.
    """
    z = np.asarray(z)
    a = np.angle(z)
    return np.where(z == 0, 0.0, a)


# ----------------------------------------------------------------------------
# constellation / mapping / demapping
# ----------------------------------------------------------------------------

def _gray_decode(g: int) -> int:
    b = 0
    while g:
        b ^= g
        g >>= 1
    return b


def _square_qam(bits_per_axis: int):
    """Square QAM table following the 16QAM rule of T5/constellation_func.m:17-18.

    Index bits = [I-code | Q-code]; I level ascends with the Gray rank of its
    code (00->-3, 01->-1, 11->+1, 10->+3), Q level descends (00->+3, 01->+1,
    11->-1, 10->-3).  64QAM / 256QAM are the BASELINE.json extensions with
    3 / 4 bits per axis ("constellation").
    """
    L = 1 << bits_per_axis
    d = np.empty(L * L, dtype=np.complex128)
    for idx in range(L * L):
        ci = idx >> bits_per_axis
        cq = idx & (L - 1)
        i_lvl = 2 * _gray_decode(ci) - (L - 1)
        q_lvl = -(2 * _gray_decode(cq) - (L - 1))
        d[idx] = complex(i_lvl, q_lvl)
    return d


def constellation_func(Constellation):
    """T5/constellation_func.m:4-35.  Returns (Dictionary[2^bps], bps)."""
    name = str(Constellation)
    if name == "BPSK":
        D = np.array([-1 + 0j, 1 + 0j])
        bps = 1
    elif name == "QPSK":
        D = np.array([-1 - 1j, -1 + 1j, 1 - 1j, 1 + 1j])
        bps = 2
    elif name == "8PSK":
        gray_map = np.array([5, 4, 2, 3, 6, 7, 1, 0], dtype=np.float64)
        D = np.exp(1j * (gray_map * 2 * np.pi / 8))
        bps = 3
    elif name == "16QAM":
        D = np.array([-3 + 3j, -3 + 1j, -3 - 3j, -3 - 1j, -1 + 3j, -1 + 1j, -1 - 3j,
                      -1 - 1j, 3 + 3j, 3 + 1j, 3 - 3j, 3 - 1j, 1 + 3j, 1 + 1j, 1 - 3j, 1 - 1j])
        bps = 4
    elif name == "64QAM":      # extension, see _square_qam
        D = _square_qam(3)
        bps = 6
    elif name == "256QAM":     # extension
        D = _square_qam(4)
        bps = 8
    else:
        # MATLAB: `switch` falls through and `Dictionary` is undefined -> error at :27
        raise ValueError(f"unknown constellation {name!r}")
    N = D.shape[0]
    norm = np.sqrt(np.sum(D * np.conj(D)) / N)     # :28
    D = D / norm                                   # :29
    return D.astype(np.complex128), bps


def mapping(bits, constellation):
    """T5/mapping.m:1-25.  Returns (IQ row, pad); pad = -1 when no padding."""
    dictionary, bit_depth = constellation_func(constellation)
    b = np.asarray(bits)
    is_row = b.ndim == 2 and b.shape[0] == 1 and b.shape[1] > 1
    flat = b.ravel(order="F").astype(np.int64)
    pad = -1
    remainder = flat.size % bit_depth
    if remainder != 0:
        if is_row:
            # :11 vertcat(row, column) is a MATLAB dimension error
            raise ValueError("mapping: vertcat dimension mismatch (row input needs padding)")
        pad = bit_depth - remainder
        flat = np.concatenate([flat, np.zeros(pad, dtype=np.int64)])
    grp = flat.reshape(-1, bit_depth)                       # :15 (column-major reshape then .')
    weights = 1 << np.arange(bit_depth - 1, -1, -1)         # :18 left-msb
    symbols_index = grp @ weights
    return dictionary[symbols_index], pad                   # :21


def demapping(pad, IQ, Constellation):
    """T5/demapping.m:1-25.  Hard decision, first minimum wins, MSB-first bits."""
    D, bps = constellation_func(Constellation)
    iq = np.asarray(IQ).ravel(order="F")
    # :8-10 squared Euclidean distance, real and imaginary parts separately
    dist = (iq.real[None, :] - D.real[:, None]) ** 2 + (iq.imag[None, :] - D.imag[:, None]) ** 2
    idx = np.argmin(dist, axis=0)                           # :12 (first min)
    shifts = np.arange(bps - 1, -1, -1)
    de_bits = ((idx[:, None] >> shifts[None, :]) & 1).reshape(-1)   # :15,:18
    if pad != -1:
        de_bits = de_bits[: de_bits.size - pad]             # :21-23
    return de_bits.astype(np.uint8)


# ----------------------------------------------------------------------------
# scrambler
# ----------------------------------------------------------------------------

DEFAULT_REGISTER = (1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0)   # T5/Main_model_Task_5.m:55


def Scrambler(Register, sequence):
    """T5/Scrambler.m:1-28.  Multiplicative scrambler, taps R(13)^R(14) (:20-21)."""
    R = [int(x) & 1 for x in np.asarray(Register).ravel()]
    seq = np.asarray(sequence).ravel(order="F").astype(np.int64)
    out = np.empty(seq.size, dtype=np.uint8)
    for i in range(seq.size):
        symb_reg = R[12] ^ R[13]                 # array_xor, :18-28
        symbol = symb_reg ^ int(seq[i] & 1)      # :9
        out[i] = symbol
        R = [symbol] + R[:-1]                    # :13-14 circshift then R(1)=feedback
    return out, np.array(R, dtype=np.uint8)


def DeScrambler(Register, sequence):
    """T5/DeScrambler.m:1-28.  Self-synchronising inverse (feedback = received bit)."""
    R = [int(x) & 1 for x in np.asarray(Register).ravel()]
    seq = np.asarray(sequence).ravel(order="F").astype(np.int64)
    out = np.empty(seq.size, dtype=np.uint8)
    for i in range(seq.size):
        fb = int(seq[i] & 1)                     # :8
        out[i] = (R[12] ^ R[13]) ^ fb            # :9-10
        R = [fb] + R[:-1]                        # :12-13
    return out, np.array(R, dtype=np.uint8)


def DeScrambler_fast(Register, sequence):
    """Vectorised DeScrambler (same result; used for large oracle runs)."""
    R0 = np.asarray(Register).ravel().astype(np.uint8) & 1
    s = np.asarray(sequence).ravel(order="F").astype(np.uint8) & 1
    # history h[i] for i<0 from the register: R(j) = s[i-j]
    ext = np.concatenate([R0[::-1], s])          # ext[15+i] = s[i]; ext[15-j] = R(j)
    n = s.size
    out = s ^ ext[15 - 13: 15 - 13 + n] ^ ext[15 - 14: 15 - 14 + n]
    regf = ext[ext.size - 15:][::-1].copy()
    return out, regf


def Scrambler_fast(Register, sequence):
    """Vectorised Scrambler via 13-bit blocks (s[i] depends on s[i-13], s[i-14])."""
    R0 = np.asarray(Register).ravel().astype(np.uint8) & 1
    x = np.asarray(sequence).ravel(order="F").astype(np.uint8) & 1
    n = x.size
    ext = np.zeros(15 + n, dtype=np.uint8)
    ext[:15] = R0[::-1]
    for start in range(0, n, 13):
        stop = min(start + 13, n)
        m = stop - start
        ext[15 + start:15 + stop] = x[start:stop] ^ ext[2 + start:2 + start + m] ^ ext[1 + start:1 + start + m]
    out = ext[15:].copy()
    regf = ext[ext.size - 15:][::-1].copy()
    return out, regf


# ----------------------------------------------------------------------------
# carriers / OFDM modulator / demodulator
# ----------------------------------------------------------------------------

def OFDM_map_carriers(QAM_payload, N_symb, Nfft, dataCarriers, pilotCarriers, pilotValues):
    """T5/OFDM_map_carriers.m:2-9 (T3-T5 variant).  Pilots written after data."""
    N_symb = int(N_symb)
    Nfft = int(Nfft)
    dc = _idx0(dataCarriers)
    pc = _idx0(pilotCarriers)
    out = np.zeros((Nfft, N_symb), dtype=np.complex128)
    data = np.asarray(QAM_payload).ravel(order="F").reshape((dc.size, N_symb), order="F")   # :4
    out[dc, :] = data                                                                        # :6
    pv = np.asarray(pilotValues)
    if pv.ndim == 0 or pv.size == 1:           # scalar broadcast (T3/Main_model_Task_3.m:59)
        out[pc, :] = pv.reshape(())
    else:
        out[pc, :] = pv.reshape((pc.size, N_symb), order="F")                                # :8
    return out


def get_payload(RX_OFDM_symbols, dataCarriers):
    """T5/get_payload.m:2-4."""
    return np.asarray(RX_OFDM_symbols)[_idx0(dataCarriers), :]


def OFDM_modulator(OFDM_symbols, T_guard):
    """T5/OFDM_modulator.m:2-11."""
    X = np.asarray(OFDM_symbols)
    T_guard = int(T_guard)
    t = np.fft.ifft(X, axis=0)                       # :5
    cp = t[t.shape[0] - T_guard:, :]                 # :8
    return np.concatenate([cp, t], axis=0)           # :9


def OFDM_demodulator(OFDM_time_guarded, T_guard):
    """T5/OFDM_demodulator.m:2-10."""
    y = np.asarray(OFDM_time_guarded)
    return np.fft.fft(y[int(T_guard):, :], axis=0)   # :5,:8


# ----------------------------------------------------------------------------
# channel
# ----------------------------------------------------------------------------

def get_MP_channel_resp(channel_taps, Nfft):
    """T5/get_MP_channel_resp.m:2-19.  Returns (h[1 x maxdelay+1], H[1 x Nfft])."""
    taps = np.atleast_2d(np.asarray(channel_taps))
    max_delay = int(np.max(taps[:, 0].real))
    h = np.zeros(max_delay + 1, dtype=taps.dtype if np.iscomplexobj(taps) else np.float64)
    for i in range(taps.shape[0]):
        h[int(round(float(np.real(taps[i, 0]))))] = taps[i, 1]        # :12-14 later duplicate overwrites
    H = np.fft.fft(h, int(Nfft))                                      # :18
    return h, H


def apply_channel(x, h):
    """T5/Main_model_Task_5.m:126-127: conv(x,h.','full') truncated to len(x)."""
    x = np.asarray(x).ravel(order="F")
    h = np.asarray(h).ravel()
    return np.convolve(x, h, mode="full")[: x.size]


def Noise(SNR, IQ_TX, randn_re=None, randn_im=None, rng=None):
    """T5/Noise.m:1-12.  normrnd draws are INPUTS (MATLAB RNG not reproducible)."""
    x = np.asarray(IQ_TX)
    P = np.mean(np.abs(x) ** 2)                         # :3
    NoisePower = P / (10 ** (SNR / 10))                 # :5
    if randn_re is None:
        rng = rng or np.random.default_rng(0)
        randn_re = rng.standard_normal(x.shape)
        randn_im = rng.standard_normal(x.shape)
    noise = np.sqrt(NoisePower / 2) * np.asarray(randn_re) + 1j * np.sqrt(NoisePower / 2) * np.asarray(randn_im)
    return x + noise, np.sqrt(NoisePower)               # :10-11 (N_var is sigma, not sigma^2)


def add_STO(y, nSTO):
    """T5/add_STO.m:1-10."""
    y = np.asarray(y).ravel(order="F")
    n = int(nSTO)
    if n >= 0:
        return np.concatenate([y[n:], np.zeros(min(n, y.size), dtype=y.dtype)])
    return np.concatenate([np.zeros(min(-n, y.size), dtype=y.dtype), y[: max(y.size + n, 0)]])


def add_CFO(y, CFO, Nfft):
    """T5/add_CFO.m:1-8."""
    y = np.asarray(y).ravel(order="F")
    nn = np.arange(y.size, dtype=np.float64)
    return y * np.exp(2j * np.pi * CFO * nn / Nfft)


# ----------------------------------------------------------------------------
# coarse sync
# ----------------------------------------------------------------------------

ACF_THRESHOLD = 0.77          # T5/AutoCorrFunction.m:10, T5/remove_IFO.m:6
ACF_FALLBACK_POSITION = 65    # T5/AutoCorrFunction.m:23


def autocorr_only(RxSignal, WidthWindow, Nfft):
    """T5/AutoCorrFunction.m:3-7 only (rho vector)."""
    x = np.asarray(RxSignal).ravel(order="F").astype(np.complex128)
    W = int(WidthWindow)
    Nfft = int(Nfft)
    n_out = x.size - W - Nfft
    if n_out <= 0:
        return np.zeros(0, dtype=np.complex128)
    prod = x[: x.size - Nfft] * np.conj(x[Nfft:])
    pw = np.abs(x) ** 2
    sw = np.lib.stride_tricks.sliding_window_view
    num = sw(prod, W)[:n_out].sum(axis=1)
    e1 = sw(pw, W)[:n_out].sum(axis=1)
    e2 = sw(pw[Nfft:], W)[:n_out].sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return num / np.sqrt(e1 * e2)


def acf_plateau(amp, WidthWindow):
    """T5/AutoCorrFunction.m:10-24.  Returns (TgPosition (1-based), ok)."""
    th = np.nonzero(np.asarray(amp) > ACF_THRESHOLD)[0] + 1        # :12 (1-based)
    th = th[th > int(WidthWindow)]                                 # :13
    if th.size == 0:
        return ACF_FALLBACK_POSITION, False
    diffs = np.diff(th)
    mask = np.concatenate([[True], np.abs(diffs) != 1])            # :16
    result = np.nonzero(mask)[0] + 1                               # :17 (1-based)
    if result.size < 2:
        return ACF_FALLBACK_POSITION, False                        # :21-24 catch
    first = th[result[0] - 1]
    last = th[result[1] - 1 - 1]
    return int((int(first) + int(last)) // 2), True                # :20


def AutoCorrFunction(RxSignal, WidthWindow, Nfft):
    """T5/AutoCorrFunction.m:1-28.  Returns (AutoCorr row, TgPosition, FreqOffset, ok)."""
    rho = autocorr_only(RxSignal, WidthWindow, Nfft)
    pos, ok = acf_plateau(np.abs(rho), WidthWindow)
    if not ok:
        warnings.warn("AutoCorrFunction: guard-interval plateau not found; fallback 65")
    FreqOffset = -np.angle(rho[pos - 1]) / (2 * np.pi)             # :27
    return rho, pos, float(FreqOffset), ok


def remove_IFO(rx_signal, Nfft):
    """T5/remove_IFO.m:1-11."""
    x = np.asarray(rx_signal).ravel(order="F")
    Nfft = int(Nfft)
    spectrum = np.abs(np.fft.fft(x[Nfft: 2 * Nfft]))               # :5
    inds = np.nonzero(spectrum > ACF_THRESHOLD)[0]
    if inds.size == 0:
        raise IndexError("remove_IFO: no spectral line above 0.77")  # :8 inds(1) errors
    IFO = int(inds[0])                                             # :8 (inds(1)-1, 1-based)
    return add_CFO(x, -IFO, Nfft), IFO


def fine_sync(rx_signal, pilotCarriers, pilotValues, time_desync, freq_desync, variant="T5"):
    """T5/fine_sync.m:1-45 (variant='T4': T4/fine_sync.m also masks diff~=0, :33, and its `taus` has no trailing
    zero: `zeros(1, Np)` on :8 is grown by the loop of :25-30 to numel-1 entries, where T5's :8 allocates numel).

    Documented deviations: nn = 0:size(rx,1)-1 instead of the hard-coded
    0:1024-1 (:24; identical when Nfft=1024); angle(0) := 0 (signed zeros).
    """
    X = np.array(rx_signal, dtype=np.complex128, copy=True)
    pc = _idx0(pilotCarriers)
    pc1 = np.asarray(pilotCarriers, dtype=np.float64).ravel()
    tx = np.asarray(pilotValues)
    if tx.ndim == 1:
        tx = tx[:, None]
    txf = tx.ravel(order="F")
    rxf = X[pc, :].ravel(order="F")                                 # :4
    deltak = pc1[1] - pc1[0]                                        # :6
    q = txf * np.conj(rxf)                                          # :11-12
    taus = np.zeros(txf.size)
    taus[:-1] = _angle0(q[1:] * np.conj(q[:-1])) / (2 * np.pi * deltak)   # :14
    if variant == "T4" and txf.size - 1 >= pc.size:
        taus = taus[:-1]                                            # T4/fine_sync.m:8 + :25-30
    diffs = np.diff(taus)
    if variant == "T4":
        mask = np.concatenate([[False], (np.abs(diffs) < 1e-3) & (diffs != 0)])
    else:
        mask = np.concatenate([[False], np.abs(diffs) < 1e-3])      # :18
    taus_result = taus[mask]
    sel = taus_result[pc.size:]                                     # :20
    tau = np.mean(sel) if sel.size else np.nan
    if time_desync:
        nn = np.arange(X.shape[0], dtype=np.float64)                # :24 (generalised)
        nn_exp = np.exp(-2j * np.pi * tau * nn)
        X = X * np.conj(nn_exp)[:, None]                            # :27  (nn_exp' conjugates)
    rxf = X[pc, :].ravel(order="F")                                 # :32
    qks = _angle0(txf * np.conj(rxf))                               # :35
    selq = qks[np.abs(qks) > 1e-3]
    phase_shift = np.mean(selq) if selq.size else np.nan            # :37
    if freq_desync:
        X = X * np.exp(1j * phase_shift)                            # :40
    return X, float(tau), float(phase_shift)


# ----------------------------------------------------------------------------
# interpolation (MATLAB interp1 linear / 'spline' = not-a-knot cubic, extrapolating)
# ----------------------------------------------------------------------------

def _spline_nak_eval(x, y, xq):
    """Not-a-knot cubic spline through (x, y) evaluated at xq (with extrapolation).

    MATLAB `spline` degeneracies: 2 points -> straight line, 3 points -> parabola.
    y may be complex.  Dense solve in float64/complex128 (oracle, not fast).
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.complex128)
    xq = np.asarray(xq, dtype=np.float64)
    n = x.size
    if n < 2:
        raise ValueError("spline needs at least 2 points")
    if n == 2:
        s = (y[1] - y[0]) / (x[1] - x[0])
        return y[0] + s * (xq - x[0])
    if n == 3:
        # parabola through three points (Newton form)
        d01 = (y[1] - y[0]) / (x[1] - x[0])
        d12 = (y[2] - y[1]) / (x[2] - x[1])
        d012 = (d12 - d01) / (x[2] - x[0])
        return y[0] + d01 * (xq - x[0]) + d012 * (xq - x[0]) * (xq - x[1])
    h = np.diff(x)
    delta = np.diff(y) / h
    A = np.zeros((n, n))
    r = np.zeros(n, dtype=np.complex128)
    # interior continuity of the second derivative, unknowns = slopes s_i
    for i in range(1, n - 1):
        A[i, i - 1] = h[i]
        A[i, i] = 2 * (h[i - 1] + h[i])
        A[i, i + 1] = h[i - 1]
        r[i] = 3 * (h[i] * delta[i - 1] + h[i - 1] * delta[i])
    # not-a-knot ends
    A[0, 0] = h[1]
    A[0, 1] = h[0] + h[1]
    r[0] = ((3 * h[0] + 2 * h[1]) * h[1] * delta[0] + h[0] ** 2 * delta[1]) / (h[0] + h[1])
    A[-1, -1] = h[-2]
    A[-1, -2] = h[-1] + h[-2]
    r[-1] = (h[-1] ** 2 * delta[-2] + (2 * h[-2] + 3 * h[-1]) * h[-2] * delta[-1]) / (h[-2] + h[-1])
    s = np.linalg.solve(A, r)
    seg = np.clip(np.searchsorted(x, xq, side="right") - 1, 0, n - 2)
    t = xq - x[seg]
    hs = h[seg]
    c2 = (3 * delta[seg] - 2 * s[seg] - s[seg + 1]) / hs
    c3 = (s[seg] + s[seg + 1] - 2 * delta[seg]) / hs ** 2
    return y[seg] + t * (s[seg] + t * (c2 + t * c3))


def interp1_spline(x, y, xq):
    return _spline_nak_eval(x, y, xq)


def interp1_linear(x, y, xq):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.complex128)
    xq = np.asarray(xq, dtype=np.float64)
    re = np.interp(xq, x, y.real, left=np.nan, right=np.nan)
    im = np.interp(xq, x, y.imag, left=np.nan, right=np.nan)
    return re + 1j * im


def interpolate(H, pilot_loc, Nfft, method):
    """T5/interpolate.m:1-24.  `Nfft` is whatever the caller passes (N_carrier)."""
    H = np.asarray(H, dtype=np.complex128).ravel()
    loc = np.asarray(pilot_loc, dtype=np.float64).ravel()
    N = int(Nfft)
    if loc[0] > 1:                                                   # :7-10
        slope = (H[1] - H[0]) / (loc[1] - loc[0])
        H = np.concatenate([[H[0] - slope * (loc[0] - 1)], H])
        loc = np.concatenate([[1.0], loc])
    if loc[-1] < N:                                                  # :12-16
        slope = (H[-1] - H[-2]) / (loc[-1] - loc[-2])
        H = np.concatenate([H, [H[-1] + slope * (N - loc[-1])]])
        loc = np.concatenate([loc, [float(N)]])
    xq = np.arange(1, N + 1, dtype=np.float64)
    if str(method)[0].lower() == "l":                                # :18
        return interp1_linear(loc, H, xq)
    return interp1_spline(loc, H, xq)                                # :21


# ----------------------------------------------------------------------------
# channel estimators / equaliser
# ----------------------------------------------------------------------------

def interp1_v5cubic(x, y, xq):
    """MATLAB `interp1(x, y, xq, 'cubic')` as of R2020b: cubic convolution (Keys, a = -1/2; the 'v5cubic' rule)
    on uniformly spaced knots with the end points extended by y(-1) = 3y(0) - 3y(1) + y(2); NaN outside the
    knots.  (Before R2020b 'cubic' meant pchip; the reference needs >= R2021b for int2bit, demapping.m:15.)
    Only used for the interpolation-type table of Task 4/README.md:181-183."""
    x = np.asarray(x, dtype=np.float64).ravel()
    y = np.asarray(y).ravel()
    xq = np.asarray(xq, dtype=np.float64).ravel()
    h = x[1] - x[0]
    if not np.allclose(np.diff(x), h):
        raise ValueError("interp1 'cubic' (v5cubic) needs uniformly spaced knots")
    ye = np.concatenate([[3 * y[0] - 3 * y[1] + y[2]], y, [3 * y[-1] - 3 * y[-2] + y[-3]]])
    t = (xq - x[0]) / h
    k = np.clip(np.floor(t).astype(np.int64), 0, x.size - 2)
    s = t - k
    w = ((-s ** 3 + 2 * s ** 2 - s) / 2, (3 * s ** 3 - 5 * s ** 2 + 2) / 2, (-3 * s ** 3 + 4 * s ** 2 + s) / 2,
         (s ** 3 - s ** 2) / 2)
    out = sum(w[i] * ye[k + i] for i in range(4))
    out = np.asarray(out, dtype=np.result_type(y.dtype, np.float64))
    out[(xq < x[0]) | (xq > x[-1])] = np.nan
    return out


def estimate_channel(rx_signal, allCarriers, pilotCarriers, pilotValues, method="spline"):
    """T5/estimate_channel.m:1-9.  Returns (H_est over allCarriers, Hest_at_pilots).  `method` is the last
    argument of the `interp1` call on :8 -- 'spline' as committed; 'linear' / 'cubic' are the edits behind the
    table of Task 4/README.md:181-183 (linear: NaN outside the pilots like interp1; cubic: interp1_v5cubic)."""
    X = np.asarray(rx_signal)
    pc = _idx0(pilotCarriers)
    tx = np.asarray(pilotValues)
    if tx.ndim == 1:
        tx = tx[:, None]
    Hp = np.mean(X[pc, :] / tx, axis=1)                              # :6
    xk = np.asarray(pilotCarriers, dtype=np.float64).ravel()
    xq = np.asarray(allCarriers, dtype=np.float64).ravel()
    if method == "spline":
        H_est = interp1_spline(xk, Hp, xq)                           # :8
    elif method == "linear":
        H_est = interp1_linear(xk, Hp, xq)
    elif method == "cubic":
        H_est = interp1_v5cubic(xk, Hp, xq)
    else:
        raise ValueError(method)
    return H_est, Hp


def equalize_signal(OFDM_demod, Hest, N_carrier):
    """T5/equalize_signal.m:1-8."""
    X = np.asarray(OFDM_demod)
    N_carrier = int(N_carrier)
    out = np.zeros(X.shape, dtype=np.complex128)
    H = np.asarray(Hest).ravel()[:N_carrier]
    out[:N_carrier, :] = X[:N_carrier, :] / H[:, None]               # :6
    return out


def LS_CE(Y, Xp, pilot_loc, N_carrier):
    """T5/LS_CE.m:1-34.  Linear indexing => first OFDM symbol only (:27-28)."""
    Yf = np.asarray(Y).ravel(order="F")
    Xpf = np.asarray(Xp).ravel(order="F")
    loc = _idx0(pilot_loc)
    Np = loc.size
    LS_est = Yf[loc] / Xpf[:Np]                                      # :28
    return interpolate(LS_est, pilot_loc, N_carrier, "spline")       # :31


def MMSE_CE(Y, Xp, pilot_loc, Nfft, N_carrier, h, SNR, df=None):
    """T5/MMSE_CE.m:1-39 (quirks kept: Nps*(K1-K2), df=1/N_carrier, re-interpolation :38).

    `df=None` is the committed file (`df = 1/N_carrier`, :25).  The comment on that line gives the textbook
    form `1/(ts*Nfft)`; passing `df=1/Nfft` is the variant that reproduces the MMSE curve of the published
    graph `Task 5/graphs/mse(snr), comb1.png` (tests/test_oracle_published.py)."""
    Y = np.asarray(Y)
    Xp = np.asarray(Xp)
    if Y.ndim == 1:
        Y = Y[:, None]
    if Xp.ndim == 1:
        Xp = Xp[:, None]
    N_carrier = int(N_carrier)
    snr = 10 ** (SNR * 0.1)                                          # :13
    loc = _idx0(pilot_loc)
    Np = loc.size
    pl = np.asarray(pilot_loc, dtype=np.float64).ravel()
    Nps = pl[1] - pl[0]                                              # :15
    H_tilde = Y[loc, 0] / Xp[:Np, 0]                                 # :17
    h = np.asarray(h, dtype=np.complex128).ravel()
    k = np.arange(h.size, dtype=np.float64)                          # :19
    hh = np.sum(h * np.conj(h))                                      # :20
    tmp = h * np.conj(h) * k                                         # :21
    r = np.sum(tmp) / hh                                             # :22
    r2 = np.sum(tmp * k) / hh                                        # :23
    tau_rms = np.sqrt(r2 - r ** 2)                                   # :24
    if df is None:
        df = 1.0 / N_carrier                                         # :25
    j2pi_tau_df = 1j * 2 * np.pi * tau_rms * df                      # :26
    K1 = np.arange(N_carrier, dtype=np.float64)[:, None]
    K2 = np.arange(Np, dtype=np.float64)[None, :]
    rf = 1.0 / (1 + j2pi_tau_df * Nps * (K1 - K2))                   # :30
    K3 = np.arange(Np, dtype=np.float64)[:, None]
    rf2 = 1.0 / (1 + j2pi_tau_df * Nps * (K3 - K2))                  # :33
    Rpp = rf2 + np.eye(Np) / snr                                     # :35
    # :36  (Rhp/Rpp)*H_tilde.'  -- mrdivide first
    RhpRppInv = np.linalg.solve(Rpp.T, rf.T).T
    H_MMSE = RhpRppInv @ H_tilde
    return interpolate(H_MMSE[:Np], pilot_loc, N_carrier, "spline"), float(np.real(tau_rms))   # :38


def sensing_matrix(pilotCarriers, Nfft, K):
    """T5/Main_model_Task_5.m:182-190: S = P*F(:,1:K), closed form (never dftmtx)."""
    p = (_idx0(pilotCarriers)).astype(np.int64)[:, None]
    k = np.arange(int(K), dtype=np.int64)[None, :]
    ph = (p * k) % int(Nfft)
    return np.exp(-2j * np.pi * ph / float(Nfft))


def MP_estimate(Y, sensing_matrix_, Nfft, dominant_taps):
    """T5/MP_estimate.m:1-34.  Returns (H_MP row[Nfft], h[Nfft], kp (1-based picks))."""
    S = np.asarray(sensing_matrix_, dtype=np.complex128)
    Np = S.shape[0]
    if S.shape[1] < Np:
        raise IndexError("MP_estimate: loop bound is Np columns (:10) but K < Np")
    residue = np.asarray(Y, dtype=np.complex128).ravel().copy()
    T = int(dominant_taps)
    kp = np.zeros(T, dtype=np.int64)
    x = np.zeros(T, dtype=np.complex128)
    A = S[:, :Np]
    nrm2 = np.sum(np.abs(A) ** 2, axis=0)
    for i1 in range(T):
        proj = np.abs(A.conj().T @ residue) ** 2 / nrm2              # :15
        for j in range(i1):
            proj[kp[j] - 1] = -100.0                                 # :11-12
        kp[i1] = int(np.argmax(proj)) + 1                            # :18
        a = S[:, kp[i1] - 1]
        n2 = np.sum(np.abs(a) ** 2)
        x[i1] = (a.conj() @ residue) / n2                            # :22
        residue = residue - a * ((a.conj() @ residue) / n2)          # :21,:23
    h = np.zeros(int(Nfft), dtype=np.complex128)
    for i1 in range(T):
        h[kp[i1] - 1] = x[i1]                                        # :28-30
    return np.fft.fft(h), h, kp                                      # :33


def _pinv_matlab(A):
    """MATLAB pinv: tolerance max(size(A))*eps(norm(A))."""
    U, s, Vh = np.linalg.svd(A, full_matrices=False)
    tol = max(A.shape) * np.spacing(s[0]) if s.size else 0.0
    keep = s > tol
    return (Vh[keep].conj().T / s[keep]) @ U[:, keep].conj().T


def OMP_estimate(Y, sensing_matrix_, Nfft, dominant_taps, SNR_dB=0.0):
    """T5/OMP_estimate.m:1-37.  Returns (H_OMP row[Nfft], h row[Nfft], index (1-based))."""
    S = np.asarray(sensing_matrix_, dtype=np.complex128)
    y = np.asarray(Y, dtype=np.complex128).ravel()
    T = int(dominant_taps)
    index = [int(np.argmax(np.abs(S.conj().T @ y))) + 1]             # :7
    A = S[:, [index[0] - 1]]
    x = _pinv_matlab(A) @ y                                          # :9
    res_prev = y - A @ x                                             # :11
    for _ in range(2, T + 1):
        index.append(int(np.argmax(np.abs(S.conj().T @ res_prev))) + 1)   # :14
        A = np.concatenate([A, S[:, [index[-1] - 1]]], axis=1)       # :16
        x = _pinv_matlab(A) @ y                                      # :17
        res = y - A @ x                                              # :18
        nprev = np.linalg.norm(res_prev)
        with np.errstate(divide="ignore", invalid="ignore"):
            stop = (np.linalg.norm(res - res_prev) / nprev) < 1e-2   # :20
        res_prev = res
        if stop:
            break
    h = np.zeros(int(Nfft), dtype=np.complex128)
    for i1, idx in enumerate(index):
        h[idx - 1] = x[i1]                                           # :31-33 (last write wins)
    return np.fft.fft(h), h, np.array(index, dtype=np.int64)         # :36


# ----------------------------------------------------------------------------
# metrics
# ----------------------------------------------------------------------------

def BER_func(Bit_Tx, Bit_Rx):
    """T5/BER_func.m:1-7."""
    a = np.asarray(Bit_Tx).ravel()
    b = np.asarray(Bit_Rx).ravel()
    return np.count_nonzero(a != b) / a.size


def MER_func(IQ_RX, Constellation):
    """T5/MER_func.m:1-26."""
    D, _ = constellation_func(Constellation)
    iq = np.asarray(IQ_RX).ravel(order="F")
    idx = np.argmin(np.abs(iq[None, :] - D[:, None]), axis=0)        # :10-16 strict '<' = first min
    ideal = D[idx]
    sum1 = np.sum(ideal.real ** 2 + ideal.imag ** 2)
    sum2 = np.sum((ideal - iq).real ** 2 + (ideal - iq).imag ** 2)
    return 10 * np.log10(sum1 / sum2)


# ----------------------------------------------------------------------------
# PAPR study of Task 2
# ----------------------------------------------------------------------------

def calculatePAPR(OFDM_signal):
    """T2/calculatePAPR.m:2-11: 10 log10(max(abs(x))^2 / mean(abs(x).^2))."""
    x = np.asarray(OFDM_signal, dtype=np.complex128).ravel(order="F")
    a = np.abs(x)
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(10 * np.log10(np.max(a) ** 2 / np.mean(a ** 2))) if x.size else float("nan")


def calculate_window_PAPR(Tx_OFDM_Signal, Nfft):
    """T2/calculate_window_PAPR.m:2-15: PAPR of every window x(i : i+Nfft-1), each recomputed from scratch."""
    x = np.asarray(Tx_OFDM_Signal, dtype=np.complex128).ravel(order="F")
    n_PAPRs = x.size - int(Nfft) + 1                                  # :4
    if n_PAPRs <= 0:
        return np.zeros(0)
    a = np.abs(x)
    w = np.lib.stride_tricks.sliding_window_view(a, int(Nfft))        # :8-13, one row per window
    out = np.empty(n_PAPRs)
    with np.errstate(divide="ignore", invalid="ignore"):
        for i0 in range(0, n_PAPRs, 4096):                            # bounded temporaries
            blk = w[i0:i0 + 4096]
            out[i0:i0 + 4096] = 10 * np.log10(np.max(blk, axis=1) ** 2 / np.mean(blk ** 2, axis=1))
    return out


def calculateCCDF(PAPR_values):
    """T2/calculateCCDF.m:2-6: [F, x] = ecdf(values) (NaN dropped; x = sorted distinct values with the first one
    repeated, F = [0, cumulative fraction]) -> (PAPR_ccdf = x, CCDF = 1 - F)."""
    v = np.asarray(PAPR_values, dtype=np.float64).ravel()
    v = v[~np.isnan(v)]
    if v.size == 0:
        return np.zeros(0), np.zeros(0)
    u, counts = np.unique(v + 0.0, return_counts=True)
    F = np.concatenate([[0.0], np.cumsum(counts) / v.size])
    return np.concatenate([[u[0]], u]), 1.0 - F


# ----------------------------------------------------------------------------
# pilot layouts of the drivers
# ----------------------------------------------------------------------------

def pilot_layout_percent(Nfft, N_carrier, Percent_pilot, tail):
    """T1/Main_model.m:14-24 (tail=1), T4/Main_model_Task_4.m:14-24 (tail=2).

    pilotCarriers = [1:pilot_step:N_carrier-tail, N_carrier]; returns 1-based
    (pilotCarriers, dataCarriers) as float64 like `linspace` produces.
    """
    amount = int(np.floor(Percent_pilot / 100 * N_carrier + 0.5))    # MATLAB round (half away from zero)
    step = N_carrier // amount
    pc = list(range(1, N_carrier - tail + 1, step)) + [N_carrier]
    pc = np.array(pc, dtype=np.float64)
    allc = np.arange(1, N_carrier + 1, dtype=np.float64)
    dc = allc[~np.isin(allc, pc)]
    return pc, dc


def pilot_layout_comb(N_carrier, comb):
    """T5/Main_model_Task_5.m:18-35 / T5/Task5_part2.m:50-56,:77 for comb != 1."""
    pc = np.arange(1, N_carrier + 1, comb, dtype=np.float64)
    allc = np.arange(1, N_carrier + 1, dtype=np.float64)
    dc = allc[~np.isin(allc, pc)]
    return pc, dc


# ----------------------------------------------------------------------------
# the metric chain
# ----------------------------------------------------------------------------

def rx_chain_task5(rx_frames, Nfft, T_guard, N_carrier, pilotCarriers, dataCarriers, pilotValues_col,
                   K, dominant_taps, Constellation, ref_bits=None, Register=None, want_iq=False):
    """Full Task-5 RX per frame: demod -> OMP (symbol 1) -> equalise -> payload -> demap -> BER.

    Call order of T5/Task5_part2.m:169-193,:272,:279-303 with Y formed as
    RX(pilots,1)./pilotValues(:,1) (:190) and S from T5/Main_model_Task_5.m:182-190.

    rx_frames : complex [(Nfft+Tg)*S, F]  (one column per frame, S symbols each)
    returns dict(bits=[F, Nd*S*bps] uint8, errors=[F] int64, H=[F, N_carrier], index=list
                 [, iq=[F, Nd*S] the equalised payload points handed to demapping, want_iq=True])
    """
    rx = np.asarray(rx_frames)
    if rx.ndim == 1:
        rx = rx[:, None]
    L = Nfft + T_guard
    S_sym = rx.shape[0] // L
    F = rx.shape[1]
    Smat = sensing_matrix(pilotCarriers, Nfft, K)
    pc = _idx0(pilotCarriers)
    pv = np.asarray(pilotValues_col).ravel()
    _, bps = constellation_func(Constellation)
    nd = len(np.asarray(dataCarriers).ravel())
    bits = np.zeros((F, nd * S_sym * bps), dtype=np.uint8)
    errors = np.zeros(F, dtype=np.int64)
    Hs = np.zeros((F, N_carrier), dtype=np.complex128)
    picks = []
    iqs = np.zeros((F, nd * S_sym), dtype=np.complex128) if want_iq else None
    for f in range(F):
        Xf = OFDM_demodulator(rx[:, f].reshape((L, S_sym), order="F"), T_guard)
        Yp = Xf[pc, 0] / pv
        H_omp, _, index = OMP_estimate(Yp, Smat, Nfft, dominant_taps, 0.0)
        picks.append(index)
        Hs[f] = H_omp[:N_carrier]
        eq = equalize_signal(Xf, H_omp, N_carrier)
        iq = get_payload(eq, dataCarriers).ravel(order="F")
        bits[f] = demapping(-1, iq, Constellation)
        if want_iq:
            iqs[f] = iq
        if Register is not None:                                   # T5/Main_model_Task_5.m:257-274, register reset per frame
            bits[f] = DeScrambler_fast(Register, bits[f])[0]
        if ref_bits is not None:
            errors[f] = np.count_nonzero(bits[f] != np.asarray(ref_bits)[f])
    out = dict(bits=bits, errors=errors, H=Hs, index=picks)
    if want_iq:
        out["iq"] = iqs
    return out


# ----------------------------------------------------------------------------
# counter-based RNG restatement (Philox4x32-10) used by the build's AWGN generator
# ----------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., SC'11).  counter: [n,4] uint32, key: [2] uint32."""
    c = np.array(counter, dtype=np.uint64, copy=True).reshape(-1, 4)
    k0 = int(key[0]) & 0xFFFFFFFF
    k1 = int(key[1]) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _PHILOX_M0 * c[:, 0]
        p1 = _PHILOX_M1 * c[:, 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        n0 = hi1 ^ c[:, 1] ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c[:, 3] ^ np.uint64(k1)
        n3 = lo0
        c = np.stack([n0, n1, n2, n3], axis=1)
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def awgn_philox(n, seed, stream=0):
    """Standard-normal complex draws for sample indices 0..n-1 (re, im), float64.

    Sample i uses counter (i_lo, i_hi, stream, 0); u = (r + 0.5) * 2^-32;
    Box-Muller: re = sqrt(-2 ln u0) cos(2 pi u1), im = sqrt(-2 ln u0) sin(2 pi u1),
    with (u0,u1) = words (0,1).  Same definition as csrc/channel.hip.
    """
    i = np.arange(n, dtype=np.uint64)
    ctr = np.stack([i & np.uint64(0xFFFFFFFF), i >> np.uint64(32),
                    np.full(n, stream, dtype=np.uint64), np.zeros(n, dtype=np.uint64)], axis=1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    r = philox4x32_10(ctr, key).astype(np.float64)
    u0 = (r[:, 0] + 0.5) * 2.0 ** -32
    u1 = (r[:, 1] + 0.5) * 2.0 ** -32
    rad = np.sqrt(-2.0 * np.log(u0))
    return rad * np.cos(2 * np.pi * u1), rad * np.sin(2 * np.pi * u1)


def payload_codes_philox(n_symbols, bps, seed, stream=0):
    """Payload draw of ofdm_tx_frames (csrc/ofdm_txgen.hip): QAM symbol j of a frame uses counter (j_lo, j_hi, stream, 1),
    key = seed; its code is the top `bps` bits of word 0.  An INPUT convention of the build (the reference reads an image)."""
    j = np.arange(int(n_symbols), dtype=np.uint64)
    ctr = np.stack([j & np.uint64(0xFFFFFFFF), j >> np.uint64(32), np.full(j.size, stream, dtype=np.uint64),
                    np.ones(j.size, dtype=np.uint64)], axis=1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    return (philox4x32_10(ctr, key)[:, 0] >> np.uint32(32 - int(bps))).astype(np.int64)


def sto_cfo_draw_philox(span, seed, stream=0):
    """Per-frame impairment draw of ofdm_tx_frames_ex (csrc/ofdm_txgen.hip) standing in for T4/Main_model_Task_4.m:101-110:
    counter (0, 0, stream, 2), key = seed; Time_Delay = word0 mod span (span = Nfft + T_Guard + 1: randi([0, Nfft+T_Guard])),
    Freq_Shift = (word1 mod 31) + ((word2 + 0.5) 2^-32 - 0.5)  (randi([0,30]) + (rand - 0.5)).  An INPUT convention."""
    ctr = np.array([[0, 0, stream, 2]], dtype=np.uint64)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    r = philox4x32_10(ctr, key)[0]
    return int(r[0]) % int(span), float(int(r[1]) % 31) + ((float(r[2]) + 0.5) * 2.0 ** -32 - 0.5)


def tx_frame(payload_bits, Nfft, T_guard, N_symb, dataCarriers, pilotCarriers, pilotValues, Constellation, h=None, SNR=None,
             noise=None, Register=None, Time_Delay=None, Freq_Shift=None, noise_first=True):
    """One frame through the TX + channel sections of the drivers, composed from the functions above:
    [Scrambler] (T5/Main_model_Task_5.m:55-69) -> mapping (:72) -> OFDM_map_carriers (:75) -> OFDM_modulator (:82-84), then
    noise_first=True, the reference's order: Noise (:106-109) -> add_STO / add_CFO (T4/Main_model_Task_4.m:101-110) ->
    conv(h) (:112-127); noise_first=False: add_STO -> add_CFO -> conv -> Noise.  `noise` = (randn_re, randn_im) draws (inputs).
    Returns (rx stream, scrambled bits or None)."""
    bits = np.asarray(payload_bits).ravel().astype(np.uint8)
    sc = None
    if Register is not None:
        sc, _ = Scrambler_fast(Register, bits)
    iq, _ = mapping(sc if sc is not None else bits, Constellation)
    X = OFDM_map_carriers(iq, N_symb, Nfft, dataCarriers, pilotCarriers, pilotValues)
    y = OFDM_modulator(X, T_guard).ravel(order="F")

    def imp(v):
        if Time_Delay is not None:
            v = add_STO(v, Time_Delay)
        if Freq_Shift is not None:
            v = add_CFO(v, Freq_Shift, Nfft)
        return v

    def awgn(v):
        return Noise(SNR, v, noise[0], noise[1])[0] if SNR is not None else v
    if noise_first:
        y = imp(awgn(y))
        if h is not None:
            y = apply_channel(y, h)
    else:
        y = imp(y)
        if h is not None:
            y = apply_channel(y, h)
        y = awgn(y)
    return y, sc


def payload_bits_philox(n_symbols, bps, seed, stream=0):
    """The frame's bit vector for `payload_codes_philox`: MSB of each symbol first (mapping.m:15-18)."""
    codes = payload_codes_philox(n_symbols, bps, seed, stream)
    sh = np.arange(int(bps) - 1, -1, -1)
    return ((codes[:, None] >> sh[None, :]) & 1).astype(np.uint8).ravel()
