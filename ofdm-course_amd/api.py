"""Host-side mirror of the reference's operator interface (one Python function per `.m` file,
same names, argument order and error behaviour) on top of the C ABI of libofdm_mi355x.so.

Data arguments may be
  * numpy arrays  -> host-pointer flavour of the ABI (what a MEX gateway does): the library stages
    the data through HBM, runs the HIP kernels and copies the results back;
  * torch CUDA tensors -> device-pointer flavour: zero-copy, asynchronous on torch's current stream.

Precision follows the dtype of the primary argument: complex128/float64 -> fp64 kernels (parity
mode, MATLAB is double everywhere), complex64/float32 -> fp32 kernels (throughput mode).
Matrices use MATLAB shapes `[rows, cols]` and are handed to the library in column-major order.
Index vectors are the reference's 1-based carrier indices.

Nothing in this module computes on the CPU: every function ends in a C-ABI call, and a missing
library / GPU raises OfdmError.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib as L
from ._lib import OfdmError  # noqa: F401

try:  # torch is optional plumbing (device memory + streams)
    import torch
except Exception:  # pragma: no cover
    torch = None

__all__ = [
    "OfdmError", "init", "shutdown", "constellation_func", "mapping", "demapping", "Scrambler",
    "DeScrambler", "Scrambler_frames", "DeScrambler_frames", "OFDM_map_carriers", "get_payload",
    "OFDM_modulator", "OFDM_demodulator", "get_MP_channel_resp", "apply_channel", "Noise", "add_STO",
    "add_CFO", "add_STO_CFO_frames", "apply_channel_frames", "Noise_frames", "AutoCorrFunction", "remove_IFO", "fine_sync", "estimate_channel", "equalize_signal",
    "interpolate", "LS_CE", "MMSE_CE", "sensing_matrix", "MP_estimate", "OMP_estimate", "BER_func",
    "MER_func", "calculatePAPR", "calculate_window_PAPR", "calculateCCDF", "RxPlan", "rx_chain_task5", "rx_chain_task4", "task5_part2_tile", "task5_mse_tile", "DEFAULT_REGISTER",
]

DEFAULT_REGISTER = (1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0)   # T5/Main_model_Task_5.m:55


def init(device_id: int = -1):
    L.check(L.load().ofdm_init(int(device_id)), "ofdm_init")


def shutdown():
    L.check(L.load().ofdm_shutdown(), "ofdm_shutdown")


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


class _Call:
    """Marshals the arguments of one C-ABI call (placement + precision from the primary arg)."""

    def __init__(self, primary, f64=None):
        self.lib = L.load()
        self.dev = _is_torch(primary) and primary.is_cuda
        if _is_torch(primary) and not primary.is_cuda:
            primary = primary.numpy()
        if f64 is None:
            if _is_torch(primary):
                f64 = primary.dtype in (torch.complex128, torch.float64)
            else:
                dt = np.asarray(primary).dtype
                f64 = dt not in (np.complex64, np.float32)
        self.f64 = bool(f64)
        self.flags = (L.OFDM_F64 if self.f64 else L.OFDM_F32) | (L.OFDM_DEVICE if self.dev else L.OFDM_HOST)
        self.keep = []
        if self.dev:
            self.device = primary.device
            init(self.device.index if self.device.index is not None else torch.cuda.current_device())
            L.check(self.lib.ofdm_set_stream(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                    "ofdm_set_stream")
        else:
            L.check(self.lib.ofdm_set_stream(None), "ofdm_set_stream")

    # ---- dtypes
    @property
    def cdt(self):
        return np.complex128 if self.f64 else np.complex64

    @property
    def tcdt(self):
        return torch.complex128 if self.f64 else torch.complex64

    # ---- inputs
    def _flat(self, x, npdt, tdt):
        if self.dev:
            if not _is_torch(x):
                x = torch.as_tensor(np.asarray(x))
            x = x.to(device=self.device, dtype=tdt)
            if x.ndim >= 2:
                x = x.movedim(tuple(range(x.ndim)), tuple(reversed(range(x.ndim)))).contiguous()
            else:
                x = x.contiguous()
            self.keep.append(x)
            return C.c_void_p(x.data_ptr()), x.numel()
        if _is_torch(x):
            x = x.cpu().numpy()
        a = np.ascontiguousarray(np.asarray(x, dtype=npdt).ravel(order="F"))
        self.keep.append(a)
        return a.ctypes.data_as(C.c_void_p), a.size

    def cin(self, x):
        return self._flat(x, self.cdt, self.tcdt if self.dev else None)[0]

    def bits_in(self, x):
        if self.dev and _is_torch(x):
            x = (x != 0)
        elif not self.dev:
            x = (np.asarray(x.cpu().numpy() if _is_torch(x) else x) != 0)
        return self._flat(x, np.uint8, torch.uint8 if self.dev else None)[0]

    def idx(self, v):
        """1-based index vector (MATLAB doubles) -> host int32 array."""
        if _is_torch(v):
            v = v.cpu().numpy()
        a = np.asarray(v).ravel()
        r = np.rint(a).astype(np.int32)
        if a.dtype.kind == "f" and not np.array_equal(r, a):
            raise OfdmError("index vector must hold integers")
        r = np.ascontiguousarray(r)
        self.keep.append(r)
        return r.ctypes.data_as(C.c_void_p), r.size

    # ---- outputs
    def cout(self, shape):
        return self._out(shape, self.cdt, self.tcdt if self.dev else None)

    def bits_out(self, n):
        return self._out((int(n),), np.uint8, torch.uint8 if self.dev else None)

    def _out(self, shape, npdt, tdt):
        shape = tuple(int(s) for s in shape)
        if self.dev:
            t = torch.empty(tuple(reversed(shape)), dtype=tdt, device=self.device)
            view = t.movedim(tuple(range(t.ndim)), tuple(reversed(range(t.ndim)))) if t.ndim >= 2 else t
            self.keep.append(t)
            return view, C.c_void_p(t.data_ptr())
        a = np.empty(shape, dtype=npdt, order="F")
        return a, a.ctypes.data_as(C.c_void_p)


def _shape2(x):
    s = tuple(x.shape)
    if len(s) == 1:
        return (s[0], 1)
    if len(s) != 2:
        raise OfdmError("expected a vector or a matrix")
    return s


def _cstr(s):
    return str(s).encode()


# ------------------------------------------------------------------------------------------------
# constellation / mapping / demapping
# ------------------------------------------------------------------------------------------------

def constellation_func(Constellation):
    """T5/constellation_func.m:4-35 -> (Dictionary[2^bps] complex128, Bit_depth_Dict)."""
    lib = L.load()
    d = np.empty(256, dtype=np.complex128)
    bps = C.c_int(0)
    L.check(lib.ofdm_constellation_func(_cstr(Constellation), d.ctypes.data_as(C.c_void_p), C.byref(bps), L.OFDM_F64),
            "constellation_func")
    return d[: 1 << bps.value].copy(), bps.value


def _bps(constellation):
    bps = C.c_int(0)
    L.check(L.load().ofdm_constellation_func(_cstr(constellation), None, C.byref(bps), 0), "constellation_func")
    return bps.value


def mapping(bits, constellation, precision="fp64"):
    """T5/mapping.m:1-25 -> (IQ row, pad).  Row input that needs padding is an error (mapping.m:11)."""
    bps = _bps(constellation)
    shp = tuple(bits.shape)
    n = int(np.prod(shp)) if len(shp) else 1
    is_row = len(shp) == 2 and shp[0] == 1 and shp[1] > 1
    if n % bps != 0 and is_row:
        raise OfdmError("mapping: vertcat dimension mismatch (row input needs padding)")
    call = _Call(bits if _is_torch(bits) else np.zeros(1, np.complex128 if precision == "fp64" else np.complex64),
                 f64=(precision == "fp64"))
    pb = call.bits_in(bits)
    n_iq = (n + bps - 1) // bps
    iq, piq = call.cout((n_iq,))
    pad = C.c_int(0)
    L.check(call.lib.ofdm_mapping(pb, n, _cstr(constellation), piq, C.byref(pad), call.flags), "mapping")
    return iq, pad.value


def demapping(pad, IQ, Constellation):
    """T5/demapping.m:1-25 -> bit row (uint8 0/1)."""
    bps = _bps(Constellation)
    call = _Call(IQ)
    n = int(np.prod(tuple(IQ.shape)))
    nb = n * bps - (int(pad) if pad != -1 else 0)
    if nb < 0:
        raise OfdmError("demapping: pad larger than the bit count")
    out, pout = call.bits_out(nb)
    L.check(call.lib.ofdm_demapping(int(pad), call.cin(IQ), n, _cstr(Constellation), pout, call.flags), "demapping")
    return out


# ------------------------------------------------------------------------------------------------
# scrambler
# ------------------------------------------------------------------------------------------------

def _scr(fn_name, Register, sequence):
    call = _Call(sequence if _is_torch(sequence) else np.zeros(1, np.complex64), f64=False)
    reg = np.ascontiguousarray((np.asarray(Register).ravel() != 0).astype(np.uint8))
    if reg.size != 15:
        raise OfdmError(f"{fn_name}: Register must have 15 elements")
    n = int(np.prod(tuple(sequence.shape)))
    out, pout = call.bits_out(n)
    fn = getattr(call.lib, "ofdm_" + fn_name)
    L.check(fn(reg.ctypes.data_as(C.c_void_p), call.bits_in(sequence), n, pout, call.flags), fn_name)
    return out, reg


def Scrambler(Register, sequence):
    """T5/Scrambler.m:1-28 -> (sc_sequence, Register)."""
    return _scr("Scrambler", Register, sequence)


def DeScrambler(Register, sequence):
    """T5/DeScrambler.m:1-28 -> (dsc_sequence, Register)."""
    return _scr("DeScrambler", Register, sequence)


def _scr_frames(fn_name, Register, seq_matrix):
    call = _Call(seq_matrix if _is_torch(seq_matrix) else np.zeros(1, np.complex64), f64=False)
    reg = np.ascontiguousarray((np.asarray(Register).ravel() != 0).astype(np.uint8))
    flen, nfr = _shape2(seq_matrix)
    out, pout = call._out((flen, nfr), np.uint8, torch.uint8 if call.dev else None)
    fn = getattr(call.lib, "ofdm_" + fn_name)
    L.check(fn(reg.ctypes.data_as(C.c_void_p), call.bits_in(seq_matrix), flen, nfr, pout, call.flags), fn_name)
    return out


def Scrambler_frames(Register, seq_matrix):
    """Per-frame loop of T5/Main_model_Task_5.m:58-69: one column per frame, register reset per column."""
    return _scr_frames("Scrambler_frames", Register, seq_matrix)


def DeScrambler_frames(Register, seq_matrix):
    """Per-frame loop of T5/Main_model_Task_5.m:260-271."""
    return _scr_frames("DeScrambler_frames", Register, seq_matrix)


# ------------------------------------------------------------------------------------------------
# carriers, modulator, demodulator
# ------------------------------------------------------------------------------------------------

def OFDM_map_carriers(QAM_payload, N_symb, Nfft, dataCarriers, pilotCarriers, pilotValues):
    """T5/OFDM_map_carriers.m:2-9."""
    call = _Call(QAM_payload)
    N_symb, Nfft = int(N_symb), int(Nfft)
    pdc, nd = call.idx(dataCarriers)
    ppc, npil = call.idx(pilotCarriers)
    n_pay = int(np.prod(tuple(QAM_payload.shape)))
    if n_pay != nd * N_symb:
        raise OfdmError("OFDM_map_carriers: reshape size mismatch (payload != numel(dataCarriers)*N_symb)")
    pv_n = int(np.prod(tuple(pilotValues.shape))) if hasattr(pilotValues, "shape") else 1
    scalar = 1 if pv_n == 1 else 0
    if not scalar and pv_n != npil * N_symb:
        raise OfdmError("OFDM_map_carriers: pilotValues must be [numel(pilotCarriers) x N_symb] or scalar")
    pv = pilotValues if hasattr(pilotValues, "shape") else np.asarray([pilotValues])
    out, pout = call.cout((Nfft, N_symb))
    L.check(call.lib.ofdm_OFDM_map_carriers(call.cin(QAM_payload), N_symb, Nfft, pdc, nd, ppc, npil,
                                            call.cin(pv), scalar, pout, call.flags), "OFDM_map_carriers")
    return out


def get_payload(RX_OFDM_symbols, dataCarriers):
    """T5/get_payload.m:2-4."""
    call = _Call(RX_OFDM_symbols)
    nfft, ns = _shape2(RX_OFDM_symbols)
    pdc, nd = call.idx(dataCarriers)
    out, pout = call.cout((nd, ns))
    L.check(call.lib.ofdm_get_payload(call.cin(RX_OFDM_symbols), nfft, ns, pdc, nd, pout, call.flags), "get_payload")
    return out


def OFDM_modulator(OFDM_symbols, T_guard):
    """T5/OFDM_modulator.m:2-11."""
    call = _Call(OFDM_symbols)
    nfft, ns = _shape2(OFDM_symbols)
    tg = int(T_guard)
    out, pout = call.cout((nfft + tg, ns))
    L.check(call.lib.ofdm_OFDM_modulator(call.cin(OFDM_symbols), pout, nfft, ns, tg, call.flags), "OFDM_modulator")
    return out


def OFDM_demodulator(OFDM_time_guarded, T_guard):
    """T5/OFDM_demodulator.m:2-10."""
    call = _Call(OFDM_time_guarded)
    rows, ns = _shape2(OFDM_time_guarded)
    tg = int(T_guard)
    nfft = rows - tg
    out, pout = call.cout((nfft, ns))
    L.check(call.lib.ofdm_OFDM_demodulator(call.cin(OFDM_time_guarded), pout, nfft, ns, tg, call.flags),
            "OFDM_demodulator")
    return out


# ------------------------------------------------------------------------------------------------
# channel side
# ------------------------------------------------------------------------------------------------

def get_MP_channel_resp(channel_taps, Nfft):
    """T5/get_MP_channel_resp.m:2-19 -> (impulse_response row, frequency_response row), complex128."""
    lib = L.load()
    init()
    taps = np.atleast_2d(np.asarray(channel_taps))
    nt = taps.shape[0]
    t_re = np.ascontiguousarray(np.real(taps).astype(np.float64).ravel(order="F"))
    t_im = np.ascontiguousarray(np.imag(taps[:, 1]).astype(np.float64)) if np.iscomplexobj(taps) else None
    maxd = int(np.max(np.real(taps[:, 0])))
    h = np.empty(maxd + 1, dtype=np.complex128)
    H = np.empty(int(Nfft), dtype=np.complex128)
    hl = C.c_int(0)
    L.check(lib.ofdm_get_MP_channel_resp(t_re.ctypes.data_as(C.c_void_p),
                                         t_im.ctypes.data_as(C.c_void_p) if t_im is not None else None,
                                         nt, int(Nfft), h.ctypes.data_as(C.c_void_p), C.byref(hl),
                                         H.ctypes.data_as(C.c_void_p), L.OFDM_F64), "get_MP_channel_resp")
    return h[: hl.value], H


def apply_channel(x, h):
    """conv(x, h.', 'full')(1:numel(x)) -- T5/Main_model_Task_5.m:126-127."""
    call = _Call(x)
    n = int(np.prod(tuple(x.shape)))
    hh = np.ascontiguousarray(np.asarray(h.cpu().numpy() if _is_torch(h) else h).ravel().astype(call.cdt))
    out, pout = call.cout((n,))
    L.check(call.lib.ofdm_channel_conv(call.cin(x), n, hh.ctypes.data_as(C.c_void_p), hh.size, pout, call.flags),
            "channel_conv")
    return out


def Noise(SNR, IQ_TX, seed=0, stream=0):
    """T5/Noise.m:1-12 with the build's counter-based generator -> (IQ_RX, N_var)."""
    call = _Call(IQ_TX)
    n = int(np.prod(tuple(IQ_TX.shape)))
    out, pout = call.cout(tuple(IQ_TX.shape) if len(IQ_TX.shape) <= 2 else (n,))
    nv = C.c_double(0)
    L.check(call.lib.ofdm_Noise(float(SNR), call.cin(IQ_TX), n, int(seed), int(stream), pout, C.byref(nv), call.flags),
            "Noise")
    return out, nv.value


def apply_channel_frames(x, h):
    """apply_channel on every column of x = [frame_len, n_frames] independently (Monte-Carlo frames)."""
    call = _Call(x)
    flen, nfr = _shape2(x)
    hh = np.ascontiguousarray(np.asarray(h.cpu().numpy() if _is_torch(h) else h).ravel().astype(call.cdt))
    out, pout = call.cout((flen, nfr))
    L.check(call.lib.ofdm_channel_conv_frames(call.cin(x), flen, nfr, hh.ctypes.data_as(C.c_void_p), hh.size, pout,
                                              call.flags), "channel_conv_frames")
    return out


def Noise_frames(SNR, IQ_TX, seed=0, stream0=0):
    """Noise() applied per column of IQ_TX = [frame_len, n_frames]; column f uses Philox stream stream0+f."""
    call = _Call(IQ_TX)
    flen, nfr = _shape2(IQ_TX)
    out, pout = call.cout((flen, nfr))
    L.check(call.lib.ofdm_Noise_frames(float(SNR), call.cin(IQ_TX), flen, nfr, int(seed), int(stream0), pout,
                                       call.flags), "Noise_frames")
    return out


def add_STO(y, nSTO):
    """T5/add_STO.m:1-10."""
    call = _Call(y)
    n = int(np.prod(tuple(y.shape)))
    out, pout = call.cout((n,))
    L.check(call.lib.ofdm_add_STO(call.cin(y), n, int(nSTO), pout, call.flags), "add_STO")
    return out


def add_STO_CFO_frames(y, nSTO=None, CFO=None, Nfft=1):
    """add_STO then add_CFO per frame of y [frame_len, n_frames] (column = frame), every frame with its own nSTO[f] / CFO[f]
    (T4/Main_model_Task_4.m:101-110 per Monte-Carlo run); None leaves a stage out."""
    call = _Call(y)
    frame_len, n_frames = _shape2(y)
    out, pout = call.cout((frame_len, n_frames))
    ps = pc = None
    if nSTO is not None:
        ps = call._flat(nSTO, np.int64, torch.int64 if call.dev else None)[0]
    if CFO is not None:
        pc = call._flat(CFO, np.float64, torch.float64 if call.dev else None)[0]
    L.check(call.lib.ofdm_add_STO_CFO_frames(call.cin(y), frame_len, n_frames, ps, pc, int(Nfft), pout, call.flags),
            "add_STO_CFO_frames")
    return out


def add_CFO(y, CFO, Nfft):
    """T5/add_CFO.m:1-8."""
    call = _Call(y)
    n = int(np.prod(tuple(y.shape)))
    out, pout = call.cout((n,))
    L.check(call.lib.ofdm_add_CFO(call.cin(y), n, float(CFO), int(Nfft), pout, call.flags), "add_CFO")
    return out


# ------------------------------------------------------------------------------------------------
# synchronisation
# ------------------------------------------------------------------------------------------------

def AutoCorrFunction(RxSignal, WidthWindow, Nfft):
    """T5/AutoCorrFunction.m:1-28 -> (AutoCorr row, TgPosition, FreqOffset).

    Issues the reference's warning and returns TgPosition = 65 when no plateau is found."""
    call = _Call(RxSignal)
    n = int(np.prod(tuple(RxSignal.shape)))
    W, Nfft = int(WidthWindow), int(Nfft)
    n_out = max(n - W - Nfft, 0)
    rho, prho = call.cout((n_out,))
    pos = C.c_int64(0)
    fo = C.c_double(0)
    rc = L.check(call.lib.ofdm_AutoCorrFunction(call.cin(RxSignal), n, W, Nfft, prho, C.byref(pos), C.byref(fo),
                                                call.flags), "AutoCorrFunction")
    if rc == L.OFDM_SOFT_ACF_FALLBACK:
        warnings.warn("AutoCorrFunction: problem locating the guard interval; TgPosition = 65")
    return rho, int(pos.value), fo.value


def remove_IFO(rx_signal, Nfft):
    """T5/remove_IFO.m:1-11 -> (fixed_rx_signal, IFO)."""
    call = _Call(rx_signal)
    n = int(np.prod(tuple(rx_signal.shape)))
    out, pout = call.cout((n,))
    ifo = C.c_int(0)
    L.check(call.lib.ofdm_remove_IFO(call.cin(rx_signal), n, int(Nfft), pout, C.byref(ifo), call.flags), "remove_IFO")
    return out, ifo.value


def fine_sync(rx_signal, pilotCarriers, pilotValues, time_desync, freq_desync, variant="T5", return_estimates=False):
    """T5/fine_sync.m:1-45 (variant='T4' -> T4/fine_sync.m)."""
    call = _Call(rx_signal)
    nfft, ns = _shape2(rx_signal)
    ppc, npil = call.idx(pilotCarriers)
    out, pout = call.cout((nfft, ns))
    tau, ph = C.c_double(0), C.c_double(0)
    L.check(call.lib.ofdm_fine_sync(call.cin(rx_signal), nfft, ns, ppc, npil, call.cin(pilotValues),
                                    int(bool(time_desync)), int(bool(freq_desync)), 1 if variant == "T4" else 0,
                                    pout, C.byref(tau), C.byref(ph), call.flags), "fine_sync")
    if return_estimates:
        return out, tau.value, ph.value
    return out


# ------------------------------------------------------------------------------------------------
# channel estimation / equalisation
# ------------------------------------------------------------------------------------------------

def interpolate(H, pilot_loc, Nfft, method):
    """T5/interpolate.m:1-24."""
    call = _Call(H)
    ploc, npil = call.idx(pilot_loc)
    out, pout = call.cout((int(Nfft),))
    L.check(call.lib.ofdm_interpolate(call.cin(H), ploc, npil, int(Nfft), str(method)[0].lower().encode(), pout,
                                      call.flags), "interpolate")
    return out


def estimate_channel(rx_signal, allCarriers, pilotCarriers, pilotValues):
    """T5/estimate_channel.m:1-9 -> (H_est row over allCarriers, Hest_at_pilots column)."""
    call = _Call(rx_signal)
    nfft, ns = _shape2(rx_signal)
    pall, nall = call.idx(allCarriers)
    ppc, npil = call.idx(pilotCarriers)
    h, ph = call.cout((nall,))
    hp, php = call.cout((npil,))
    L.check(call.lib.ofdm_estimate_channel(call.cin(rx_signal), nfft, ns, pall, nall, ppc, npil,
                                           call.cin(pilotValues), ph, php, call.flags), "estimate_channel")
    return h, hp


def equalize_signal(OFDM_demod, Hest, N_carrier):
    """T5/equalize_signal.m:1-8."""
    call = _Call(OFDM_demod)
    nfft, ns = _shape2(OFDM_demod)
    nc = int(N_carrier)
    nh = int(np.prod(tuple(Hest.shape)))
    if nh < nc:
        raise OfdmError("equalize_signal: index exceeds the number of elements of Hest")
    hflat = Hest.reshape(-1)[:nc] if _is_torch(Hest) else np.asarray(Hest).ravel()[:nc]
    out, pout = call.cout((nfft, ns))
    L.check(call.lib.ofdm_equalize_signal(call.cin(OFDM_demod), nfft, ns, call.cin(hflat), nc, pout, call.flags),
            "equalize_signal")
    return out


def LS_CE(Y, Xp, pilot_loc, N_carrier):
    """T5/LS_CE.m:1-34."""
    call = _Call(Y)
    nfft, ns = _shape2(Y)
    ploc, npil = call.idx(pilot_loc)
    out, pout = call.cout((int(N_carrier),))
    L.check(call.lib.ofdm_LS_CE(call.cin(Y), nfft, ns, call.cin(Xp), ploc, npil, int(N_carrier), pout, call.flags),
            "LS_CE")
    return out


def MMSE_CE(Y, Xp, pilot_loc, Nfft, N_carrier, h, SNR):
    """T5/MMSE_CE.m:1-39."""
    call = _Call(Y)
    nfft, ns = _shape2(Y)
    ploc, npil = call.idx(pilot_loc)
    nh = int(np.prod(tuple(h.shape)))
    out, pout = call.cout((int(N_carrier),))
    L.check(call.lib.ofdm_MMSE_CE(call.cin(Y), nfft, ns, call.cin(Xp), ploc, npil, int(N_carrier), call.cin(h), nh,
                                  float(SNR), pout, call.flags), "MMSE_CE")
    return out


def sensing_matrix(pilotCarriers, Nfft, K, precision="fp64", device=None):
    """S = P*F(:,1:K) of T5/Main_model_Task_5.m:182-190 in closed form (never builds dftmtx)."""
    prim = (torch.zeros(1, dtype=torch.complex128 if precision == "fp64" else torch.complex64, device=device)
            if device is not None else np.zeros(1, np.complex128 if precision == "fp64" else np.complex64))
    call = _Call(prim)
    ppc, npil = call.idx(pilotCarriers)
    out, pout = call.cout((npil, int(K)))
    L.check(call.lib.ofdm_sensing_matrix(ppc, npil, int(Nfft), int(K), pout, call.flags), "sensing_matrix")
    return out


def MP_estimate(Y, sensing_matrix_, Nfft, dominant_taps, return_picks=False):
    """T5/MP_estimate.m:1-34 -> (H_MP row, h_impulse_est column)."""
    call = _Call(Y)
    npil, k = _shape2(sensing_matrix_)
    T = int(dominant_taps)
    H, pH = call.cout((int(Nfft),))
    h, ph = call.cout((int(Nfft),))
    picks = np.zeros(max(T, 1), dtype=np.int32)
    L.check(call.lib.ofdm_MP_estimate(call.cin(Y), call.cin(sensing_matrix_), npil, k, int(Nfft), T, pH, ph,
                                      picks.ctypes.data_as(C.c_void_p), call.flags), "MP_estimate")
    if return_picks:
        return H, h, picks[:T]
    return H, h


def OMP_estimate(Y, sensing_matrix_, Nfft, dominant_taps, SNR_dB=0.0):
    """T5/OMP_estimate.m:1-37 -> (H_OMP row, h_impulse_est row, index)."""
    call = _Call(Y)
    npil, k = _shape2(sensing_matrix_)
    T = int(dominant_taps)
    H, pH = call.cout((int(Nfft),))
    h, ph = call.cout((int(Nfft),))
    index = np.zeros(max(T, 1), dtype=np.int32)
    n_idx = C.c_int(0)
    L.check(call.lib.ofdm_OMP_estimate(call.cin(Y), call.cin(sensing_matrix_), npil, k, int(Nfft), T, float(SNR_dB),
                                       pH, ph, index.ctypes.data_as(C.c_void_p), C.byref(n_idx), call.flags),
            "OMP_estimate")
    return H, h, index[: n_idx.value].astype(np.int64)


# ------------------------------------------------------------------------------------------------
# metrics
# ------------------------------------------------------------------------------------------------

def BER_func(Bit_Tx, Bit_Rx, return_count=False):
    """T5/BER_func.m:1-7."""
    call = _Call(Bit_Tx if _is_torch(Bit_Tx) else np.zeros(1, np.complex64), f64=False)
    n = int(np.prod(tuple(Bit_Tx.shape)))
    if int(np.prod(tuple(Bit_Rx.shape))) != n:
        raise OfdmError("BER_func: arrays have incompatible sizes")
    ne = C.c_int64(0)
    L.check(call.lib.ofdm_BER_func(call.bits_in(Bit_Tx), call.bits_in(Bit_Rx), n, C.byref(ne), call.flags), "BER_func")
    if return_count:
        return ne.value
    return ne.value / n


def MER_func(IQ_RX, Constellation):
    """T5/MER_func.m:1-26."""
    call = _Call(IQ_RX)
    n = int(np.prod(tuple(IQ_RX.shape)))
    mer = C.c_double(0)
    L.check(call.lib.ofdm_MER_func(call.cin(IQ_RX), n, _cstr(Constellation), C.byref(mer), call.flags), "MER_func")
    return mer.value


# ------------------------------------------------------------------------------------------------
# PAPR study (Task 2)
# ------------------------------------------------------------------------------------------------

def calculatePAPR(OFDM_signal):
    """T2/calculatePAPR.m:2-11 -> PAPR in dB (host scalar)."""
    call = _Call(OFDM_signal)
    n = int(np.prod(tuple(OFDM_signal.shape)))
    v = C.c_double(0)
    L.check(call.lib.ofdm_calculatePAPR(call.cin(OFDM_signal), n, C.byref(v), call.flags), "calculatePAPR")
    return v.value


def calculate_window_PAPR(Tx_OFDM_Signal, Nfft):
    """T2/calculate_window_PAPR.m:2-15 -> PAPRs row (float64, length(signal) - Nfft + 1 entries)."""
    call = _Call(Tx_OFDM_Signal)
    n = int(np.prod(tuple(Tx_OFDM_Signal.shape)))
    n_out = max(n - int(Nfft) + 1, 0)
    out, pout = call._out((n_out,), np.float64, torch.float64 if call.dev else None)
    L.check(call.lib.ofdm_calculate_window_PAPR(call.cin(Tx_OFDM_Signal), n, int(Nfft), pout, call.flags),
            "calculate_window_PAPR")
    return out


def calculateCCDF(PAPR_values):
    """T2/calculateCCDF.m:2-6 -> (PAPR_ccdf, CCDF): ecdf abscissae (smallest value twice) and 1 - F."""
    call = _Call(PAPR_values, f64=True)
    n = int(np.prod(tuple(PAPR_values.shape)))
    pv = call._flat(PAPR_values, np.float64, torch.float64 if call.dev else None)[0]
    x, px = call._out((n + 1,), np.float64, torch.float64 if call.dev else None)
    c, pc = call._out((n + 1,), np.float64, torch.float64 if call.dev else None)
    n_out = C.c_int64(0)
    L.check(call.lib.ofdm_calculateCCDF(pv, n, px, pc, C.byref(n_out), call.flags), "calculateCCDF")
    return x[: n_out.value], c[: n_out.value]


# ------------------------------------------------------------------------------------------------
# fused Task-5 RX chain
# ------------------------------------------------------------------------------------------------

class RxPlan:
    """Device-resident description of one Task-5 RX configuration (ofdm_rx_plan_create)."""

    def __init__(self, Nfft, T_guard, N_symb, N_carrier, pilotCarriers, dataCarriers, pilotValues_col, K,
                 dominant_taps, Constellation, precision="fp32", device=None):
        self.lib = L.load()
        init(-1 if device is None else int(device))
        self.f64 = precision == "fp64"
        self.Nfft, self.T_guard, self.N_symb, self.N_carrier = int(Nfft), int(T_guard), int(N_symb), int(N_carrier)
        self.K, self.taps = int(K), int(dominant_taps)
        self.bps = _bps(Constellation)
        pc = np.ascontiguousarray(np.rint(np.asarray(pilotCarriers).ravel()).astype(np.int32))
        dc = np.ascontiguousarray(np.rint(np.asarray(dataCarriers).ravel()).astype(np.int32))
        pv = np.ascontiguousarray(np.asarray(pilotValues_col).ravel().astype(np.complex128 if self.f64 else np.complex64))
        if pv.size != pc.size:
            raise OfdmError("RxPlan: pilotValues_col must have numel(pilotCarriers) entries")
        self.n_pilots, self.n_data = pc.size, dc.size
        h = C.c_void_p(None)
        L.check(self.lib.ofdm_rx_plan_create(C.byref(h), self.Nfft, self.T_guard, self.N_symb, self.N_carrier,
                                             pc.ctypes.data_as(C.c_void_p), pc.size, dc.ctypes.data_as(C.c_void_p),
                                             dc.size, pv.ctypes.data_as(C.c_void_p), self.K, self.taps,
                                             _cstr(Constellation), L.OFDM_F64 if self.f64 else L.OFDM_F32),
                "rx_plan_create")
        self.handle = h
        self.frame_bytes = int(self.lib.ofdm_rx_plan_frame_bytes(h))
        self.frame_bits = self.n_data * self.N_symb * self.bps
        self.frame_samples = (self.Nfft + self.T_guard) * self.N_symb

    def set_mmse(self, h=None, SNR=0.0):
        """Switch the plan's estimator to MMSE_CE(Y, Xp, pilot_loc, Nfft, N_carrier, h, SNR) (T5/MMSE_CE.m:1-39) for
        every frame of a batch; `h=None` returns to OMP_estimate.  h: the impulse response handed to MMSE_CE."""
        if h is None:
            L.check(self.lib.ofdm_rx_plan_set_mmse(self.handle, None, 0, 0.0, L.OFDM_F64), "rx_plan_set_mmse")
            return
        hh = np.ascontiguousarray(np.asarray(h.cpu().numpy() if _is_torch(h) else h).ravel().astype(np.complex128))
        L.check(self.lib.ofdm_rx_plan_set_mmse(self.handle, hh.ctypes.data_as(C.c_void_p), hh.size, float(SNR),
                                               L.OFDM_F64 | L.OFDM_HOST), "rx_plan_set_mmse")

    def set_descrambler(self, Register=None):
        """Per-frame DeScrambler(Register, .) inside rx_chain_task5 / rx_chain_task4 (T5/Main_model_Task_5.m:257-274): the
        demapped bits of every frame are descrambled in the pack stage before they are written / compared.  None = off."""
        if Register is None:
            L.check(self.lib.ofdm_rx_plan_set_descrambler(self.handle, None), "rx_plan_set_descrambler")
            return
        reg = np.ascontiguousarray(np.asarray(Register).ravel().astype(np.uint8))
        if reg.size != 15:
            raise OfdmError("set_descrambler: Register must have 15 entries")
        L.check(self.lib.ofdm_rx_plan_set_descrambler(self.handle, reg.ctypes.data_as(C.c_void_p)), "rx_plan_set_descrambler")

    def tx_frames(self, n_frames, h=None, SNR=None, seed=1, frame0=0, device=None, want_bits=False, Register=None,
                  Time_Delay=None, Freq_Shift=None, noise_first=False, want_draws=False):
        """Synthetic RX frames of this plan's geometry generated on the device (ofdm_tx_frames_ex): payload ->
        [Scrambler(Register, .) per frame] -> mapping -> OFDM_map_carriers -> OFDM_modulator -> channel stages.
        noise_first=True is the reference's order (T5/Main_model_Task_5.m:106-127, T4/Main_model_Task_4.m:94-110,:257-267):
        Noise(SNR) -> add_STO -> add_CFO -> conv(h); the default (False) is add_STO -> add_CFO -> conv(h) -> Noise(SNR).
        h=None: no channel; SNR=None: no noise.  Time_Delay / Freq_Shift: None = off, a number = that value for every
        frame, "random" = the per-frame draw of T4/Main_model_Task_4.m:101-110.
        Returns dict(rx=[frame_samples, n_frames], packed=[n_frames, frame_bytes] (the payload bits)
        (+ bits=[n_frames, frame_bits]) (+ sc_packed = the scrambled bits when Register is given)
        (+ Time_Delay [n_frames] int64, Freq_Shift [n_frames] float64 with want_draws));
        torch CUDA tensors when `device` is given, numpy arrays otherwise."""
        n_frames = int(n_frames)
        cdt_np = np.complex128 if self.f64 else np.complex64
        flags = (L.OFDM_F64 if self.f64 else L.OFDM_F32)
        scr = Register is not None
        if device is not None:
            dev = torch.device(device)
            rx = torch.empty((n_frames, self.frame_samples), dtype=torch.complex128 if self.f64 else torch.complex64,
                             device=dev)
            packed = torch.empty((n_frames, self.frame_bytes), dtype=torch.uint8, device=dev)
            bits = torch.empty((n_frames, self.frame_bits), dtype=torch.uint8, device=dev) if want_bits else None
            scp = torch.empty((n_frames, self.frame_bytes), dtype=torch.uint8, device=dev) if scr else None
            sto = torch.empty((n_frames,), dtype=torch.int64, device=dev) if want_draws else None
            cfo = torch.empty((n_frames,), dtype=torch.float64, device=dev) if want_draws else None
            L.check(self.lib.ofdm_set_stream(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "set_stream")
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
            flags |= L.OFDM_DEVICE
        else:
            L.check(self.lib.ofdm_set_stream(None), "set_stream")
            rx = np.empty((n_frames, self.frame_samples), dtype=cdt_np)
            packed = np.empty((n_frames, self.frame_bytes), dtype=np.uint8)
            bits = np.empty((n_frames, self.frame_bits), dtype=np.uint8) if want_bits else None
            scp = np.empty((n_frames, self.frame_bytes), dtype=np.uint8) if scr else None
            sto = np.empty((n_frames,), dtype=np.int64) if want_draws else None
            cfo = np.empty((n_frames,), dtype=np.float64) if want_draws else None
            ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        hh = None
        if h is not None:
            hh = np.ascontiguousarray(np.asarray(h.cpu().numpy() if _is_torch(h) else h).ravel().astype(cdt_np))
        reg = None
        if scr:
            reg = np.ascontiguousarray(np.asarray(Register).ravel().astype(np.uint8))
            if reg.size != 15:
                raise OfdmError("tx_frames: Register must have 15 entries")

        def mode(v):
            if v is None:
                return 0, 0
            if isinstance(v, str):
                if v != "random":
                    raise OfdmError("tx_frames: Time_Delay / Freq_Shift must be None, a number or 'random'")
                return 2, 0
            return 1, v
        sm, sv = mode(Time_Delay)
        cm, cv = mode(Freq_Shift)
        L.check(self.lib.ofdm_tx_frames_ex(self.handle, hh.ctypes.data_as(C.c_void_p) if hh is not None else None,
                                           0 if hh is None else hh.size, float(SNR if SNR is not None else 0.0),
                                           int(SNR is not None), int(seed), int(frame0), n_frames,
                                           reg.ctypes.data_as(C.c_void_p) if scr else None, sm, int(sv), cm, float(cv),
                                           int(bool(noise_first)), ptr(rx), ptr(packed), ptr(bits), ptr(scp), ptr(sto),
                                           ptr(cfo), flags), "tx_frames")
        out = dict(rx=rx.t() if device is not None else rx.T, packed=packed)
        if want_bits:
            out["bits"] = bits
        if scr:
            out["sc_packed"] = scp
        if want_draws:
            out["Time_Delay"], out["Freq_Shift"] = sto, cfo
        return out

    def set_timing(self, enable=True):
        L.check(self.lib.ofdm_rx_plan_set_timing(self.handle, int(bool(enable))), "rx_plan_set_timing")

    def last_kernel_ms(self):
        """(symbol-1 kernel, OMP kernel, symbols kernel) milliseconds of the last chain call (HIP events)."""
        ms = (C.c_float * 3)()
        L.check(self.lib.ofdm_rx_plan_last_kernel_ms(self.handle, ms), "rx_plan_last_kernel_ms")
        return tuple(float(v) for v in ms)

    def last_stage_ms(self):
        """Stage milliseconds of the last rx_chain_task4 call (HIP events on the launch stream)."""
        ms = (C.c_float * 5)()
        L.check(self.lib.ofdm_rx_plan_last_task4_ms(self.handle, ms), "rx_plan_last_task4_ms")
        return dict(zip(("AutoCorrFunction", "remove_IFO", "OFDM_demodulator", "fine_sync+estimate_channel", "equalize+demap"),
                        (float(v) for v in ms)))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ofdm_rx_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def rx_chain_task5(plan: RxPlan, rx, ref_bits_packed=None, want_h=False, want_index=False):
    """Fused demod -> OMP -> equalise -> payload -> demap -> BER over a batch of frames.

    rx: [(Nfft+Tg)*N_symb, n_frames] complex (numpy -> host flavour, torch.cuda -> device flavour).
    Returns dict(bits=[n_frames, frame_bytes] uint8 packed MSB-first (frame-major), errors=[n_frames] uint32 or None,
    H=[N_carrier, n_frames] or None, index=[taps, n_frames] int32 or None)."""
    call = _Call(rx, f64=plan.f64)
    rows, nfr = _shape2(rx)
    if rows != plan.frame_samples:
        raise OfdmError("rx_chain_task5: rx must have (Nfft+T_guard)*N_symb rows")
    flat_bits, pbits = call._out((plan.frame_bytes * nfr,), np.uint8, torch.uint8 if call.dev else None)
    bits = flat_bits.reshape(nfr, plan.frame_bytes)
    pref = None
    errors, perr = None, None
    if ref_bits_packed is not None:
        ref = ref_bits_packed
        if tuple(ref.shape) != (nfr, plan.frame_bytes):
            raise OfdmError("rx_chain_task5: ref_bits_packed must be [n_frames, frame_bytes]")
        ref = ref.contiguous().view(-1) if _is_torch(ref) else np.ascontiguousarray(ref).reshape(-1)
        pref = call._flat(ref, np.uint8, torch.uint8 if call.dev else None)[0]
        errors, perr = call._out((nfr,), np.uint32, torch.int32 if call.dev else None)
    H, pH = (call.cout((plan.N_carrier, nfr)) if want_h else (None, None))
    idx, pidx = (call._out((plan.taps, nfr), np.int32, torch.int32 if call.dev else None) if want_index else (None, None))
    L.check(call.lib.ofdm_rx_chain_task5(plan.handle, call.cin(rx), nfr, pbits, pref, perr, pH, pidx, call.flags),
            "rx_chain_task5")
    return dict(bits=bits, errors=errors, H=H, index=idx)


def rx_chain_task4(plan: RxPlan, rx, time_desync=1, freq_desync=1, mp_desync=1, ref_bits_packed=None, want_h=False):
    """Task-4 receiver over a batch of frames (T4/Main_model_Task_4.m:278-347 per frame): AutoCorrFunction -> add_STO x2 ->
    add_CFO -> remove_IFO -> OFDM_demodulator -> fine_sync -> estimate_channel -> equalize_signal -> get_payload -> demapping.

    rx: [(Nfft+Tg)*N_symb, n_frames] complex (numpy -> host flavour, torch.cuda -> device flavour).
    Returns dict(bits=[n_frames, frame_bytes] packed demapped bits (not descrambled), errors vs ref_bits_packed or None,
    TgPosition=[n_frames] int64, FreqOffset=[n_frames] float64, IFO=[n_frames] int32, status=[n_frames] int32
    (0 ok, 1 AutoCorrFunction fallback, -1 no IFO line, -2 TgPosition out of range), H=[N_carrier, n_frames] or None)."""
    call = _Call(rx, f64=plan.f64)
    rows, nfr = _shape2(rx)
    if rows != plan.frame_samples:
        raise OfdmError("rx_chain_task4: rx must have (Nfft+T_guard)*N_symb rows")
    flat_bits, pbits = call._out((plan.frame_bytes * nfr,), np.uint8, torch.uint8 if call.dev else None)
    bits = flat_bits.reshape(nfr, plan.frame_bytes)
    pref, errors, perr = None, None, None
    if ref_bits_packed is not None:
        ref = ref_bits_packed
        if tuple(ref.shape) != (nfr, plan.frame_bytes):
            raise OfdmError("rx_chain_task4: ref_bits_packed must be [n_frames, frame_bytes]")
        ref = ref.contiguous().view(-1) if _is_torch(ref) else np.ascontiguousarray(ref).reshape(-1)
        pref = call._flat(ref, np.uint8, torch.uint8 if call.dev else None)[0]
        errors, perr = call._out((nfr,), np.uint32, torch.int32 if call.dev else None)
    tg, ptg = call._out((nfr,), np.int64, torch.int64 if call.dev else None)
    fo, pfo = call._out((nfr,), np.float64, torch.float64 if call.dev else None)
    ifo, pifo = call._out((nfr,), np.int32, torch.int32 if call.dev else None)
    stt, pst = call._out((nfr,), np.int32, torch.int32 if call.dev else None)
    H, pH = (call.cout((plan.N_carrier, nfr)) if want_h else (None, None))
    L.check(call.lib.ofdm_rx_chain_task4(plan.handle, call.cin(rx), nfr, int(bool(time_desync)), int(bool(freq_desync)),
                                         int(bool(mp_desync)), pbits, pref, perr, ptg, pfo, pifo, pst, pH, call.flags),
            "rx_chain_task4")
    return dict(bits=bits, errors=errors, TgPosition=tg, FreqOffset=fo, IFO=ifo, status=stt, H=H)


def task5_part2_tile(plan: RxPlan, tx_noised, taps_list, SNR_dB, ref_bits_packed):
    """One tile of T5/Task5_part2.m:148-205, :269-304: every channel realisation of `taps_list` (each a [(delay, amplitude)]
    array like `channel_taps`, all of the same length) applied to the scenario's noisy TX stream `tx_noised`, then LS_CE,
    MMSE_CE (h = the true CIR, SNR_dB), MP_estimate and OMP_estimate (dominant_taps = the plan's), their NMSE against
    fft(h) and the four equalise / demap / BER passes -- one device-resident pass, only the sums come back.

    ref_bits_packed: the scenario's payload, ONE packed frame [frame_bytes] (every realisation shares the TX frame).
    Returns dict(nmse=[4, n] float64, errors=[4, n] uint32) with rows LS, MMSE, MP, OMP."""
    call = _Call(tx_noised, f64=plan.f64)
    n = len(taps_list)
    t0 = np.asarray(taps_list[0])
    T = t0.shape[0]
    delay = np.empty((n, T), dtype=np.int32)
    amp = np.empty((n, T), dtype=np.complex128)
    for i, t in enumerate(taps_list):
        t = np.asarray(t)
        if t.shape[0] != T:
            raise OfdmError("task5_part2_tile: every realisation needs the same number of channel taps")
        delay[i] = np.real(t[:, 0]).astype(np.int32)
        amp[i] = t[:, 1]
    ref = ref_bits_packed
    ref = ref.contiguous().view(-1) if _is_torch(ref) else np.ascontiguousarray(ref).reshape(-1)
    if ref.shape[0] != plan.frame_bytes:
        raise OfdmError("task5_part2_tile: ref_bits_packed must be one packed frame")
    pref = call._flat(ref, np.uint8, torch.uint8 if call.dev else None)[0]
    nm, pnm = call._out((n, 4), np.float64, torch.float64 if call.dev else None)      # memory [4][n]
    er, per = call._out((n, 4), np.uint32, torch.int32 if call.dev else None)
    ampv = np.ascontiguousarray(amp).view(np.float64)
    L.check(call.lib.ofdm_task5_part2_tile(plan.handle, call.cin(tx_noised), C.c_void_p(delay.ctypes.data),
                                           C.c_void_p(ampv.ctypes.data), T, n, float(SNR_dB), pref, pnm, per, call.flags),
            "task5_part2_tile")
    return dict(nmse=nm.T, errors=er.T)


def task5_mse_tile(plan: RxPlan, Tx, channel_taps, SNRs, seed=0, stream0=0):
    """One tile of the MSE(SNR) sweep of T5/Main_model_Task_5.m:303-346 (ofdm_task5_mse_tile): for every SNR of `SNRs`
    Noise(SNR, Tx) (Philox stream stream0 + i) -> conv(h) truncated -> OFDM_demodulator -> LS_CE, MMSE_CE(h = ifft(H_LS), SNR),
    MP_estimate, OMP_estimate -> mean squared error against fft(h) on 1..N_carrier.  Tx: the clean TX stream of one frame
    (numpy -> host flavour, torch.cuda -> device flavour); channel_taps: [(delay, amplitude)] rows like the script's.
    Returns MSEs [4, n] float64, rows LS, MMSE, MP, OMP."""
    call = _Call(Tx, f64=plan.f64)
    t = np.asarray(channel_taps)
    delay = np.ascontiguousarray(np.real(t[:, 0]).astype(np.int32))
    amp = np.ascontiguousarray(t[:, 1].astype(np.complex128)).view(np.float64)
    snr = np.ascontiguousarray(np.asarray(SNRs, dtype=np.float64).ravel())
    n = snr.size
    ms, pms = call._out((n, 4), np.float64, torch.float64 if call.dev else None)      # memory [4][n]
    L.check(call.lib.ofdm_task5_mse_tile(plan.handle, call.cin(Tx), C.c_void_p(delay.ctypes.data), C.c_void_p(amp.ctypes.data),
                                         delay.size, C.c_void_p(snr.ctypes.data), n, int(seed), int(stream0), pms, call.flags),
            "task5_mse_tile")
    return ms.T
