"""ctypes binding of libofdm_mi355x.so (the C ABI declared in include/ofdm_mi355x.h).

There is no CPU fallback: if the shared library is missing or no GPU is visible the calls fail
loudly (OfdmError)."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFDM_LIB_PATH") or os.path.join(HERE, "libofdm_mi355x.so")

OFDM_F32, OFDM_F64, OFDM_HOST, OFDM_DEVICE = 0, 1, 0, 2
OFDM_SOFT_ACF_FALLBACK = 1


class OfdmError(RuntimeError):
    pass


_lib = None

_vp, _i, _i64, _cp, _d = C.c_void_p, C.c_int, C.c_int64, C.c_char_p, C.c_double
_pi, _pi64, _pd = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_double)

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "ofdm_init": [_i],
    "ofdm_shutdown": [],
    "ofdm_last_error_string": [],
    "ofdm_set_stream": [_vp],
    "ofdm_synchronize": [],
    "ofdm_version": [],
    "ofdm_constellation_func": [_cp, _vp, _pi, _i],
    "ofdm_mapping": [_vp, _i64, _cp, _vp, _pi, _i],
    "ofdm_demapping": [_i, _vp, _i64, _cp, _vp, _i],
    "ofdm_Scrambler": [_vp, _vp, _i64, _vp, _i],
    "ofdm_DeScrambler": [_vp, _vp, _i64, _vp, _i],
    "ofdm_Scrambler_frames": [_vp, _vp, _i64, _i64, _vp, _i],
    "ofdm_DeScrambler_frames": [_vp, _vp, _i64, _i64, _vp, _i],
    "ofdm_OFDM_map_carriers": [_vp, _i64, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i],
    "ofdm_get_payload": [_vp, _i, _i64, _vp, _i, _vp, _i],
    "ofdm_OFDM_modulator": [_vp, _vp, _i, _i64, _i, _i],
    "ofdm_OFDM_demodulator": [_vp, _vp, _i, _i64, _i, _i],
    "ofdm_get_MP_channel_resp": [_vp, _vp, _i, _i, _vp, _pi, _vp, _i],
    "ofdm_channel_conv": [_vp, _i64, _vp, _i, _vp, _i],
    "ofdm_Noise": [_d, _vp, _i64, C.c_uint64, C.c_uint32, _vp, _pd, _i],
    "ofdm_channel_conv_frames": [_vp, _i64, _i64, _vp, _i, _vp, _i],
    "ofdm_Noise_frames": [_d, _vp, _i64, _i64, C.c_uint64, C.c_uint32, _vp, _i],
    "ofdm_Noise_frames_snr": [_vp, _vp, _i64, _i64, C.c_uint64, C.c_uint32, _vp, _i],
    "ofdm_add_STO": [_vp, _i64, _i64, _vp, _i],
    "ofdm_add_CFO": [_vp, _i64, _d, _i, _vp, _i],
    "ofdm_add_STO_CFO_frames": [_vp, _i64, _i64, _vp, _vp, _i, _vp, _i],
    "ofdm_AutoCorrFunction": [_vp, _i64, _i, _i, _vp, _pi64, _pd, _i],
    "ofdm_remove_IFO": [_vp, _i64, _i, _vp, _pi, _i],
    "ofdm_fine_sync": [_vp, _i, _i64, _vp, _i, _vp, _i, _i, _i, _vp, _pd, _pd, _i],
    "ofdm_interpolate": [_vp, _vp, _i, _i, C.c_char, _vp, _i],
    "ofdm_estimate_channel": [_vp, _i, _i64, _vp, _i, _vp, _i, _vp, _vp, _vp, _i],
    "ofdm_equalize_signal": [_vp, _i, _i64, _vp, _i, _vp, _i],
    "ofdm_LS_CE": [_vp, _i, _i64, _vp, _vp, _i, _i, _vp, _i],
    "ofdm_MMSE_CE": [_vp, _i, _i64, _vp, _vp, _i, _i, _vp, _i, _d, _vp, _i],
    "ofdm_sensing_matrix": [_vp, _i, _i, _i, _vp, _i],
    "ofdm_MP_estimate": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i],
    "ofdm_OMP_estimate": [_vp, _vp, _i, _i, _i, _i, _d, _vp, _vp, _vp, _pi, _i],
    "ofdm_BER_func": [_vp, _vp, _i64, _pi64, _i],
    "ofdm_MER_func": [_vp, _i64, _cp, _pd, _i],
    "ofdm_calculatePAPR": [_vp, _i64, _pd, _i],
    "ofdm_calculate_window_PAPR": [_vp, _i64, _i, _vp, _i],
    "ofdm_calculateCCDF": [_vp, _i64, _vp, _vp, _pi64, _i],
    "ofdm_rx_plan_create": [C.POINTER(_vp), _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _cp, _i],
    "ofdm_rx_plan_destroy": [_vp],
    "ofdm_rx_plan_set_mmse": [_vp, _vp, _i64, _d, _i],
    "ofdm_rx_chain_task4": [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "ofdm_tx_frames": [_vp, _vp, _i, _d, _i, C.c_uint64, _i64, _i64, _vp, _vp, _vp, _i],
    "ofdm_tx_frames_ex": [_vp, _vp, _i, _d, _i, C.c_uint64, _i64, _i64, _vp, _i, _i64, _i, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "ofdm_rx_plan_set_descrambler": [_vp, _vp],
    "ofdm_rx_plan_frame_bytes": [_vp],
    "ofdm_rx_plan_set_timing": [_vp, _i],
    "ofdm_rx_plan_last_kernel_ms": [_vp, C.POINTER(C.c_float)],
    "ofdm_rx_plan_last_task4_ms": [_vp, C.POINTER(C.c_float)],
    "ofdm_rx_chain_task5": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i],
    "ofdm_task5_part2_tile": [_vp, _vp, _vp, _vp, _i, _i64, _d, _vp, _vp, _vp, _i],
    "ofdm_task5_mse_tile": [_vp, _vp, _vp, _vp, _i, _vp, _i64, C.c_uint64, C.c_uint32, _vp, _i],
}
_RESTYPES = {"ofdm_last_error_string": C.c_char_p, "ofdm_rx_plan_frame_bytes": C.c_int64}


def load(path: str | None = None):
    """dlopen the library and declare every prototype.  Does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OfdmError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(rc: int, what: str) -> int:
    if rc < 0:
        msg = load().ofdm_last_error_string()
        raise OfdmError(f"{what}: {msg.decode() if msg else 'error'} (rc={rc})")
    return rc
