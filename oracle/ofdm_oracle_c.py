"""ctypes wrapper of the oracle's C twin (oracle/c/ofdm_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "c", "libofdm_oracle_c.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        _lib = C.CDLL(LIB)
        _lib.oracle_rx_chain_task5.restype = C.c_int
        _lib.oracle_c_threads.restype = C.c_int
    return _lib


def rx_chain_task5(rx_frames, Nfft, T_guard, N_carrier, pilotCarriers, dataCarriers, pilotValues_col, K,
                   dominant_taps, dictionary, ref_bits=None, n_threads=1, frame_major=False):
    """Same contract as ofdm_oracle.rx_chain_task5 (dict of bits / errors / H / index).

    frame_major=True: rx_frames is already a C-contiguous complex128 [n_frames, frame_samples] array
    (no copy; lets a caller time the C code alone).  The dict gains `seconds` = time inside the C call."""
    import time
    lib = load()
    if frame_major:
        rx = np.asarray(rx_frames)
        assert rx.dtype == np.complex128 and rx.flags["C_CONTIGUOUS"]
    else:
        rx = np.ascontiguousarray(np.asarray(rx_frames, dtype=np.complex128).T)      # frame-major
    F = rx.shape[0]
    L = int(Nfft) + int(T_guard)
    n_symb = rx.shape[1] // L
    pc = np.ascontiguousarray(np.rint(np.asarray(pilotCarriers).ravel()).astype(np.int32))
    dc = np.ascontiguousarray(np.rint(np.asarray(dataCarriers).ravel()).astype(np.int32))
    pv = np.ascontiguousarray(np.asarray(pilotValues_col, dtype=np.complex128).ravel())
    D = np.ascontiguousarray(np.asarray(dictionary, dtype=np.complex128).ravel())
    bps = int(np.log2(D.size))
    nb = dc.size * n_symb * bps
    bits = np.zeros((F, nb), dtype=np.uint8)
    errors = np.zeros(F, dtype=np.int64)
    H = np.zeros((F, int(N_carrier)), dtype=np.complex128)
    idx = np.zeros((F, int(dominant_taps)), dtype=np.int32)
    ref = None
    if ref_bits is not None:
        ref = np.ascontiguousarray(np.asarray(ref_bits, dtype=np.uint8).reshape(F, nb))
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    t0 = time.perf_counter()
    lib.oracle_rx_chain_task5(p(rx), C.c_int64(F), int(Nfft), int(T_guard), int(n_symb), int(N_carrier), p(pc), pc.size,
                              p(dc), dc.size, p(pv), int(K), int(dominant_taps), p(D), bps, p(ref), p(bits), p(errors),
                              p(H), p(idx), int(n_threads))
    seconds = time.perf_counter() - t0
    return dict(bits=bits, errors=errors, H=H, index=[list(r[r > 0]) for r in idx], seconds=seconds)
