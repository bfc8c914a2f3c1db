"""Shared bookkeeping of the driver replays (host-side index / seed logic only, no signal arithmetic).

Every numeric step of a driver goes through `lib` -- by default this package's API, i.e. the HIP
kernels behind the C ABI.  The parity tests hand in an adapter with the same function names over the
CPU oracle, so a driver run on the GPU is compared call for call with the same replay on the oracle.
"""
from __future__ import annotations

import json

import numpy as np

REGISTER = np.array([1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint8)   # T2/Main_model_Task_2.m:36


def default_lib():
    import ofdm_course_amd as ofdm
    ofdm.init()
    return ofdm


def layout_percent(Nfft, N_carrier, Percent_pilot, tail=2):
    """Pilot rule of T1/Main_model.m:14-21 (tail=2) and T5/Main_model_Task_5.m:24-29 (tail=1).

    Returns (allCarriers, pilotCarriers, dataCarriers) as 1-based float64 vectors like `linspace`.
    """
    allCarriers = np.arange(1, Nfft + 1, dtype=np.float64)
    amount_pilots = int(np.floor(Percent_pilot / 100 * N_carrier + 0.5))          # MATLAB round (positive)
    pilot_step = N_carrier // amount_pilots
    pilotCarriers = np.concatenate([allCarriers[0:N_carrier - tail:pilot_step], [float(N_carrier)]])
    dataCarriers = allCarriers[:N_carrier][~np.isin(allCarriers[:N_carrier], pilotCarriers)]
    return allCarriers, pilotCarriers, dataCarriers


def layout_comb(Nfft, N_carrier, comb):
    """Comb rule of T5/Main_model_Task_5.m:18-22 / T5/Task5_part2.m:50-56; comb == 1 falls back to the
    100 % rule (T5/Main_model_Task_5.m:24-33)."""
    if comb == 1:
        return layout_percent(Nfft, N_carrier, 100, tail=1)
    allCarriers = np.arange(1, Nfft + 1, dtype=np.float64)
    pilotCarriers = allCarriers[0:N_carrier:comb]
    dataCarriers = allCarriers[:N_carrier][~np.isin(allCarriers[:N_carrier], pilotCarriers)]
    return allCarriers, pilotCarriers, dataCarriers


def alternating_pilots(amp_pilots, n_pilots, N_symb):
    """pilotValues of T4/Main_model_Task_4.m:28-31 / T5/Task5_part2.m:86-91: +amp, -amp, ... repeated per symbol."""
    col = np.where(np.arange(n_pilots) % 2 == 0, amp_pilots, -amp_pilots).astype(np.complex128)
    return np.repeat(col[:, None], N_symb, axis=1)


def synthetic_bits(Size_Buffer, seed):
    """Stand-in of file_reader (image I/O is out of scope): seeded uniform bits, PCG64."""
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 2, int(Size_Buffer), dtype=np.uint8)


def payload_bits(Size_Buffer, seed, input_bits=None):
    """`file_reader(File, Size_Buffer)` (file_reader.m:2-13): the caller's bits cut to Size_Buffer, else the
    synthetic stand-in."""
    if input_bits is None:
        return synthetic_bits(Size_Buffer, seed)
    b = np.asarray(input_bits, dtype=np.uint8).ravel()
    if b.size < Size_Buffer:
        raise ValueError(f"payload has {b.size} bits, the driver needs {int(Size_Buffer)}")
    return b[:int(Size_Buffer)].copy()


def scramble_per_frame(lib, fn, bits, n_frames):
    """The per-frame register-reset loops (T2/Main_model_Task_2.m:40-50, :126-137): the frame slices
    are equal (Size_Buffer is a multiple of the frame length), so one batched call does all frames."""
    bits = np.asarray(bits, dtype=np.uint8).ravel()
    per = bits.size // n_frames
    assert per * n_frames == bits.size
    batch = getattr(lib, fn + "_frames", None)
    if batch is not None:
        out = batch(REGISTER, bits.reshape(n_frames, per).T)                     # one column per frame
        return np.asarray(out).T.reshape(-1)
    f = getattr(lib, fn)
    return np.concatenate([np.asarray(f(REGISTER, bits[i * per:(i + 1) * per])[0]).ravel() for i in range(n_frames)])


def conv_truncate(lib, x, h):
    """`conv(x, h.', "full")` + truncation to length(x) (T5/Main_model_Task_5.m:126-127)."""
    return np.asarray(lib.apply_channel(np.asarray(x).ravel(), np.asarray(h).ravel())).ravel()


def mse_row(H_true, H_est, N_carrier):
    """(H - He)*(H - He)'/N_carrier of T5/Main_model_Task_5.m:196-205 (host scalar bookkeeping of the driver)."""
    d = np.asarray(H_true).ravel()[:N_carrier] - np.asarray(H_est).ravel()[:N_carrier]
    return float(np.real(np.vdot(d, d)) / N_carrier)


# 3GPP TS 36.101 Annex B.2 delay profiles (ns, dB) -- public tables; lteFadingChannel itself is closed source
DELAY_PROFILES = {
    "EPA": ([0, 30, 70, 90, 110, 190, 410], [0.0, -1.0, -2.0, -3.0, -8.0, -17.2, -20.8]),
    "EVA": ([0, 30, 150, 310, 370, 710, 1090, 1730, 2510], [0.0, -1.5, -1.4, -3.6, -0.6, -9.1, -7.0, -12.0, -16.9]),
    "ETU": ([0, 50, 120, 200, 230, 500, 1600, 2300, 5000], [-1.0, -1.0, -1.0, 0.0, 0.0, 0.0, -3.0, -5.0, -7.0]),
}


def fading_taps(DelayProfile, SamplingRate, Seed):
    """Static (DopplerFreq = 0) tap-delay-line draw standing in for lteFadingChannel (T5/Task5_part2.m:27-34,
    :152): table delays rounded to samples (equal delays merged), power-normalised, random initial phases
    from PCG64(Seed).  The toolbox's fractional-delay filter is not restated (closed source) -- documented
    deviation, DESIGN.md section 5.  Returns channel_taps [[delay, complex amplitude], ...]."""
    d_ns, p_db = DELAY_PROFILES[DelayProfile]
    rng = np.random.Generator(np.random.PCG64(int(Seed)))
    ph = rng.uniform(0.0, 2 * np.pi, len(d_ns))
    amp = np.sqrt(10.0 ** (np.asarray(p_db) / 10.0)) * np.exp(1j * ph)
    delay = np.floor(np.asarray(d_ns) * 1e-9 * SamplingRate + 0.5).astype(int)
    merged = {}
    for d, a in zip(delay, amp):
        merged[d] = merged.get(d, 0) + a
    ds = np.array(sorted(merged))
    a = np.array([merged[d] for d in ds])
    a = a / np.sqrt(np.sum(np.abs(a) ** 2))
    return np.stack([ds.astype(complex), a], axis=1)


def to_jsonable(o):
    if isinstance(o, dict):
        return {k: to_jsonable(v) for k, v in o.items() if not k.startswith("_")}
    if isinstance(o, (list, tuple)):
        return [to_jsonable(v) for v in o]
    if isinstance(o, np.ndarray):
        if np.iscomplexobj(o):
            return {"re": o.real.tolist(), "im": o.imag.tolist()}
        return o.tolist()
    if isinstance(o, (np.floating, np.integer, np.bool_)):
        return o.item()
    return o


def cli(run, description):
    """`python -m ofdm_course_amd.drivers.taskN [--json out.json] [key=value ...]` -> result tables as JSON."""
    import argparse
    ap = argparse.ArgumentParser(description=description)
    ap.add_argument("--json", default=None, help="write the result tables to this file (default: stdout)")
    ap.add_argument("overrides", nargs="*", help="parameter overrides, e.g. Nfft=2048 Constellation=QPSK")
    a = ap.parse_args()
    kw = {}
    for item in a.overrides:
        k, v = item.split("=", 1)
        try:
            kw[k] = json.loads(v)
        except json.JSONDecodeError:
            kw[k] = v
    res = to_jsonable(run(**kw))
    text = json.dumps(res)
    if a.json:
        with open(a.json, "w") as f:
            f.write(text)
    else:
        print(text)
