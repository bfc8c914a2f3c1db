"""Frame geometry of the BASELINE configurations and synthetic-frame generation.

Everything numeric here runs through the product's own HIP kernels (mapping -> OFDM_map_carriers
-> OFDM_modulator -> channel conv -> Noise); numpy is used only for seeded bit draws, index
bookkeeping and bit packing.  Mirrors the TX/channel call order of
T5/Main_model_Task_5.m:50-127 and T5/Task5_part2.m:84-134.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

TAPS6 = np.array([[0, 1.0], [4, .8], [10, .6], [15, .4], [21, .2], [25, .1]])   # T5/Main_model_Task_5.m:112-119


@dataclass
class FrameConfig:
    """One row of SURVEY.md section 8 'Config table'."""
    name: str
    Nfft: int
    N_carrier: int
    comb: int
    Constellation: str
    N_symb: int = 14                       # Amount_OFDM_Frames*Amount_ODFM_SpF (T5/Main_model_Task_5.m:9-11)
    taps: np.ndarray = field(default_factory=lambda: TAPS6.copy())
    SNR_dB: float = 20.0                   # T5/Main_model_Task_5.m:106
    dominant_taps: int = 6

    @property
    def T_guard(self):
        return self.Nfft // 8              # T5/Main_model_Task_5.m:8

    @property
    def pilotCarriers(self):
        return np.arange(1, self.N_carrier + 1, self.comb, dtype=np.float64)   # T5:21

    @property
    def dataCarriers(self):
        allc = np.arange(1, self.N_carrier + 1, dtype=np.float64)
        return allc[~np.isin(allc, self.pilotCarriers)]                          # T5:34

    @property
    def K(self):
        return int(np.ceil(self.N_carrier / self.comb))                         # T5:184

    @property
    def frame_samples(self):
        return (self.Nfft + self.T_guard) * self.N_symb


def config_M():
    """Metric config: Nfft=2048, 64-QAM, comb 4 over 512 carriers, OMP(6), frames of 14."""
    return FrameConfig("M", 2048, 512, 4, "64QAM")


def config_C5(seed=5):
    """Nfft=8192, 256-QAM, sparse 32-tap channel (seeded), OMP(32).  The 32 delays are drawn below K = ceil(N_carrier/comb)
    = 512 samples (T5/Main_model_Task_5.m:184: the dictionary the estimator is given has K columns, so a longer echo is
    not representable and only sets an error floor -- round 1 drew them below T_guard = 1024 and its BER(SNR) sweep
    flattened at 0.11); still well inside the guard interval."""
    rng = np.random.default_rng(seed)
    tg = 8192 // 8
    d = np.sort(rng.choice(512, 32, replace=False))
    a = (rng.standard_normal(32) + 1j * rng.standard_normal(32)) / np.sqrt(2) * np.exp(-d / (tg / 4))
    taps = np.stack([d.astype(complex), a], axis=1)
    return FrameConfig("C5", 8192, 2048, 4, "256QAM", taps=taps, dominant_taps=32)


def config_small(nfft=256, n_carrier=64, comb=4, const="16QAM", n_symb=5, taps=None, dominant_taps=3):
    """Tiny configuration for oracle-speed parity tests."""
    if taps is None:
        taps = np.array([[0, 1.0], [3, .6], [7, .3]])
    return FrameConfig("small", nfft, n_carrier, comb, const, N_symb=n_symb, taps=taps, dominant_taps=dominant_taps)


def pilot_column(cfg, api):
    """Pilot values of one symbol: alternating +-2*max|dict| (T5/Task5_part2.m:85-91)."""
    d, _ = api.constellation_func(cfg.Constellation)
    amp = 2 * np.max(np.abs(d))
    n = len(cfg.pilotCarriers)
    return np.where(np.arange(n) % 2 == 0, amp, -amp).astype(np.complex128)


def frame_bits(cfg, api):
    _, bps = api.constellation_func(cfg.Constellation)
    return len(cfg.dataCarriers) * cfg.N_symb * bps


def pack_bits(bits01: np.ndarray) -> np.ndarray:
    """[F, n_bits] 0/1 -> [F, 4*ceil(n_bits/32)] uint8, MSB-first inside each byte (chain layout)."""
    b = np.atleast_2d(np.asarray(bits01, dtype=np.uint8))
    F, n = b.shape
    nb = 4 * ((n + 31) // 32)
    padded = np.zeros((F, nb * 8), dtype=np.uint8)
    padded[:, :n] = b
    return np.packbits(padded, axis=1, bitorder="big")


def unpack_bits(packed: np.ndarray, n_bits: int) -> np.ndarray:
    p = np.atleast_2d(np.asarray(packed, dtype=np.uint8))
    return np.unpackbits(p, axis=1, bitorder="big")[:, :n_bits]


def make_frames(cfg, api, n_frames, seed=1, precision="fp32", device=None, noise=True, frame0=0, noise_first=False):
    """Synthetic RX frames through the product's own TX + channel kernels.

    Returns dict(rx=[frame_samples, n_frames] complex (torch.cuda if device is not None else numpy),
    bits=[n_frames, frame_bits] uint8, packed=[n_frames, frame_bytes] uint8, pilots=[Np] complex).
    Global frame g = frame0 + f draws its payload from PCG64(seed, g) and its noise from Philox key `seed`, stream g
    (each frame is one Noise() call, i.e. the SNR is relative to that frame's measured power, as in
    T5/Task5_part2.m:134) -- so results do not depend on how frames are sharded over GPUs.
    noise_first=True is the order of the reference's drivers, Noise -> conv(h) (T5/Main_model_Task_5.m:106-127,
    T5/Task5_part2.m:134,:152): the SNR is set on the TX signal and the channel colours the noise; the default (False)
    is conv(h) -> Noise, the SNR set at the receiver's input.
    """
    cdt = np.complex128 if precision == "fp64" else np.complex64
    nb = frame_bits(cfg, api)
    bits = np.empty((n_frames, nb), dtype=np.uint8)
    for f in range(n_frames):
        bits[f] = np.random.Generator(np.random.PCG64([seed, frame0 + f])).integers(0, 2, nb, dtype=np.uint8)
    pv_col = pilot_column(cfg, api)
    pv = np.repeat(pv_col[:, None], cfg.N_symb * n_frames, axis=1)
    h, _ = api.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    CH = 4096                                          # frames per Noise_frames call (<= 65535)
    if device is not None:
        import torch
        tb = torch.from_numpy(bits.reshape(-1)).to(device)
        iq, _ = api.mapping(tb, cfg.Constellation, precision=precision)
        X = api.OFDM_map_carriers(iq, cfg.N_symb * n_frames, cfg.Nfft, cfg.dataCarriers, cfg.pilotCarriers,
                                  torch.from_numpy(np.ascontiguousarray(pv.T.astype(cdt))).to(device).t())
        tx = api.OFDM_modulator(X, cfg.T_guard)
        del X, iq
        flat = tx.t().contiguous().view(n_frames, cfg.frame_samples).t()     # [frame_samples, n_frames]
    else:
        iq, _ = api.mapping(bits.reshape(-1), cfg.Constellation, precision=precision)
        X = api.OFDM_map_carriers(iq, cfg.N_symb * n_frames, cfg.Nfft, cfg.dataCarriers, cfg.pilotCarriers,
                                  pv.astype(cdt))
        tx = api.OFDM_modulator(X, cfg.T_guard)
        flat = np.asarray(tx).reshape((cfg.frame_samples, n_frames), order="F")
    def add_noise(rx):
        if n_frames <= CH:
            return api.Noise_frames(cfg.SNR_dB, rx, seed=seed, stream0=frame0)
        parts = []
        for f0 in range(0, n_frames, CH):
            parts.append(api.Noise_frames(cfg.SNR_dB, rx[:, f0:f0 + CH], seed=seed, stream0=frame0 + f0))
        if device is not None:
            import torch
            return torch.cat([p.t() for p in parts], dim=0).t()
        return np.concatenate(parts, axis=1)
    # per-frame stream: the conv transient of a frame falls inside its first CP
    if noise and noise_first:
        rx = api.apply_channel_frames(add_noise(flat), h)
    else:
        rx = api.apply_channel_frames(flat, h)
        if noise:
            rx = add_noise(rx)
    return dict(rx=rx, bits=bits, packed=pack_bits(bits), pilots=pv_col)


def make_plan(cfg, api, precision="fp32", device=None):
    return api.RxPlan(cfg.Nfft, cfg.T_guard, cfg.N_symb, cfg.N_carrier, cfg.pilotCarriers, cfg.dataCarriers,
                      pilot_column(cfg, api), cfg.K, cfg.dominant_taps, cfg.Constellation, precision=precision,
                      device=device)


def make_frames_device(cfg, api, plan, n_frames, seed=1, device=None, noise=True, frame0=0, want_bits=False, **kw):
    """Same role as make_frames, generated entirely on the device by the plan (ofdm_tx_frames): no host payload, no host
    packing -- what the sharded sweeps use.  The payload is the library's Philox draw (not make_frames' PCG64 bits), so the
    two generators give different -- equally valid -- frames."""
    h, _ = api.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    return plan.tx_frames(n_frames, h=h, SNR=cfg.SNR_dB if noise else None, seed=seed, frame0=frame0, device=device,
                          want_bits=want_bits, **kw)
