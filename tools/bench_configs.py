"""Secondary measurements of SURVEY.md section 8(d): sym/s of the partial configurations C2..C5 on one MI355X,
inputs resident in HBM (torch CUDA tensors -> device-pointer flavour of the C ABI), HIP-event timing on the
library's stream.  One JSON object per configuration on stdout; `bench.py` (config M) stays the headline.

  C2  Nfft 1024, 16QAM, 100 000 symbols:  OFDM_modulator / Noise / OFDM_demodulator / get_payload+demapping
  C3  Nfft 2048, 64QAM, frame of 50 symbols with STO + CFO + 3-tap multipath: the Task-4 receiver call by call
  C4  Nfft 4096, comb 4, 64QAM, frames of 14: demod -> LS_CE -> MMSE_CE -> equalise -> demap -> BER (per frame)
  C5  Nfft 8192, 256QAM, 32-tap sparse channel: fused RX chain (generic kernel), OMP with 32 taps
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ofdm_course_amd as ofdm
from ofdm_course_amd import frames as fr
from ofdm_course_amd.drivers import common as dc

dev = torch.device("cuda:0")
HBM = 8000.0


def read_once_bytes(nfft, tg, nd, npil, n_symb, bps, C=8):
    """Algorithmic bytes per OFDM symbol of a fused receiver, the M formula of SURVEY.md 8(d) for any geometry: the RX samples
    read ONCE + packed decided bits written + reference bits read + the pilot column amortised over the frame.  Intermediates
    (X, H, the autocorrelation curve, equalised IQ) are not counted -- whatever a receiver moves beyond this is overhead."""
    return (nfft + tg) * C + 2 * nd * bps / 8.0 + npil * C / n_symb


def roofline(b_sym, nsym, ms):
    a = b_sym * nsym / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": a, "peak": HBM, "unit": "GB/s", "frac": a / HBM, "bytes_per_symbol": b_sym,
            "basis": "read-once algorithmic bytes: samples once + bits out + reference bits in + pilots / N_symb"}


def timed(fn, reps=5, warm=2, settle_s=0.3):
    """ms per call over `reps` calls, after `warm` calls and at least `settle_s` seconds of back-to-back calls: a fresh process
    reaches its steady clocks only after ~0.2 s of launches (short cold runs read 10-25 % slow)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


def c2():
    Nfft, Tg, N_carrier, N_symb, const = 1024, 128, 400, 100_000, "16QAM"
    _, pil, dat = dc.layout_percent(Nfft, N_carrier, 25, tail=2)
    d, bps = ofdm.constellation_func(const)
    bits = torch.from_numpy(dc.synthetic_bits(N_symb * len(dat) * bps, 2)).to(dev)
    iq, pad = ofdm.mapping(bits, const, precision="fp32")
    X = ofdm.OFDM_map_carriers(iq, N_symb, Nfft, dat, pil, 2 * float(np.max(np.abs(d))))
    res = {"config": "C2", "Nfft": Nfft, "symbols": N_symb, "dtype": "f32", "kernels": {}}
    C = 8
    ms, tx = timed(lambda: ofdm.OFDM_modulator(X, Tg))
    res["kernels"]["OFDM_modulator"] = {"ms": ms, "sym_per_s": N_symb / ms * 1e3,
                                        "hbm_frac": (2 * Nfft + Tg) * C * N_symb / (ms * 1e-3) / 1e9 / HBM}
    # PAPR study of Task 2 on the same signal: one double per window start, fp32 samples in
    txv = tx.t().reshape(-1)                                    # the stream Tx_OFDM_Signal_matrix(:)
    ms, pw = timed(lambda: ofdm.calculate_window_PAPR(txv, Nfft))
    res["kernels"]["calculate_window_PAPR"] = {"ms": ms, "sym_per_s": N_symb / ms * 1e3, "windows_per_s": pw.numel() / ms * 1e3,
                                               "hbm_frac": (C + 8) * pw.numel() / (ms * 1e-3) / 1e9 / HBM}
    ms, (cx_, cc_) = timed(lambda: ofdm.calculateCCDF(pw), reps=3)
    res["kernels"]["calculateCCDF"] = {"ms": ms, "values_per_s": pw.numel() / ms * 1e3, "distinct": int(cx_.numel()) - 1}
    del pw, cx_, cc_
    ms, rx = timed(lambda: ofdm.Noise(12.0, tx, seed=3)[0])
    res["kernels"]["Noise"] = {"ms": ms, "sym_per_s": N_symb / ms * 1e3,
                               "hbm_frac": 3 * (Nfft + Tg) * C * N_symb / (ms * 1e-3) / 1e9 / HBM}
    ms, Xr = timed(lambda: ofdm.OFDM_demodulator(rx, Tg))
    res["kernels"]["OFDM_demodulator"] = {"ms": ms, "sym_per_s": N_symb / ms * 1e3,
                                          "hbm_frac": 2 * Nfft * C * N_symb / (ms * 1e-3) / 1e9 / HBM}
    ms, out = timed(lambda: ofdm.demapping(pad, ofdm.get_payload(Xr, dat), const))
    res["kernels"]["get_payload+demapping"] = {"ms": ms, "sym_per_s": N_symb / ms * 1e3}
    res["BER"] = float(ofdm.BER_func(bits, out))
    tot = sum(k["ms"] for n, k in res["kernels"].items() if n in ("OFDM_demodulator", "get_payload+demapping"))
    res["rx_sym_per_s"] = N_symb / tot * 1e3
    return res


def c3():
    Nfft, Tg, N_carrier, N_symb, const = 2048, 256, 800, 50, "64QAM"
    allc, pil, dat = dc.layout_percent(Nfft, N_carrier, 15, tail=2)
    d, bps = ofdm.constellation_func(const)
    pv = dc.alternating_pilots(4 / 3 * float(np.max(np.abs(d))), len(pil), N_symb)
    bits = dc.synthetic_bits(N_symb * len(dat) * bps, 3)
    iq, pad = ofdm.mapping(bits, const)
    tx = np.asarray(ofdm.OFDM_modulator(ofdm.OFDM_map_carriers(iq, N_symb, Nfft, dat, pil, pv), Tg)).ravel(order="F")
    # impairment draw on which the reference's (fragile) coarse sync decodes at this size: most draws end in the
    # IFO search picking a leakage line (same in the oracle) -- tests/test_gpu_drivers.py covers parity either way
    rx, _ = ofdm.Noise(30.0, tx, seed=2)
    rx = ofdm.add_CFO(ofdm.add_STO(rx, 1920), 16.1521, Nfft)
    h, _ = ofdm.get_MP_channel_resp(np.array([[0, 1.0], [4, 0.6], [10, 0.3]]), Nfft)      # T4/Main_model_Task_4.m:257-261
    rx = torch.from_numpy(np.asarray(ofdm.apply_channel(rx, h)).astype(np.complex64)).to(dev)
    pvd = torch.from_numpy(np.ascontiguousarray(pv.T.astype(np.complex64))).to(dev).t()

    def receiver():
        _, pos, fo = ofdm.AutoCorrFunction(rx, Tg, Nfft)
        y = ofdm.add_STO(ofdm.add_STO(rx, pos), -(Nfft + Tg))
        y = ofdm.add_CFO(y, -fo, Nfft)
        y, ifo = ofdm.remove_IFO(y, Nfft)
        X = ofdm.OFDM_demodulator(y.view(N_symb, Nfft + Tg).t(), Tg)
        X = ofdm.fine_sync(X, pil, pvd, 1, 1, variant="T4")
        H, _ = ofdm.estimate_channel(X, allc, pil, pvd)
        X = ofdm.equalize_signal(X, H, N_carrier)
        return ofdm.demapping(pad, ofdm.get_payload(X, dat), const), pos, ifo

    t0 = time.perf_counter()
    ms, (out, pos, ifo) = timed(receiver, reps=10)
    res = {"config": "C3", "Nfft": Nfft, "symbols_per_frame": N_symb, "dtype": "f32",
           "ms_per_frame": ms, "sym_per_s": N_symb / ms * 1e3, "TgPosition": int(pos), "IFO": float(ifo),
           "BER": float(ofdm.BER_func(torch.from_numpy(bits).to(dev), out)),
           "note": "per_call = one 50-symbol frame per call sequence (11 launches + 3 host scalars), launch-latency bound; "
                   "batched = ofdm_rx_chain_task4 over 4096 frames with their own STO / CFO draws (1024 frames per call: 0.53 of the HBM peak, 4096: 0.60, 8192: 0.57)"}
    res["batched"] = c3_batched()
    # the O(L) autocorrelation alone on a long stream (HBM-bound kernel)
    long_rx = rx.repeat(200)
    ms2, _ = timed(lambda: ofdm.AutoCorrFunction(long_rx, Tg, Nfft), reps=5)
    nsym = long_rx.numel() / (Nfft + Tg)
    res["AutoCorrFunction_long_stream"] = {"symbols": nsym, "ms": ms2, "sym_per_s": nsym / ms2 * 1e3,
                                           "hbm_frac": 3 * 8 * long_rx.numel() / (ms2 * 1e-3) / 1e9 / HBM}
    return res


def c3_batched(F=4096, reps=5):
    """BASELINE config 3 as a batch: the Task-4 receiver (T4/Main_model_Task_4.m:278-347) over F frames of 50 symbols, every
    frame with its own STO / CFO draw; the frames come from ONE device call (ofdm_tx_frames_ex: Noise -> add_STO -> add_CFO ->
    conv, the reference's order, T4:94-110,:257-267) -- no host loop."""
    Nfft, Tg, N_carrier, N_symb, const = 2048, 256, 800, 50, "64QAM"
    allc, pil, dat = dc.layout_percent(Nfft, N_carrier, 15, tail=2)
    d, bps = ofdm.constellation_func(const)
    pv = dc.alternating_pilots(4 / 3 * float(np.max(np.abs(d))), len(pil), N_symb)
    h, _ = ofdm.get_MP_channel_resp(np.array([[0, 1.0], [4, 0.6], [10, 0.3]]), Nfft)      # T4/Main_model_Task_4.m:257-261
    plan = ofdm.RxPlan(Nfft, Tg, N_symb, N_carrier, pil, dat, pv[:, 0], int(np.ceil(N_carrier / 6)), 3, const,
                       precision="fp32", device=0)
    gen = plan.tx_frames(F, h=h, SNR=30.0, seed=9, device=dev, Time_Delay="random", Freq_Shift="random", noise_first=True)
    rxb = gen["rx"]
    msb, outb = timed(lambda: ofdm.rx_chain_task4(plan, rxb, 1, 1, 1, ref_bits_packed=gen["packed"]), reps=reps, warm=2)
    plan.set_timing(True)
    k = []
    for _ in range(3):
        ofdm.rx_chain_task4(plan, rxb, 1, 1, 1, ref_bits_packed=gen["packed"])
        k.append(plan.last_stage_ms())
    plan.set_timing(False)
    okf = (outb["status"] >= 0)
    nsym = F * N_symb
    r = {"workload": "C3: Nfft=2048 Tg=256 N_carrier=800 64QAM, frames of 50 symbols, 3-tap multipath + per-frame STO/CFO at 30 dB; "
                     "AutoCorrFunction + remove_IFO + fine_sync + estimate_channel (LS + spline) + equalise + demap + BER",
         "frames": F, "ms": msb, "sym_per_s": nsym / msb * 1e3,
         "kernels_ms": {n: float(np.mean([x[n] for x in k])) for n in k[0]},
         "roofline": roofline(read_once_bytes(Nfft, Tg, len(dat), len(pil), N_symb, bps), nsym, msb),
         "frames_with_ifo_line": int(okf.sum().item()),
         "median_frame_BER": float(torch.median(outb["errors"].float() / plan.frame_bits).item())}
    plan.close()
    return r


def c4():
    cfg = fr.FrameConfig("C4", 4096, 1024, 4, "64QAM")
    h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    hh = np.zeros(cfg.N_carrier, dtype=np.complex64)
    hh[: len(h)] = h
    # (a) function by function, one frame per call sequence -- what the reference's Monte-Carlo loop does
    F = 64
    data = fr.make_frames(cfg, ofdm, F, seed=4, precision="fp32", device=dev)
    rx = data["rx"]                                           # [frame_samples, F]
    pvt = torch.from_numpy(np.ascontiguousarray(np.repeat(data["pilots"][:, None], cfg.N_symb, axis=1).T
                                                .astype(np.complex64))).to(dev).t()
    hd = torch.from_numpy(hh).to(dev)
    bits = torch.from_numpy(data["bits"]).to(dev)

    def one(f):
        X = ofdm.OFDM_demodulator(rx[:, f].contiguous().view(cfg.N_symb, cfg.Nfft + cfg.T_guard).t(), cfg.T_guard)
        H = ofdm.MMSE_CE(X, pvt, cfg.pilotCarriers, cfg.Nfft, cfg.N_carrier, hd, cfg.SNR_dB)
        X = ofdm.equalize_signal(X, H, cfg.N_carrier)
        out = ofdm.demapping(0, ofdm.get_payload(X, cfg.dataCarriers), cfg.Constellation)
        return ofdm.BER_func(bits[f], out, return_count=True)

    ms, errs = timed(lambda: [one(f) for f in range(F)], reps=2, warm=1)
    res = {"config": "C4", "Nfft": cfg.Nfft, "dtype": "f32",
           "per_call": {"frames": F, "ms_per_frame": ms / F, "sym_per_s": F * cfg.N_symb / ms * 1e3,
                        "BER": float(sum(errs)) / bits.numel()}}
    del data, rx, bits
    torch.cuda.empty_cache()
    res["batched"] = c4_batched()
    return res


def c4_batched(F=8192, reps=20):
    """BASELINE config 4 as a batch: plan in MMSE mode (MMSE_CE + interpolate for every frame, T5/Task5_part2.m:176-177)."""
    cfg = fr.FrameConfig("C4", 4096, 1024, 4, "64QAM")
    h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    hh = np.zeros(cfg.N_carrier, dtype=np.complex64)
    hh[: len(h)] = h
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    data = fr.make_frames_device(cfg, ofdm, plan, F, seed=4, device=dev, noise_first=True)
    t0 = time.perf_counter()
    plan.set_mmse(hh, cfg.SNR_dB)
    t_plan = time.perf_counter() - t0
    ref = data["packed"]
    ms, out = timed(lambda: ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref), reps=reps, warm=5)
    plan.set_timing(True)
    k = []
    for _ in range(5):
        ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
        k.append(plan.last_kernel_ms())
    k = np.mean(np.array(k), axis=0)
    nsym = F * cfg.N_symb
    _, bps = ofdm.constellation_func(cfg.Constellation)
    flops = 8.0 * cfg.N_carrier * len(cfg.pilotCarriers) * F
    r = {"workload": "C4: Nfft=4096 Tg=512 N_carrier=1024 comb=4 64QAM, frames of 14, MMSE_CE + spline interpolation, 6-tap channel 20 dB",
         "frames": F, "ms": ms, "sym_per_s": nsym / ms * 1e3,
         "kernels_ms": {"rx_pilot_kernel": float(k[0]), "mmse_apply": float(k[1]), "rx_symbols": float(k[2])},
         "roofline": roofline(read_once_bytes(cfg.Nfft, cfg.T_guard, len(cfg.dataCarriers), len(cfg.pilotCarriers), cfg.N_symb, bps),
                              nsym, ms),
         "mmse_gemm_tflops_dense_equivalent": flops / (float(k[1]) * 1e-3) / 1e12, "operator_build_s": t_plan,
         "ber": float(out["errors"].sum().item()) / (F * plan.frame_bits)}
    plan.close()
    return r


def c5_batched(F=3072, reps=5, generic=False):
    """BASELINE config 5: Nfft 8192, 256QAM, 32-tap sparse channel, OMP with 32 taps, the fused chain's split form.
    F = a multiple of 3 x 256 CUs x 4 frames: the 32-tap pursuit keeps three four-frame workgroups per CU resident."""
    cfg = fr.config_C5()
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    data = fr.make_frames_device(cfg, ofdm, plan, F, seed=5, device=dev, noise_first=True)
    ref = data["packed"]
    nsym = F * cfg.N_symb
    _, bps = ofdm.constellation_func(cfg.Constellation)
    b_sym = read_once_bytes(cfg.Nfft, cfg.T_guard, len(cfg.dataCarriers), len(cfg.pilotCarriers), cfg.N_symb, bps)
    ms, out = timed(lambda: ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref), reps=reps, warm=2)
    plan.set_timing(True)
    k = []
    for _ in range(5):
        ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
        k.append(plan.last_kernel_ms())
    plan.set_timing(False)
    k = np.mean(np.array(k), axis=0)
    r = {"workload": "C5: Nfft=8192 Tg=1024 N_carrier=2048 comb=4 256QAM, frames of 14, OMP(32 taps) on a sparse 32-tap channel 20 dB",
         "frames": F, "ms": ms, "sym_per_s": nsym / ms * 1e3,
         "kernels_ms": {"symbol1+pilot_ls": float(k[0]), "omp_batch_kernel": float(k[1]), "symbols": float(k[2])},
         "roofline": roofline(b_sym, nsym, ms),
         "ber": float(out["errors"].sum().item()) / (F * plan.frame_bits)}
    if generic:
        os.environ["OFDM_CHAIN_GENERIC"] = "1"
        try:
            msg, outg = timed(lambda: ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref), reps=2, warm=1)
        finally:
            os.environ.pop("OFDM_CHAIN_GENERIC", None)
        r["generic_single_kernel"] = {"ms": msg, "sym_per_s": nsym / msg * 1e3,
                                      "ber": float(outg["errors"].sum().item()) / (F * plan.frame_bits)}
    plan.close()
    return r


def c5():
    r = c5_batched(generic=True)
    r["config"] = "C5"
    return r


def c2_quick(N_symb=100_000):
    """BASELINE config 2: Nfft 1024, 16QAM, 100 k symbols through OFDM_modulator and OFDM_demodulator (+ payload / demapping)."""
    Nfft, Tg, N_carrier, const = 1024, 128, 400, "16QAM"
    _, pil, dat = dc.layout_percent(Nfft, N_carrier, 25, tail=2)
    d, bps = ofdm.constellation_func(const)
    bits = torch.from_numpy(dc.synthetic_bits(N_symb * len(dat) * bps, 2)).to(dev)
    iq, pad = ofdm.mapping(bits, const, precision="fp32")
    X = ofdm.OFDM_map_carriers(iq, N_symb, Nfft, dat, pil, 2 * float(np.max(np.abs(d))))
    C = 8
    ms_mod, tx = timed(lambda: ofdm.OFDM_modulator(X, Tg), reps=10)
    ms_dem, Xr = timed(lambda: ofdm.OFDM_demodulator(tx, Tg), reps=10)
    ms_bits, out = timed(lambda: ofdm.demapping(pad, ofdm.get_payload(Xr, dat), const), reps=5)
    return {"workload": "C2: Nfft=1024 Tg=128 16QAM, 100k symbols, batched IFFT + CP / strip CP + FFT (per-function entries)",
            "symbols": N_symb, "ms": ms_mod + ms_dem, "sym_per_s": N_symb / (ms_mod + ms_dem) * 1e3,
            "kernels_ms": {"OFDM_modulator": ms_mod, "OFDM_demodulator": ms_dem, "get_payload+demapping": ms_bits},
            "roofline": {"bound": "hbm", "peak": HBM, "unit": "GB/s",
                         "OFDM_modulator": {"bytes_per_symbol": (2 * Nfft + Tg) * C,
                                            "frac": (2 * Nfft + Tg) * C * N_symb / (ms_mod * 1e-3) / 1e9 / HBM},
                         "OFDM_demodulator": {"bytes_per_symbol": 2 * Nfft * C,
                                              "frac": 2 * Nfft * C * N_symb / (ms_dem * 1e-3) / 1e9 / HBM},
                         "achieved": (4 * Nfft + Tg) * C * N_symb / ((ms_mod + ms_dem) * 1e-3) / 1e9,
                         "frac": (4 * Nfft + Tg) * C * N_symb / ((ms_mod + ms_dem) * 1e-3) / 1e9 / HBM,
                         "basis": "X read + guarded signal written (modulator), useful part read + X written (demodulator): each array once"},
            "ber": float(ofdm.BER_func(bits, out))}


def secondary_all():
    """C2..C5 for bench.py's `secondary` key: about 10 s of GPU work, most of it frame generation."""
    ofdm.init(0)
    out = {}
    for name, fn in (("C2", c2_quick), ("C3", c3_batched), ("C4", c4_batched), ("C5", c5_batched)):
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        out[name] = fn()
        out[name]["wall_s"] = time.perf_counter() - t0
    return out


if __name__ == "__main__":
    ofdm.init(0)
    which = sys.argv[1:] or ["C2", "C3", "C4", "C5"]
    for name in which:
        r = {"C2": c2, "C3": c3, "C4": c4, "C5": c5}[name]()
        print(json.dumps(r), flush=True)
