"""One parity test per BASELINE.json configuration, each through the
C ABI against the CPU oracle on the same seeded inputs at an oracle-sized slice, plus size-independent
properties at the configuration's full size where the oracle would be too slow."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle_lib import OracleLib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def olib(oracle):
    return OracleLib(oracle)


def test_c1_task1_plumbing(ofdm, olib):
    """configs[0]: Task 1 Main_model.m, Nfft=64, QPSK, AWGN, 1k symbols (200 frames x 5)."""
    from ofdm_course_amd.drivers import task1
    kw = dict(Nfft=64, N_carrier=24, Amount_OFDM_Frames=200, Amount_ODFM_SpF=5, Percent_pilot=25,
              Constellation="QPSK", SNR_dB=6)
    g, o = task1.run(ofdm, **kw), task1.run(olib, **kw)
    assert g["amount_pilots"] == 7 and g["amount_data_carriers"] == 17
    assert np.array_equal(g["_output_bits"], o["_output_bits"]) and g["BER"] == o["BER"] > 0
    clean = task1.run(ofdm, **{**kw, "SNR_dB": None})
    assert clean["passed"]                                                       # T1/Main_model.m:99


def test_c2_batched_ifft_fft(ofdm, oracle):
    """configs[1]: Nfft=1024, 16-QAM, AWGN, 100k-symbol batched IFFT/FFT + CP."""
    import torch
    from ofdm_course_amd.drivers import common as dc
    Nfft, Tg, Nc, const = 1024, 128, 400, "16QAM"
    _, pil, dat = dc.layout_percent(Nfft, Nc, 25, tail=2)
    d, bps = ofdm.constellation_func(const)
    amp = 2 * float(np.max(np.abs(d)))
    # oracle-sized slice, fp64 host flavour: identical bits after mod -> Noise -> demod -> demap
    ns = 500
    bits = dc.synthetic_bits(ns * len(dat) * bps, 21)
    iq, pad = ofdm.mapping(bits, const)
    X = ofdm.OFDM_map_carriers(iq, ns, Nfft, dat, pil, amp)
    tx = ofdm.OFDM_modulator(X, Tg)
    rx, nvar = ofdm.Noise(9.0, np.asarray(tx).ravel(order="F"), seed=99)
    Xr = ofdm.OFDM_demodulator(np.asarray(rx).reshape((Nfft + Tg, ns), order="F"), Tg)
    got = np.asarray(ofdm.demapping(pad, np.asarray(ofdm.get_payload(Xr, dat)).ravel(order="F"), const))
    oX = oracle.OFDM_map_carriers(oracle.mapping(bits, const)[0], ns, Nfft, dat, pil, amp)
    otx = oracle.OFDM_modulator(oX, Tg)
    assert rel_l2(np.asarray(tx), otx) < 1e-13 * 10
    nr, ni = oracle.awgn_philox(otx.size, 99, 0)
    orx, onvar = oracle.Noise(9.0, otx.ravel(order="F"), nr, ni)
    oXr = oracle.OFDM_demodulator(orx.reshape((Nfft + Tg, ns), order="F"), Tg)
    want = oracle.demapping(pad, oracle.get_payload(oXr, dat).ravel(order="F"), const)
    assert abs(nvar - onvar) < 1e-12 * onvar
    assert np.array_equal(got, np.asarray(want).ravel())
    assert 0 < np.count_nonzero(got != bits) < 0.1 * bits.size
    # full size, fp32 device flavour: CP is an exact copy, demod(mod(X)) == X, BER of the clean loop-back is 0
    ns = 100_000
    dev = torch.device("cuda:0")
    tb = torch.from_numpy(dc.synthetic_bits(ns * len(dat) * bps, 22)).to(dev)
    iq, pad = ofdm.mapping(tb, const, precision="fp32")
    X = ofdm.OFDM_map_carriers(iq, ns, Nfft, dat, pil, amp)
    tx = ofdm.OFDM_modulator(X, Tg)
    assert torch.equal(torch.view_as_real(tx[:Tg]), torch.view_as_real(tx[Nfft:]))          # T5/OFDM_modulator.m:8-9
    Xr = ofdm.OFDM_demodulator(tx, Tg)
    err = (torch.linalg.vector_norm(Xr - X) / torch.linalg.vector_norm(X)).item()
    assert err < 1e-6 * 10                                                                   # fp32: 1e-6 * log2(Nfft)
    out = ofdm.demapping(pad, ofdm.get_payload(Xr, dat), const)
    assert ofdm.BER_func(tb, out, return_count=True) == 0


def test_c3_sync_chain(ofdm, olib):
    """configs[2]: Nfft=2048, 64-QAM, multipath + STO/CFO, AutoCorr coarse sync + LS (spline) equalise.
    Impairment draw on which the reference's coarse sync decodes at this size (most draws do not, in the
    oracle as well: the IFO search of remove_IFO.m:5-8 picks up a leakage line)."""
    from ofdm_course_amd.drivers import task4
    kw = dict(Nfft=2048, N_carrier=800, Constellation="64QAM", noise_desync=1, time_desync=1, freq_desync=1,
              mp_desync=1, SNR_dB=30, seed=2)
    g, o = task4.run(ofdm, **kw), task4.run(olib, **kw)
    assert g["TgPosition"] == o["TgPosition"] and g["e_IFO"] == o["e_IFO"] == 16.0
    assert abs(g["FreqOffset"] - o["FreqOffset"]) < 1e-9
    assert o["passed"] and g["passed"]
    assert abs(g["BER"] - o["BER"]) <= 3 / g["_input_bits"].size
    assert abs(g["MER_dB"] - o["MER_dB"]) < 1e-6
    np.testing.assert_allclose(g["_H_est"][:800], o["_H_est"][:800], rtol=1e-9, atol=1e-9)


def test_c4_mmse_chain(ofdm, oracle):
    """configs[3]: Nfft=4096, 64-QAM, comb pilots, MMSE_CE (+ its spline interpolation) -> equalise -> BER,
    frames of 14 symbols (a slice of the 1M-symbol Monte-Carlo: every frame is independent)."""
    from ofdm_course_amd import frames as fr
    cfg = fr.FrameConfig("C4", 4096, 1024, 4, "64QAM")
    nfr = 3
    data = fr.make_frames(cfg, ofdm, nfr, seed=4, precision="fp64")
    pv = np.repeat(data["pilots"][:, None], cfg.N_symb, axis=1)
    h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    hh = np.zeros(cfg.N_carrier, dtype=np.complex128)
    hh[: len(h)] = h
    tot_g = tot_o = 0
    for f in range(nfr):
        rx = np.asarray(data["rx"])[:, f].reshape((cfg.Nfft + cfg.T_guard, cfg.N_symb), order="F")
        X = ofdm.OFDM_demodulator(rx, cfg.T_guard)
        Hl = ofdm.LS_CE(X, pv, cfg.pilotCarriers, cfg.N_carrier)
        Hm = ofdm.MMSE_CE(X, pv, cfg.pilotCarriers, cfg.Nfft, cfg.N_carrier, hh, cfg.SNR_dB)
        eq = ofdm.equalize_signal(X, Hm, cfg.N_carrier)
        bits = np.asarray(ofdm.demapping(0, np.asarray(ofdm.get_payload(eq, cfg.dataCarriers)).ravel(order="F"),
                                         cfg.Constellation))
        oX = oracle.OFDM_demodulator(rx, cfg.T_guard)
        oHl = oracle.LS_CE(oX, pv, cfg.pilotCarriers, cfg.N_carrier)
        oHm = oracle.MMSE_CE(oX, pv, cfg.pilotCarriers, cfg.Nfft, cfg.N_carrier, hh, cfg.SNR_dB)
        oHm = oHm[0] if isinstance(oHm, tuple) else oHm
        oeq = oracle.equalize_signal(oX, oHm, cfg.N_carrier)
        obits = np.asarray(oracle.demapping(0, oracle.get_payload(oeq, cfg.dataCarriers).ravel(order="F"),
                                            cfg.Constellation)).ravel()
        assert rel_l2(np.asarray(Hl), oHl) < 1e-10 and rel_l2(np.asarray(Hm), oHm) < 1e-9
        assert np.array_equal(bits, obits)
        tot_g += np.count_nonzero(bits != data["bits"][f])
        tot_o += np.count_nonzero(obits != data["bits"][f])
    assert tot_g == tot_o and 0 < tot_g < 0.2 * data["bits"].size


def test_c4_full_tile_properties(ofdm, monkeypatch):
    """configs[3] at the benchmark tile (8192 frames of 14 symbols, fp32 plan in MMSE mode, frames generated on the device in the
    reference's channel order): size-independent properties -- the one-launch MMSE stage gives the bits of the two-launch form
    (mmse_fused_kernel / mmse_apply_mfma_kernel + spline_band_kernel: same products, same order), the error counts equal the
    popcount of bits XOR reference, the first 24 frames alone give the same bits (batching independence), BER in the range of
    the operating point."""
    import torch
    from ofdm_course_amd import frames as fr
    cfg = fr.FrameConfig("C4", 4096, 1024, 4, "64QAM")
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    F = 8192
    data = fr.make_frames_device(cfg, ofdm, plan, F, seed=4, device=torch.device("cuda:0"), noise_first=True)
    h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    hh = np.zeros(cfg.N_carrier, dtype=np.complex64)
    hh[: len(h)] = h
    plan.set_mmse(hh, cfg.SNR_dB)
    a = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
    monkeypatch.setenv("OFDM_MMSE_TWO_LAUNCHES", "1")
    b = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
    monkeypatch.delenv("OFDM_MMSE_TWO_LAUNCHES")
    assert torch.equal(a["bits"], b["bits"]) and torch.equal(a["errors"], b["errors"])
    x = (a["bits"] ^ data["packed"]).cpu().numpy()                                     # [F, frame_bytes] uint8
    pop = np.unpackbits(x, axis=1).sum(axis=1)
    assert np.array_equal(pop.astype(np.int64), a["errors"].cpu().numpy().astype(np.int64))
    small = ofdm.rx_chain_task5(plan, data["rx"][:, :24], ref_bits_packed=data["packed"][:24])
    assert torch.equal(small["bits"], a["bits"][:24])
    ber = float(a["errors"].sum().item()) / (F * plan.frame_bits)
    assert 1e-3 < ber < 3e-2
    plan.close()


def test_c3_full_tile_properties(ofdm):
    """configs[2] at the benchmark tile (4096 frames of 50 symbols, every frame with its own STO / CFO draw from ONE
    ofdm_tx_frames_ex call): size-independent properties of the batched Task-4 receiver -- every frame finds its IFO line, the
    recovered TgPosition is the drawn Time_Delay up to the channel's first taps, FreqOffset + IFO give back the drawn
    Freq_Shift, error counts equal the popcount of bits XOR payload, the first 8 frames alone give the same outputs."""
    import torch
    from ofdm_course_amd.drivers import common as dc
    Nfft, Tg, N_carrier, N_symb, const = 2048, 256, 800, 50, "64QAM"
    allc, pil, dat = dc.layout_percent(Nfft, N_carrier, 15, tail=2)
    d, bps = ofdm.constellation_func(const)
    pv = dc.alternating_pilots(4 / 3 * float(np.max(np.abs(d))), len(pil), N_symb)
    h, _ = ofdm.get_MP_channel_resp(np.array([[0, 1.0], [4, 0.6], [10, 0.3]]), Nfft)
    plan = ofdm.RxPlan(Nfft, Tg, N_symb, N_carrier, pil, dat, pv[:, 0], int(np.ceil(N_carrier / 6)), 3, const, precision="fp32", device=0)
    F = 4096
    gen = plan.tx_frames(F, h=h, SNR=30.0, seed=9, device=torch.device("cuda:0"), Time_Delay="random", Freq_Shift="random",
                         noise_first=True, want_draws=True)
    out = ofdm.rx_chain_task4(plan, gen["rx"], 1, 1, 1, ref_bits_packed=gen["packed"])
    st = out["status"].cpu().numpy()
    assert (st >= 0).all()
    tg = out["TgPosition"].cpu().numpy()
    sto = gen["Time_Delay"].cpu().numpy()
    dd = (tg + sto) % (Nfft + Tg)                                    # the coarse estimate lands on the guard-interval plateau of the
    dd = np.minimum(dd, Nfft + Tg - dd)                              # delayed stream (fine_sync removes what is left)
    assert dd.max() <= Tg, dd.max()
    cfo = gen["Freq_Shift"].cpu().numpy()
    est = out["FreqOffset"].cpu().numpy() + out["IFO"].cpu().numpy()
    diff = est - cfo                                                 # the fractional part always; the integer part wherever remove_IFO's
    assert np.abs(diff - np.round(diff)).max() < 0.05                # line search lands on the drawn line.  It lands 1..4 lines low on
    ok_int = np.round(diff) == 0                                     # ~29 % of these draws (never high) and such a frame decodes to BER 0.5:
    assert ok_int.mean() > 0.6, ok_int.mean()                        # the restated remove_IFO.m does the same frame by frame
    assert (np.round(diff)[~ok_int] < 0).all()                       # (test_gpu_task4_batch.py compares them with the oracle)
    x = (out["bits"] ^ gen["packed"]).cpu().numpy()
    assert np.array_equal(np.unpackbits(x, axis=1).sum(axis=1).astype(np.int64), out["errors"].cpu().numpy().astype(np.int64))
    ber = out["errors"].cpu().numpy().astype(np.float64) / plan.frame_bits
    assert np.median(ber[ok_int]) < 0.2 and (ber[~ok_int] > 0.3).all()
    small = ofdm.rx_chain_task4(plan, gen["rx"][:, :8], 1, 1, 1, ref_bits_packed=gen["packed"][:8])
    assert torch.equal(small["bits"], out["bits"][:8]) and torch.equal(small["TgPosition"], out["TgPosition"][:8])
    plan.close()


def test_c5_full_tile_properties(ofdm, monkeypatch):
    """configs[4] at the benchmark tile (3072 frames, Nfft 8192, 256-QAM, OMP with 32 taps): the three forms of the symbol stage
    (rx_symbols_coop4_kernel, rx_symbols_r2_kernel, split form) pick the same atoms and agree on the error count of every frame
    up to the handful of decisions their differently rounded H moves; counts == popcount; batching independence."""
    import torch
    from ofdm_course_amd import frames as fr
    for v in ("OFDM_CHAIN_GENERIC", "OFDM_SPLIT_NO_COOP", "OFDM_SPLIT_NO_R2"):
        monkeypatch.delenv(v, raising=False)
    cfg = fr.config_C5()
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    F = 3072
    data = fr.make_frames_device(cfg, ofdm, plan, F, seed=5, device=torch.device("cuda:0"), noise_first=True)
    a = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_index=True)
    monkeypatch.setenv("OFDM_SPLIT_NO_COOP", "1")
    b = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_index=True)
    monkeypatch.setenv("OFDM_SPLIT_NO_R2", "1")
    c = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_index=True)
    monkeypatch.delenv("OFDM_SPLIT_NO_COOP")
    monkeypatch.delenv("OFDM_SPLIT_NO_R2")
    assert torch.equal(a["index"], b["index"]) and torch.equal(a["index"], c["index"])
    ea, eb, ec = (t["errors"].cpu().numpy().astype(np.int64) for t in (a, b, c))
    assert np.abs(ea - ec).max() <= 8 and np.abs(eb - ec).max() <= 8
    assert abs(int(ea.sum()) - int(ec.sum())) <= 1e-5 * ec.sum() and abs(int(eb.sum()) - int(ec.sum())) <= 1e-5 * ec.sum()
    x = (a["bits"] ^ data["packed"]).cpu().numpy()
    assert np.array_equal(np.unpackbits(x, axis=1).sum(axis=1).astype(np.int64), ea)
    small = ofdm.rx_chain_task5(plan, data["rx"][:, :16], ref_bits_packed=data["packed"][:16])
    assert torch.equal(small["bits"], a["bits"][:16])
    plan.close()


def test_c5_snr_sweep_tiles(ofdm, oracle):
    """configs[4]: Nfft=8192, 256-QAM, sparse 32-tap channel, OMP, SNR sweep dealt as (snr, batch) tiles.
    Tiles of two ranks reproduce the single-rank totals (the sum the RCCL all-reduce forms), and one tile is
    checked against the oracle."""
    from ofdm_course_amd import frames as fr
    from ofdm_course_amd import sweep
    cfg = fr.config_C5()
    snrs = [6.0, 18.0, 30.0]
    fpt = 2                                                      # frames per tile
    plan = fr.make_plan(cfg, ofdm, precision="fp32")

    def run_tiles(tiles):
        errs = np.zeros(len(snrs), dtype=np.int64)
        nbits = np.zeros(len(snrs), dtype=np.int64)
        for (si, bi) in tiles:
            cfg.SNR_dB = snrs[si]
            seed, stream0 = sweep.tile_seed_stream(7, si, bi, fpt)
            data = fr.make_frames(cfg, ofdm, fpt, seed=seed, precision="fp32", frame0=stream0)
            out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
            errs[si] += int(np.asarray(out["errors"]).sum())
            nbits[si] += data["bits"].size
        return errs, nbits

    full = run_tiles(sweep.tiles_for_rank(len(snrs), 2, 0, 1))
    parts = [run_tiles(sweep.tiles_for_rank(len(snrs), 2, r, 2)) for r in range(2)]
    assert np.array_equal(parts[0][0] + parts[1][0], full[0]) and np.array_equal(parts[0][1] + parts[1][1], full[1])
    ber = full[0] / full[1]
    assert ber[0] > ber[2]                                       # BER falls along the sweep
    # one tile against the oracle (fp32 kernel vs fp64 oracle: near-tied picks may differ, the BER may not)
    cfg.SNR_dB = snrs[2]
    seed, stream0 = sweep.tile_seed_stream(7, 2, 0, fpt)
    data = fr.make_frames(cfg, ofdm, fpt, seed=seed, precision="fp32", frame0=stream0)
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=True)
    ref = oracle.rx_chain_task5(np.asarray(data["rx"]).astype(np.complex128), cfg.Nfft, cfg.T_guard, cfg.N_carrier,
                                cfg.pilotCarriers, cfg.dataCarriers, data["pilots"], cfg.K, cfg.dominant_taps,
                                cfg.Constellation, ref_bits=data["bits"], want_iq=True)
    got_bits = fr.unpack_bits(np.asarray(out["bits"]), data["bits"].shape[1])
    from flip_audit import decision_flip_audit
    # every device pick is the float64 arg-max or within 1e-4 of it (tests/pick_audit.py); H is the refit on them
    from pick_audit import omp_pick_audit
    Smat = oracle.sensing_matrix(cfg.pilotCarriers, cfg.Nfft, cfg.K)
    pc = np.asarray(cfg.pilotCarriers, int) - 1
    L = cfg.Nfft + cfg.T_guard
    idx, H = np.asarray(out["index"]).T, np.asarray(out["H"]).T
    near_total, exact_frames = 0, 0
    for f in range(fpt):
        got = [int(k) for k in idx[f] if k > 0]
        X1 = oracle.OFDM_demodulator(np.asarray(data["rx"])[:L, f].astype(np.complex128)[:, None], cfg.T_guard)
        near, H_refit = omp_pick_audit(oracle, X1[pc, 0] / data["pilots"], Smat, got, cfg.Nfft)
        near_total += near
        assert rel_l2(H[f], H_refit[:cfg.N_carrier]) < 2e-4, f
        if near == 0:
            exact_frames += 1
            assert got == list(ref["index"][f])[: len(got)] and rel_l2(H[f], ref["H"][f]) < 2e-4
            # same picks: every decision that differs from the oracle's is a boundary point of the oracle's equalised IQ
            decision_flip_audit(oracle, got_bits[f], ref["bits"][f], ref["iq"][f], cfg.Constellation, what=f"C5 sweep tile frame {f}")
    print(f"C5 sweep tile fp32: {near_total} near-tied picks in {fpt} frames, {exact_frames} frames pick-identical")
