"""GPU parity of the fused Task-5 RX chain against the oracle's function-by-function chain."""
import numpy as np
import pytest

from conftest import rel_l2
from flip_audit import decision_flip_audit

pytestmark = pytest.mark.gpu


def _audit_frames(oracle, got_bits, want_bits, iq, const, what):
    """fp32 decisions against the oracle's, frame by frame: every differing decision must be a near-tie of the ORACLE's
    equalised point (flip_audit.py: < 1e-4 level spacings from the boundary).  Returns the number of flipped symbols."""
    flips, worst = 0, 0.0
    for f in range(got_bits.shape[0]):
        n, w = decision_flip_audit(oracle, got_bits[f], want_bits[f], iq[f], const, what=f"{what} frame {f}")
        flips, worst = flips + n, max(worst, w)
    print(f"{what}: {flips} boundary decisions of {iq.size} differ from the oracle's (largest distance {worst:.2e} level spacings)")
    return flips


def _run(ofdm, oracle, cfg, n_frames, precision, seed=1):
    from ofdm_course_amd import frames as fr
    data = fr.make_frames(cfg, ofdm, n_frames, seed=seed, precision=precision)
    plan = fr.make_plan(cfg, ofdm, precision=precision)
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=True)
    ref = oracle.rx_chain_task5(np.asarray(data["rx"]).astype(np.complex128), cfg.Nfft, cfg.T_guard, cfg.N_carrier,
                                cfg.pilotCarriers, cfg.dataCarriers, data["pilots"], cfg.K, cfg.dominant_taps,
                                cfg.Constellation, ref_bits=data["bits"], want_iq=True)
    nb = data["bits"].shape[1]
    got_bits = fr.unpack_bits(np.asarray(out["bits"]), nb)
    return data, out, ref, got_bits


@pytest.mark.parametrize("path", ["fast", "fast_unfused", "generic"])
@pytest.mark.parametrize("precision", ["fp64", "fp32"])
@pytest.mark.parametrize("nfft,nc,const", [(64, 32, "QPSK"), (256, 64, "16QAM"), (512, 100, "8PSK"),
                                            (1024, 256, "64QAM"), (1024, 400, "16QAM"), (2048, 512, "64QAM"),
                                            (4096, 1024, "256QAM")])
def test_chain_matches_oracle(ofdm, oracle, monkeypatch, path, precision, nfft, nc, const):
    """`fast` = wave-local FFT pipeline where supported (Nfft 512..4096): comb pilots whose Nfft/comb divides 512
    run symbol 1 + OMP as one launch with c0 from an inverse transform, `fast_unfused` forces the three-launch
    form (MFMA dictionary correlation) that other layouts use; `generic` = single fused kernel (every Nfft).
    All must reproduce the oracle."""
    from ofdm_course_amd import frames as fr
    if path == "generic":
        monkeypatch.setenv("OFDM_CHAIN_GENERIC", "1")
    else:
        monkeypatch.delenv("OFDM_CHAIN_GENERIC", raising=False)
    if path == "fast_unfused":
        if nfft < 512:
            pytest.skip("fast path starts at Nfft = 512")
        monkeypatch.setenv("OFDM_FAST_UNFUSED", "1")
    else:
        monkeypatch.delenv("OFDM_FAST_UNFUSED", raising=False)
    cfg = fr.config_small(nfft=nfft, n_carrier=nc, comb=4, const=const, n_symb=4 if nfft < 2048 else 14,
                          dominant_taps=3)
    if nfft == 2048:
        cfg = fr.config_M()
    nfr = 5
    data, out, ref, got_bits = _run(ofdm, oracle, cfg, nfr, precision)
    idx = np.asarray(out["index"]).T
    for f in range(nfr):
        want = list(ref["index"][f])
        assert list(idx[f][: len(want)]) == want and not idx[f][len(want):].any()
    tol = 1e-9 if precision == "fp64" else 2e-4
    assert rel_l2(np.asarray(out["H"]).T, ref["H"]) < tol
    if precision == "fp64":
        assert np.array_equal(got_bits, ref["bits"])
        assert np.array_equal(np.asarray(out["errors"]).astype(np.int64), ref["errors"])
    else:
        # fp32: decisions may differ only on points within 1e-4 of a decision boundary -- audited, not counted
        flips = _audit_frames(oracle, got_bits, ref["bits"], ref["iq"], cfg.Constellation, f"{path} Nfft {nfft} {const}")
        assert flips <= 2 * nfr
    # the error counter agrees with the bits this launch produced
    mine = np.count_nonzero(got_bits != data["bits"], axis=1)
    assert np.array_equal(mine, np.asarray(out["errors"]).astype(np.int64))


@pytest.mark.parametrize("const,nc,n_symb,comb,taps_n", [("BPSK", 512, 3, 4, 3), ("QPSK", 300, 2, 4, 2), ("8PSK", 512, 5, 8, 4),
                                                          ("16QAM", 448, 7, 4, 6), ("256QAM", 512, 1, 4, 3), ("256QAM", 384, 4, 16, 8),
                                                          ("64QAM", 200, 9, 2, 5)])
def test_chain_wave_symbol_kernel_variants(ofdm, oracle, monkeypatch, const, nc, n_symb, comb, taps_n):
    """rx_symbols_wave_kernel (Nfft 2048, fp32, N_carrier <= 512) beyond the benchmark instantiation: every slicer order
    (table search for BPSK / QPSK / 8PSK, 2 / 3 / 4 bits per axis), frames of 1, 2 and an odd number of symbols (stash-only
    frame, pack batches that end inside a frame), carrier counts that leave lanes without data, 2..8 taps, ragged batch."""
    from ofdm_course_amd import frames as fr
    monkeypatch.delenv("OFDM_CHAIN_GENERIC", raising=False)
    monkeypatch.delenv("OFDM_FAST_UNFUSED", raising=False)
    monkeypatch.delenv("OFDM_FAST_NO_WAVE", raising=False)
    rng = np.random.default_rng(len(const) + nc + n_symb)
    d = np.sort(rng.choice(min(nc // comb - 1, 255), taps_n, replace=False))
    d[0] = 0
    taps = np.stack([d.astype(float), np.linspace(1.0, 0.35, taps_n)], axis=1)
    cfg = fr.config_small(nfft=2048, n_carrier=nc, comb=comb, const=const, n_symb=n_symb, taps=taps, dominant_taps=taps_n)
    cfg.SNR_dB = 26.0
    nfr = 11
    data, out, ref, got_bits = _run(ofdm, oracle, cfg, nfr, "fp32")
    idx = np.asarray(out["index"]).T
    for f in range(nfr):
        want = list(ref["index"][f])
        assert list(idx[f][: len(want)]) == want and not idx[f][len(want):].any()
    assert rel_l2(np.asarray(out["H"]).T, ref["H"]) < 2e-4
    assert _audit_frames(oracle, got_bits, ref["bits"], ref["iq"], const, f"wave {const} nc {nc}") <= 2 * nfr
    mine = np.count_nonzero(got_bits != data["bits"], axis=1)
    assert np.array_equal(mine, np.asarray(out["errors"]).astype(np.int64))


def test_chain_noiseless_channel(ofdm, oracle):
    """Noiseless 6-tap channel.  OMP's greedy picks are NOT the true delays for this dictionary
    (neighbouring atoms are coherent: the published MSE floor ~3e-3 of T5/graphs/mse(snr), comb1.png),
    so a few bit errors remain -- and they are exactly the oracle's."""
    from ofdm_course_amd import frames as fr
    cfg = fr.config_M()
    data = fr.make_frames(cfg, ofdm, 3, seed=9, precision="fp64", noise=False)
    plan = fr.make_plan(cfg, ofdm, precision="fp64")
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_index=True)
    ref = oracle.rx_chain_task5(data["rx"], cfg.Nfft, cfg.T_guard, cfg.N_carrier, cfg.pilotCarriers,
                                cfg.dataCarriers, data["pilots"], cfg.K, cfg.dominant_taps, cfg.Constellation,
                                ref_bits=data["bits"])
    assert np.array_equal(np.asarray(out["errors"]).astype(np.int64), ref["errors"])
    assert list(np.asarray(out["index"])[:, 0]) == list(ref["index"][0])
    assert np.asarray(out["errors"]).sum() < 0.01 * data["bits"].size


def test_chain_device_flavour_and_batch_independence(ofdm, oracle):
    import torch
    from ofdm_course_amd import frames as fr
    cfg = fr.config_M()
    data = fr.make_frames(cfg, ofdm, 64, seed=3, precision="fp32", device="cuda:0")
    plan = fr.make_plan(cfg, ofdm, precision="fp32")
    ref = torch.from_numpy(data["packed"]).cuda()
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
    torch.cuda.synchronize()
    errs = out["errors"].cpu().numpy()
    # same frames in a different batch position give identical results (sharding independence)
    perm = torch.arange(63, -1, -1, device="cuda:0")
    rx2 = data["rx"].t()[perm].contiguous().t()
    out2 = ofdm.rx_chain_task5(plan, rx2, ref_bits_packed=ref[perm].contiguous())
    torch.cuda.synchronize()
    assert np.array_equal(out2["errors"].cpu().numpy()[::-1], errs)
    ber = errs.sum() / (64 * data["bits"].shape[1])
    assert 0 < ber < 0.2


def test_chain_config_c5_shape(ofdm, oracle):
    """BASELINE config 5 geometry: Nfft 8192, 256-QAM, sparse 32-tap channel, OMP with 32 taps, K = Np = 512
    (split form: Nfft 8192 is outside the 512*{1,2,4,8} wave-local fast path)."""
    from ofdm_course_amd import frames as fr
    cfg = fr.config_C5()
    nfr = 2
    data = fr.make_frames(cfg, ofdm, nfr, seed=5, precision="fp32")
    plan = fr.make_plan(cfg, ofdm, precision="fp32")
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=True)
    ref = oracle.rx_chain_task5(np.asarray(data["rx"]).astype(np.complex128), cfg.Nfft, cfg.T_guard, cfg.N_carrier,
                                cfg.pilotCarriers, cfg.dataCarriers, data["pilots"], cfg.K, cfg.dominant_taps,
                                cfg.Constellation, ref_bits=data["bits"], want_iq=True)
    idx = np.asarray(out["index"]).T
    H = np.asarray(out["H"]).T
    from pick_audit import omp_pick_audit
    Smat = oracle.sensing_matrix(cfg.pilotCarriers, cfg.Nfft, cfg.K)
    pc = np.asarray(cfg.pilotCarriers, int) - 1
    L = cfg.Nfft + cfg.T_guard
    near_total = 0
    for f in range(nfr):
        # a pick may differ from the float64 arg-max only if the top-2 score gap is < 1e-4 of the maximum -- audited
        # pick by pick along the device's own sequence; given its picks, H must be the least-squares refit on them
        got = [int(k) for k in idx[f] if k > 0]
        X1 = oracle.OFDM_demodulator(np.asarray(data["rx"])[:L, f].astype(np.complex128)[:, None], cfg.T_guard)
        Yp = X1[pc, 0] / data["pilots"]
        near, H_refit = omp_pick_audit(oracle, Yp, Smat, got, cfg.Nfft)
        near_total += near
        assert rel_l2(H[f], H_refit[:cfg.N_carrier]) < 2e-4, f
        if near == 0:
            assert got == list(ref["index"][f])[: len(got)] and rel_l2(H[f], ref["H"][f]) < 2e-4
    print(f"C5 fp32: {near_total} near-tied picks of {nfr * cfg.dominant_taps}")
    got_bits = fr.unpack_bits(np.asarray(out["bits"]), data["bits"].shape[1])
    if near_total == 0:
        _audit_frames(oracle, got_bits, ref["bits"], ref["iq"], cfg.Constellation, "C5 fp32")
    # fp64 (parity mode): the split form has no LDS limit at this size; picks and bits must be the oracle's
    plan64 = fr.make_plan(cfg, ofdm, precision="fp64")
    out64 = ofdm.rx_chain_task5(plan64, np.asarray(data["rx"]).astype(np.complex128), ref_bits_packed=data["packed"],
                                want_h=True, want_index=True)
    idx64 = np.asarray(out64["index"]).T
    for f in range(nfr):
        want = list(ref["index"][f])
        assert list(idx64[f][: len(want)]) == want
    assert rel_l2(np.asarray(out64["H"]).T, ref["H"]) < 1e-8
    assert np.array_equal(np.asarray(out64["errors"]).astype(np.int64), ref["errors"])


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
@pytest.mark.parametrize("mode", ["omp", "mmse"])
def test_chain_split_form_8192(ofdm, oracle, precision, mode):
    """Nfft = 8192 runs the split form (demod_keep -> pilot LS -> batch OMP | MMSE operator -> equalise + demap):
    a small frame so the oracle is quick; ragged batch."""
    from ofdm_course_amd import frames as fr
    cfg = fr.config_small(nfft=8192, n_carrier=600, comb=8, const="16QAM", n_symb=3, dominant_taps=3)
    cfg.SNR_dB = 24.0
    nfr = 7
    data = fr.make_frames(cfg, ofdm, nfr, seed=9, precision=precision)
    plan = fr.make_plan(cfg, ofdm, precision=precision)
    tol = 1e-9 if precision == "fp64" else 2e-4
    if mode == "omp":
        out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=True)
        ref = oracle.rx_chain_task5(np.asarray(data["rx"]).astype(np.complex128), cfg.Nfft, cfg.T_guard, cfg.N_carrier,
                                    cfg.pilotCarriers, cfg.dataCarriers, data["pilots"], cfg.K, cfg.dominant_taps,
                                    cfg.Constellation, ref_bits=data["bits"], want_iq=True)
        idx = np.asarray(out["index"]).T
        for f in range(nfr):
            want = list(ref["index"][f])
            assert list(idx[f][: len(want)]) == want and not idx[f][len(want):].any()
        assert rel_l2(np.asarray(out["H"]).T, ref["H"]) < tol
        errs = np.asarray(out["errors"]).astype(np.int64)
        got_bits = fr.unpack_bits(np.asarray(out["bits"]), data["bits"].shape[1])
        if precision == "fp64":
            assert np.array_equal(errs, ref["errors"]) and np.array_equal(got_bits, ref["bits"])
        else:
            _audit_frames(oracle, got_bits, ref["bits"], ref["iq"], cfg.Constellation, "split 8192 omp")
    else:
        h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
        hh = np.zeros(cfg.N_carrier, dtype=np.complex128)
        hh[: len(h)] = h
        plan.set_mmse(hh, cfg.SNR_dB)
        out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True)
        got_bits = fr.unpack_bits(np.asarray(out["bits"]), data["bits"].shape[1])
        pv = np.repeat(data["pilots"][:, None], cfg.N_symb, axis=1)
        bad = 0
        for f in range(nfr):
            rx = np.asarray(data["rx"])[:, f].astype(np.complex128).reshape((cfg.Nfft + cfg.T_guard, cfg.N_symb), order="F")
            X = oracle.OFDM_demodulator(rx, cfg.T_guard)
            Hm = oracle.MMSE_CE(X, pv, cfg.pilotCarriers, cfg.Nfft, cfg.N_carrier, hh, cfg.SNR_dB)
            Hm = Hm[0] if isinstance(Hm, tuple) else Hm
            assert rel_l2(np.asarray(out["H"])[:, f], Hm) < tol
            eq = oracle.equalize_signal(X, Hm, cfg.N_carrier)
            iq = oracle.get_payload(eq, cfg.dataCarriers).ravel(order="F")
            want = np.asarray(oracle.demapping(0, iq, cfg.Constellation)).ravel()
            if precision == "fp64":
                bad += np.count_nonzero(got_bits[f] != want)
            else:
                bad += decision_flip_audit(oracle, got_bits[f], want, iq, cfg.Constellation, what=f"split mmse frame {f}")[0]
        assert bad == 0 if precision == "fp64" else bad <= 2 * nfr


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
@pytest.mark.parametrize("nfft,nc,comb,taps_n,fpw", [(512, 128, 2, 1, 4), (512, 200, 8, 2, 1), (1024, 256, 2, 5, 2),
                                                      (2048, 512, 8, 8, 4), (2048, 384, 16, 4, 8), (4096, 512, 8, 6, 4)])
def test_chain_comb_pilot_stage(ofdm, oracle, monkeypatch, precision, nfft, nc, comb, taps_n, fpw):
    """rx_pilot_omp_kernel over its parameter space: up-sampling factors 512/(Nfft/comb) = 1, 2, 4, 8, every tap
    bucket (2, 4, 6, 8 register-resident picks), 1..8 frames per wavefront, ragged last group (21 frames)."""
    from ofdm_course_amd import frames as fr
    monkeypatch.delenv("OFDM_CHAIN_GENERIC", raising=False)
    monkeypatch.delenv("OFDM_FAST_UNFUSED", raising=False)
    monkeypatch.setenv("OFDM_PILOT_FPW", str(fpw))
    rng = np.random.default_rng(nfft + comb)
    d = np.sort(rng.choice(min(nc // comb - 1, nfft // 8 - 1), taps_n, replace=False))
    d[0] = 0
    taps = np.stack([d.astype(float), np.linspace(1.0, 0.3, taps_n)], axis=1)
    cfg = fr.FrameConfig("comb", nfft, nc, comb, "16QAM", N_symb=3, taps=taps, dominant_taps=taps_n, SNR_dB=25.0)
    nfr = 21
    data, out, ref, got_bits = _run(ofdm, oracle, cfg, nfr, precision, seed=4)
    idx = np.asarray(out["index"]).T
    for f in range(nfr):
        want = list(ref["index"][f])
        assert list(idx[f][: len(want)]) == want and not idx[f][len(want):].any()
    assert rel_l2(np.asarray(out["H"]).T, ref["H"]) < (1e-9 if precision == "fp64" else 2e-4)
    if precision == "fp64":
        assert np.array_equal(np.asarray(out["errors"]).astype(np.int64), ref["errors"])
    else:
        _audit_frames(oracle, got_bits, ref["bits"], ref["iq"], cfg.Constellation, f"comb stage Nfft {nfft}")


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
@pytest.mark.parametrize("nfft,nc,comb,const", [(512, 128, 4, "QPSK"), (1024, 400, 8, "16QAM"), (2048, 512, 4, "64QAM"),
                                                 (4096, 1024, 4, "64QAM")])
def test_chain_mmse_mode(ofdm, oracle, precision, nfft, nc, comb, const):
    """Plan in MMSE mode (ofdm_rx_plan_set_mmse): per frame the oracle runs OFDM_demodulator -> MMSE_CE -> equalize_signal
    -> get_payload -> demapping -> BER_func; the library applies the estimator as one cached operator (GEMM)."""
    from ofdm_course_amd import frames as fr
    cfg = fr.config_small(nfft=nfft, n_carrier=nc, comb=comb, const=const, n_symb=3, dominant_taps=3)
    cfg.SNR_dB = 22.0
    nfr = 37                                                       # ragged against the 32-frame GEMM tile
    data = fr.make_frames(cfg, ofdm, nfr, seed=6, precision=precision)
    plan = fr.make_plan(cfg, ofdm, precision=precision)
    h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    hh = np.zeros(cfg.N_carrier, dtype=np.complex128)
    hh[: len(h)] = h
    plan.set_mmse(hh, cfg.SNR_dB)
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True)
    got_bits = fr.unpack_bits(np.asarray(out["bits"]), data["bits"].shape[1])
    pv = np.repeat(data["pilots"][:, None], cfg.N_symb, axis=1)
    tol = 1e-9 if precision == "fp64" else 2e-4
    bad = 0
    for f in range(nfr):
        rx = np.asarray(data["rx"])[:, f].astype(np.complex128).reshape((cfg.Nfft + cfg.T_guard, cfg.N_symb), order="F")
        X = oracle.OFDM_demodulator(rx, cfg.T_guard)
        Hm = oracle.MMSE_CE(X, pv, cfg.pilotCarriers, cfg.Nfft, cfg.N_carrier, hh, cfg.SNR_dB)
        Hm = Hm[0] if isinstance(Hm, tuple) else Hm
        assert rel_l2(np.asarray(out["H"])[:, f], Hm) < tol
        eq = oracle.equalize_signal(X, Hm, cfg.N_carrier)
        iq = oracle.get_payload(eq, cfg.dataCarriers).ravel(order="F")
        want = np.asarray(oracle.demapping(0, iq, cfg.Constellation)).ravel()
        if precision == "fp64":
            bad += np.count_nonzero(got_bits[f] != want)
        else:
            bad += decision_flip_audit(oracle, got_bits[f], want, iq, cfg.Constellation, what=f"mmse mode frame {f}")[0]
        assert int(np.asarray(out["errors"])[f]) == np.count_nonzero(got_bits[f] != data["bits"][f])
    assert bad == 0 if precision == "fp64" else bad <= 2 * nfr
    # back to OMP mode: the plan behaves as before
    plan.set_mmse(None)
    out2 = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_index=True)
    assert np.asarray(out2["index"]).any()


@pytest.mark.parametrize("nfft,nc,comb,const,nsymb", [(512, 128, 4, "QPSK", 3), (2048, 512, 4, "64QAM", 4), (4096, 1024, 4, "64QAM", 14),
                                                       (4096, 1016, 4, "16QAM", 3), (1024, 256, 2, "16QAM", 2), (2048, 1022, 8, "QPSK", 2)])
def test_chain_mmse_one_launch_equals_two_launches(ofdm, monkeypatch, nfft, nc, comb, const, nsymb):
    """fp32 MMSE mode: mmse_fused_kernel (v = M Y on the matrix cores, then the banded spline from the v tile in LDS) gives the H and
    the bits of the two separate launches bit for bit (same products in the same order); ragged frame count against the 32-frame tile."""
    from ofdm_course_amd import frames as fr
    cfg = fr.config_small(nfft=nfft, n_carrier=nc, comb=comb, const=const, n_symb=nsymb, dominant_taps=3)
    cfg.SNR_dB = 18.0
    nfr = 41
    data = fr.make_frames(cfg, ofdm, nfr, seed=16, precision="fp32")
    plan = fr.make_plan(cfg, ofdm, precision="fp32")
    h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    hh = np.zeros(cfg.N_carrier, dtype=np.complex128)
    hh[: len(h)] = h
    plan.set_mmse(hh, cfg.SNR_dB)
    a = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True)
    monkeypatch.setenv("OFDM_MMSE_TWO_LAUNCHES", "1")
    c = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True)
    monkeypatch.delenv("OFDM_MMSE_TWO_LAUNCHES")
    ha, hc = np.asarray(a["H"]), np.asarray(c["H"])
    assert np.array_equal(ha.real, hc.real) and np.array_equal(ha.imag, hc.imag)
    assert np.array_equal(np.asarray(a["bits"]), np.asarray(c["bits"])) and np.array_equal(np.asarray(a["errors"]), np.asarray(c["errors"]))
    assert np.abs(ha).min() > 0
    plan.close()


def test_chain_mmse_mode_errors(ofdm):
    from ofdm_course_amd import frames as fr
    cfg = fr.config_small(nfft=256, n_carrier=64, comb=4, const="QPSK", n_symb=2)
    plan = fr.make_plan(cfg, ofdm, precision="fp64")
    plan.set_mmse(np.array([1.0, 0.5]), 20.0)
    data = fr.make_frames(cfg, ofdm, 2, seed=1, precision="fp64")
    out = ofdm.rx_chain_task5(plan, data["rx"])                    # Nfft 256: MMSE mode takes the split form
    assert np.asarray(out["bits"]).shape[0] == 2
    with pytest.raises(ofdm.OfdmError):
        plan.set_mmse(np.zeros(4), 20.0)                           # all-zero impulse response


def test_chain_many_taps_fast_path(ofdm, oracle):
    """More than 8 taps on the fast path exercises the LDS-state OMP branch (9..32 taps)."""
    from ofdm_course_amd import frames as fr
    rng = np.random.default_rng(3)
    d = np.sort(rng.choice(100, 12, replace=False))
    taps = np.stack([d.astype(float), rng.uniform(0.2, 1.0, 12)], axis=1)
    cfg = fr.FrameConfig("t12", 2048, 512, 4, "16QAM", N_symb=4, taps=taps, dominant_taps=12)
    nfr = 9
    data = fr.make_frames(cfg, ofdm, nfr, seed=8, precision="fp64")
    plan = fr.make_plan(cfg, ofdm, precision="fp64")
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=True)
    ref = oracle.rx_chain_task5(np.asarray(data["rx"]), cfg.Nfft, cfg.T_guard, cfg.N_carrier, cfg.pilotCarriers,
                                cfg.dataCarriers, data["pilots"], cfg.K, cfg.dominant_taps, cfg.Constellation,
                                ref_bits=data["bits"])
    idx = np.asarray(out["index"]).T
    for f in range(nfr):
        want = list(ref["index"][f])
        assert list(idx[f][: len(want)]) == want and not idx[f][len(want):].any()
    assert rel_l2(np.asarray(out["H"]).T, ref["H"]) < 1e-8
    assert np.array_equal(np.asarray(out["errors"]).astype(np.int64), ref["errors"])


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
@pytest.mark.parametrize("nfft,nc,comb,taps_n", [(1024, 400, 4, 9), (2048, 512, 2, 17), (4096, 1024, 4, 25), (2048, 500, 4, 31)])
def test_chain_long_pursuits(ofdm, oracle, precision, nfft, nc, comb, taps_n):
    """9..32 taps = the one-wavefront-per-frame pursuit of omp_wave_core.hpp (picks in lanes, R = L^-1 over a cleared triangle):
    K = 100, 256, 256, 125 atoms (not multiples of 64), odd tap counts, ragged batch.  fp64: the oracle's picks, H and bits;
    fp32: every pick the arg-max or a near-tie (SURVEY 8c rule, pick_audit.py), H = the refit on the device's picks."""
    from ofdm_course_amd import frames as fr
    from pick_audit import omp_pick_audit
    rng = np.random.default_rng(nfft + taps_n)
    K = int(np.ceil(nc / comb))
    d = np.sort(rng.choice(min(K - 1, nfft // 8 - 1), taps_n, replace=False))
    d[0] = 0
    taps = np.stack([d.astype(float), rng.uniform(0.25, 1.0, taps_n) * np.exp(-d / (nfft / 16))], axis=1)
    cfg = fr.FrameConfig(f"t{taps_n}", nfft, nc, comb, "16QAM", N_symb=3, taps=taps, dominant_taps=taps_n)
    cfg.SNR_dB = 30.0
    nfr = 7
    data = fr.make_frames(cfg, ofdm, nfr, seed=taps_n, precision=precision)
    plan = fr.make_plan(cfg, ofdm, precision=precision)
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=True)
    rx64 = np.asarray(data["rx"]).astype(np.complex128)
    ref = oracle.rx_chain_task5(rx64, cfg.Nfft, cfg.T_guard, cfg.N_carrier, cfg.pilotCarriers, cfg.dataCarriers, data["pilots"],
                                cfg.K, cfg.dominant_taps, cfg.Constellation, ref_bits=data["bits"])
    idx = np.asarray(out["index"]).T
    H = np.asarray(out["H"]).T
    if precision == "fp64":
        for f in range(nfr):
            want = list(ref["index"][f])
            assert list(idx[f][: len(want)]) == want and not idx[f][len(want):].any()
        assert rel_l2(H, ref["H"]) < 1e-8
        assert np.array_equal(np.asarray(out["errors"]).astype(np.int64), ref["errors"])
        return
    Smat = oracle.sensing_matrix(cfg.pilotCarriers, cfg.Nfft, cfg.K)
    pc = np.asarray(cfg.pilotCarriers, int) - 1
    L = cfg.Nfft + cfg.T_guard
    near_total = 0
    for f in range(nfr):
        got = [int(k) for k in idx[f] if k > 0]
        X1 = oracle.OFDM_demodulator(rx64[:L, f][:, None], cfg.T_guard)
        near, H_refit = omp_pick_audit(oracle, X1[pc, 0] / data["pilots"], Smat, got, cfg.Nfft)
        near_total += near
        assert rel_l2(H[f], H_refit[:cfg.N_carrier]) < 3e-4, f
        if near == 0:
            assert got == list(ref["index"][f])[: len(got)]
    print(f"{taps_n} taps fp32: {near_total} near-tied picks of {nfr * taps_n}")


def test_chain_full_size_properties(ofdm):
    """BASELINE size (8192 frames = 114688 symbols of config M): size-independent properties.
    (1) the error counter equals popcount(bits xor reference) recomputed from the packed outputs,
    (2) the result of a frame does not depend on its position in the batch or on the batch size,
    (3) BER is in the band the 20 dB / OMP-floor setting gives (the driver gate is BER < 0.2)."""
    import torch
    from ofdm_course_amd import frames as fr
    cfg = fr.config_M()
    F = 8192
    data = fr.make_frames(cfg, ofdm, F, seed=2, precision="fp32", device="cuda:0")
    plan = fr.make_plan(cfg, ofdm, precision="fp32")
    ref = torch.from_numpy(data["packed"]).cuda()
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
    torch.cuda.synchronize()
    errs = out["errors"].cpu().numpy().astype(np.int64)
    x = (out["bits"] ^ ref).cpu().numpy()
    pop = np.unpackbits(x, axis=1).sum(axis=1)
    assert np.array_equal(pop, errs)
    sub = torch.arange(100, 356, device="cuda:0")
    out2 = ofdm.rx_chain_task5(plan, data["rx"].t()[sub].contiguous().t(), ref_bits_packed=ref[sub].contiguous())
    torch.cuda.synchronize()
    assert np.array_equal(out2["errors"].cpu().numpy().astype(np.int64), errs[100:356])
    assert torch.equal(out2["bits"], out["bits"][100:356])
    ber = errs.sum() / (F * data["bits"].shape[1])
    assert 0.02 < ber < 0.12


def test_chain_plans_interleaved_and_growing_batches(ofdm, oracle):
    """Workspaces are owned by the plan and grow on demand: two plans used alternately with growing / shrinking
    batches (and one of them toggling MMSE mode) give the same per-frame results as a fresh single call."""
    from ofdm_course_amd import frames as fr
    cfg_a = fr.config_M()
    cfg_b = fr.config_small(nfft=1024, n_carrier=256, comb=4, const="16QAM", n_symb=4, dominant_taps=3)
    data_a = fr.make_frames(cfg_a, ofdm, 40, seed=21, precision="fp32")
    data_b = fr.make_frames(cfg_b, ofdm, 40, seed=22, precision="fp32")
    pa, pb = fr.make_plan(cfg_a, ofdm, precision="fp32"), fr.make_plan(cfg_b, ofdm, precision="fp32")
    ref_a = np.asarray(ofdm.rx_chain_task5(fr.make_plan(cfg_a, ofdm, precision="fp32"), data_a["rx"],
                                           ref_bits_packed=data_a["packed"])["errors"])
    ref_b = np.asarray(ofdm.rx_chain_task5(fr.make_plan(cfg_b, ofdm, precision="fp32"), data_b["rx"],
                                           ref_bits_packed=data_b["packed"])["errors"])
    h, _ = ofdm.get_MP_channel_resp(cfg_b.taps, cfg_b.Nfft)
    for n in (3, 17, 40, 5, 33):
        ea = np.asarray(ofdm.rx_chain_task5(pa, np.asarray(data_a["rx"])[:, :n], ref_bits_packed=data_a["packed"][:n])["errors"])
        eb = np.asarray(ofdm.rx_chain_task5(pb, np.asarray(data_b["rx"])[:, :n], ref_bits_packed=data_b["packed"][:n])["errors"])
        assert np.array_equal(ea, ref_a[:n]) and np.array_equal(eb, ref_b[:n])
        pb.set_mmse(np.asarray(h), 20.0)
        em = np.asarray(ofdm.rx_chain_task5(pb, np.asarray(data_b["rx"])[:, :n], ref_bits_packed=data_b["packed"][:n])["errors"])
        assert em.shape == (n,) and em.sum() > 0
        pb.set_mmse(None)
    out0 = ofdm.rx_chain_task5(pa, np.asarray(data_a["rx"])[:, :0])
    assert np.asarray(out0["bits"]).shape[0] == 0


def test_context_is_pinned_to_its_device_while_plans_live(ofdm):
    """One device per process: ofdm_init for another device must not tear the context down under live RX
    plans -- it used to free the twiddle cache and re-aim their launches.  With one visible GPU the other
    device id is out of range (argument error); either way the plan keeps working and the context stays on device 0."""
    from ofdm_course_amd import frames as fr, _lib
    cfg = fr.config_small()
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    lib = _lib.load()
    import torch
    n = torch.cuda.device_count()
    rc = lib.ofdm_init(1)
    assert rc < 0
    msg = lib.ofdm_last_error_string().decode()
    assert ("one device per process" in msg) if n > 1 else ("out of range" in msg), msg
    data = fr.make_frames(cfg, ofdm, 2, seed=3, precision="fp32")
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
    assert np.asarray(out["errors"]).shape == (2,)
    plan.close()
    assert lib.ofdm_init(0) == 0


@pytest.mark.parametrize("slicer", ["arithmetic", "exact"])
def test_metric_config_slicer_flips_are_boundary_points(ofdm, oracle, monkeypatch, slicer):
    """the shipped fp32 slicer of the metric kernel is the arithmetic level rank
    (demap_square_arith, OFDM_WAVE_EXACT_SLICER unset), which departs from demapping.m's first-minimum rule by
    construction.  256 frames of config M (1.38 M decisions) in the reference's channel order: every decision that differs
    from the oracle's is audited against the oracle's equalised IQ -- within 1e-4 level spacings of the boundary or the test
    fails -- and the count is reported.  `exact` = the threshold-count slicer on the same frames."""
    from ofdm_course_amd import frames as fr
    for v in ("OFDM_CHAIN_GENERIC", "OFDM_FAST_UNFUSED", "OFDM_FAST_NO_WAVE", "OFDM_WAVE_EXACT_SLICER"):
        monkeypatch.delenv(v, raising=False)
    if slicer == "exact":
        monkeypatch.setenv("OFDM_WAVE_EXACT_SLICER", "1")
    cfg = fr.config_M()
    nfr = 256
    data = fr.make_frames(cfg, ofdm, nfr, seed=1, precision="fp32", noise_first=True)
    plan = fr.make_plan(cfg, ofdm, precision="fp32")
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_index=True)
    ref = oracle.rx_chain_task5(np.asarray(data["rx"]).astype(np.complex128), cfg.Nfft, cfg.T_guard, cfg.N_carrier,
                                cfg.pilotCarriers, cfg.dataCarriers, data["pilots"], cfg.K, cfg.dominant_taps,
                                cfg.Constellation, ref_bits=data["bits"], want_iq=True)
    idx = np.asarray(out["index"]).T
    same = [f for f in range(nfr) if list(idx[f][: len(ref["index"][f])]) == list(ref["index"][f])]
    assert len(same) >= nfr - 2                                     # a near-tied pick changes H: not this test's subject
    got_bits = fr.unpack_bits(np.asarray(out["bits"]), data["bits"].shape[1])
    flips = _audit_frames(oracle, got_bits[same], ref["bits"][same], ref["iq"][same], cfg.Constellation,
                          f"config M, {slicer} slicer, {len(same)} frames")
    assert flips <= len(same)                                        # a handful in 1.4 M decisions, every one a near-tie


@pytest.mark.parametrize("nc,comb,const,n_symb,taps_n,mode", [(2048, 4, "256QAM", 3, 32, "omp"), (1000, 2, "16QAM", 3, 4, "omp"),
                                                               (1536, 8, "64QAM", 2, 5, "mmse"), (600, 4, "QPSK", 1, 3, "omp")])
def test_chain_one_pass_8192_any_layout(ofdm, monkeypatch, nc, comb, const, n_symb, taps_n, mode):
    """rx_symbols_r2_kernel (Nfft 8192, fp32, N_carrier <= 2048, ANY pilot layout: the frame loop of the symbol stage around the
    eight-wavefront 8192-point transform; taken where rx_symbols_coop4_kernel does not apply, forced here by OFDM_SPLIT_NO_COOP)
    against the split form (transform -> X in HBM -> equalise / demap) on the same frames."""
    from ofdm_course_amd import frames as fr
    for v in ("OFDM_CHAIN_GENERIC", "OFDM_SPLIT_NO_COOP", "OFDM_SPLIT_NO_R2"):
        monkeypatch.delenv(v, raising=False)
    rng = np.random.default_rng(nc + n_symb)
    d = np.sort(rng.choice(min(nc // 4 - 1, 400), taps_n, replace=False))
    d[0] = 0
    taps = np.stack([d.astype(float), np.linspace(1.0, 0.3, taps_n) * np.exp(1j * rng.uniform(0, 6.28, taps_n))], axis=1)
    cfg = fr.FrameConfig("one-pass-r2", 8192, nc, comb, const, N_symb=n_symb, taps=taps, dominant_taps=taps_n, SNR_dB=30.0)
    nfr = 9
    data = fr.make_frames(cfg, ofdm, nfr, seed=13, precision="fp32", noise_first=True)
    plan = fr.make_plan(cfg, ofdm, precision="fp32")
    nb = data["bits"].shape[1]
    if mode == "mmse":
        h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
        hh = np.zeros(cfg.N_carrier, dtype=np.complex128)
        hh[: len(h)] = h
        plan.set_mmse(hh, cfg.SNR_dB)
    monkeypatch.setenv("OFDM_SPLIT_NO_COOP", "1")
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=mode == "omp")
    monkeypatch.setenv("OFDM_SPLIT_NO_R2", "1")
    old = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=mode == "omp")
    got_bits = fr.unpack_bits(np.asarray(out["bits"]), nb)
    old_bits = fr.unpack_bits(np.asarray(old["bits"]), nb)
    assert rel_l2(np.asarray(out["H"]), np.asarray(old["H"])) < 2e-5
    assert np.count_nonzero(got_bits != old_bits) <= 2 * nfr
    assert np.array_equal(np.count_nonzero(got_bits != data["bits"], axis=1), np.asarray(out["errors"]).astype(np.int64))
    if mode == "omp":
        assert np.array_equal(np.asarray(out["index"]), np.asarray(old["index"]))
    plan.close()


@pytest.mark.parametrize("nc,const,n_symb,taps_n,mode", [(2048, "256QAM", 3, 32, "omp"), (1024, "64QAM", 4, 6, "omp"),
                                                          (512, "16QAM", 1, 3, "omp"), (256, "QPSK", 2, 2, "omp"),
                                                          (1024, "16QAM", 5, 3, "mmse")])
def test_chain_one_pass_8192(ofdm, oracle, monkeypatch, nc, const, n_symb, taps_n, mode):
    """rx_symbols_coop4_kernel (Nfft 8192, fp32, comb-4 pilots: radix-4 step across four wavefronts, each running a pruned
    2048-point transform, H by one more transform of the taps, symbol 1 from the stash) against the oracle and against the
    split form (OFDM_SPLIT_NO_COOP) on the same frames: frames of 1 .. 5 symbols, 2 .. 32 taps, a ragged batch, MMSE mode."""
    from ofdm_course_amd import frames as fr
    for v in ("OFDM_CHAIN_GENERIC", "OFDM_SPLIT_NO_COOP", "OFDM_SPLIT_NO_R2"):
        monkeypatch.delenv(v, raising=False)
    rng = np.random.default_rng(nc + n_symb)
    d = np.sort(rng.choice(min(nc // 4 - 1, 400), taps_n, replace=False))
    d[0] = 0
    taps = np.stack([d.astype(float), np.linspace(1.0, 0.3, taps_n) * np.exp(1j * rng.uniform(0, 6.28, taps_n))], axis=1)
    cfg = fr.FrameConfig("one-pass", 8192, nc, 4, const, N_symb=n_symb, taps=taps, dominant_taps=taps_n, SNR_dB=30.0)
    nfr = 7
    data = fr.make_frames(cfg, ofdm, nfr, seed=12, precision="fp32", noise_first=True)
    plan = fr.make_plan(cfg, ofdm, precision="fp32")
    nb = data["bits"].shape[1]
    rx64 = np.asarray(data["rx"]).astype(np.complex128)
    if mode == "mmse":
        h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
        hh = np.zeros(cfg.N_carrier, dtype=np.complex128)
        hh[: len(h)] = h
        plan.set_mmse(hh, cfg.SNR_dB)
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=mode == "omp")
    monkeypatch.setenv("OFDM_SPLIT_NO_COOP", "1")
    monkeypatch.setenv("OFDM_SPLIT_NO_R2", "1")
    old = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"], want_h=True, want_index=mode == "omp")
    monkeypatch.delenv("OFDM_SPLIT_NO_COOP")
    monkeypatch.delenv("OFDM_SPLIT_NO_R2")
    got_bits = fr.unpack_bits(np.asarray(out["bits"]), nb)
    old_bits = fr.unpack_bits(np.asarray(old["bits"]), nb)
    assert rel_l2(np.asarray(out["H"]), np.asarray(old["H"])) < 2e-5
    assert np.count_nonzero(got_bits != old_bits) <= 2 * nfr
    assert np.array_equal(np.count_nonzero(got_bits != data["bits"], axis=1), np.asarray(out["errors"]).astype(np.int64))
    if mode == "omp":
        assert np.array_equal(np.asarray(out["index"]), np.asarray(old["index"]))
        ref = oracle.rx_chain_task5(rx64, cfg.Nfft, cfg.T_guard, cfg.N_carrier, cfg.pilotCarriers, cfg.dataCarriers, data["pilots"],
                                    cfg.K, cfg.dominant_taps, cfg.Constellation, ref_bits=data["bits"], want_iq=True)
        from pick_audit import omp_pick_audit
        Smat = oracle.sensing_matrix(cfg.pilotCarriers, cfg.Nfft, cfg.K)
        pc = np.asarray(cfg.pilotCarriers, int) - 1
        L = cfg.Nfft + cfg.T_guard
        idx, H = np.asarray(out["index"]).T, np.asarray(out["H"]).T
        same = []
        for f in range(nfr):
            got = [int(k) for k in idx[f] if k > 0]
            X1 = oracle.OFDM_demodulator(rx64[:L, f][:, None], cfg.T_guard)
            near, H_refit = omp_pick_audit(oracle, X1[pc, 0] / data["pilots"], Smat, got, cfg.Nfft)
            assert rel_l2(H[f], H_refit[:cfg.N_carrier]) < 2e-4, f
            if near == 0 and got == list(ref["index"][f])[: len(got)]:
                same.append(f)
        assert len(same) >= nfr - 2
        _audit_frames(oracle, got_bits[same], ref["bits"][same], ref["iq"][same], const, f"one-pass 8192 nc {nc}")
