"""ofdm_tx_frames / ofdm_tx_frames_ex (device-side frame generator) against the TX + channel sections of the
reference's drivers composed from the ORACLE's functions (oracle.tx_frame = Scrambler, mapping, OFDM_map_carriers,
OFDM_modulator, Noise on the Philox draws, add_STO, add_CFO, conv) -- VERDICT round 2, item 2: nothing on the `want` side
comes from the library.  The payload / impairment / noise draws are inputs (restated in the oracle); the per-frame
DeScrambler of the fused receivers is tested against the oracle's chain on the same frames."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

REG = (1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0)          # T5/Main_model_Task_5.m:55


def _oracle_frame(oracle, cfg, pv, bps, seed, stream, h, snr, **kw):
    nd = len(cfg.dataCarriers)
    bits = oracle.payload_bits_philox(nd * cfg.N_symb, bps, seed, stream)
    noise = oracle.awgn_philox(cfg.frame_samples, seed, stream) if snr is not None else None
    rx, sc = oracle.tx_frame(bits, cfg.Nfft, cfg.T_guard, cfg.N_symb, cfg.dataCarriers, cfg.pilotCarriers, pv,
                             cfg.Constellation, h=h, SNR=snr, noise=noise, **kw)
    return bits, sc, rx


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
@pytest.mark.parametrize("noise_first", [False, True])
@pytest.mark.parametrize("nfft,nc,comb,const", [(256, 64, 4, "QPSK"), (2048, 512, 4, "64QAM"), (1024, 400, 8, "8PSK")])
def test_tx_frames_equals_the_oracle_composition(ofdm, oracle, precision, noise_first, nfft, nc, comb, const):
    from ofdm_course_amd import frames as fr
    cfg = fr.config_small(nfft=nfft, n_carrier=nc, comb=comb, const=const, n_symb=3, dominant_taps=3)
    if nfft == 2048:
        cfg = fr.config_M()
    plan = fr.make_plan(cfg, ofdm, precision=precision)
    nfr, seed, f0 = 5, 0x1234ABCD5, 7
    h, _ = oracle.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    gen = plan.tx_frames(nfr, h=h, SNR=cfg.SNR_dB, seed=seed, frame0=f0, want_bits=True, noise_first=noise_first)
    _, bps = oracle.constellation_func(cfg.Constellation)
    pv = np.repeat(fr.pilot_column(cfg, ofdm)[:, None], cfg.N_symb, axis=1)
    for f in range(nfr):
        bits, _, want = _oracle_frame(oracle, cfg, pv, bps, seed, f0 + f, h, cfg.SNR_dB, noise_first=noise_first)
        assert np.array_equal(np.asarray(gen["bits"])[f], bits)                                  # payload draw, bit-exact
        assert np.array_equal(np.asarray(gen["packed"])[f], fr.pack_bits(bits[None, :])[0])      # chain layout
        assert rel_l2(np.asarray(gen["rx"])[:, f], want) < (1e-13 if precision == "fp64" else 2e-6)
    # the two orders are different signals (the channel colours the noise when it comes first)
    other = plan.tx_frames(1, h=h, SNR=cfg.SNR_dB, seed=seed, frame0=f0, noise_first=not noise_first)
    assert rel_l2(np.asarray(other["rx"])[:, 0], np.asarray(gen["rx"])[:, 0]) > 1e-3
    # batching independence: frames 2..3 generated alone are the same arrays
    sub = plan.tx_frames(2, h=h, SNR=cfg.SNR_dB, seed=seed, frame0=f0 + 2, noise_first=noise_first)
    assert np.array_equal(np.asarray(sub["rx"]), np.asarray(gen["rx"])[:, 2:4])
    assert np.array_equal(np.asarray(sub["packed"]), np.asarray(gen["packed"])[2:4])
    if not noise_first:                                  # ofdm_tx_frames (the old entry) = the _ex defaults
        import ctypes as C
        from ofdm_course_amd import _lib as L
        rx = np.empty((nfr, cfg.frame_samples), dtype=np.complex128 if precision == "fp64" else np.complex64)
        hh = np.ascontiguousarray(h.astype(rx.dtype))
        L.check(plan.lib.ofdm_tx_frames(plan.handle, hh.ctypes.data_as(C.c_void_p), hh.size, cfg.SNR_dB, 1, seed, f0, nfr,
                                        rx.ctypes.data_as(C.c_void_p), None, None,
                                        L.OFDM_F64 if precision == "fp64" else L.OFDM_F32), "tx_frames")
        assert np.array_equal(rx.T, np.asarray(gen["rx"]))


@pytest.mark.parametrize("precision", ["fp64", "fp32"])
@pytest.mark.parametrize("draw", ["fixed", "random"])
def test_tx_frames_scrambler_sto_cfo_reference_order(ofdm, oracle, precision, draw):
    """Scrambler per frame (register reset, T5/Main_model_Task_5.m:55-69) + Noise -> add_STO -> add_CFO -> conv
    (T4/Main_model_Task_4.m:94-110,:257-267): scrambled payload bit-exact, waveform == the oracle's composition."""
    from ofdm_course_amd import frames as fr
    cfg = fr.config_small(nfft=512, n_carrier=200, comb=5, const="16QAM", n_symb=4, dominant_taps=3)
    plan = fr.make_plan(cfg, ofdm, precision=precision)
    nfr, seed, f0 = 6, 77, 1000
    h, _ = oracle.get_MP_channel_resp(np.array([[0, 1.0], [4, 0.6], [10, 0.3]]), cfg.Nfft)     # T4:257-261
    kw = dict(Time_Delay=37, Freq_Shift=3.3125) if draw == "fixed" else dict(Time_Delay="random", Freq_Shift="random")
    gen = plan.tx_frames(nfr, h=h, SNR=25.0, seed=seed, frame0=f0, want_bits=True, Register=REG, noise_first=True,
                         want_draws=True, **kw)
    _, bps = oracle.constellation_func(cfg.Constellation)
    pv = np.repeat(fr.pilot_column(cfg, ofdm)[:, None], cfg.N_symb, axis=1)
    seen = set()
    for f in range(nfr):
        if draw == "fixed":
            td, fs = 37, 3.3125
        else:
            td, fs = oracle.sto_cfo_draw_philox(cfg.Nfft + cfg.T_guard + 1, seed, f0 + f)
            assert 0 <= td <= cfg.Nfft + cfg.T_guard and -0.5 <= fs < 30.5
            seen.add(td)
        assert int(np.asarray(gen["Time_Delay"])[f]) == td and float(np.asarray(gen["Freq_Shift"])[f]) == fs
        bits, sc, want = _oracle_frame(oracle, cfg, pv, bps, seed, f0 + f, h, 25.0, Register=REG, Time_Delay=td,
                                       Freq_Shift=fs, noise_first=True)
        assert np.array_equal(sc, oracle.Scrambler(REG, bits)[0])                                  # the .m loop itself
        assert np.array_equal(np.asarray(gen["bits"])[f], bits)
        assert np.array_equal(np.asarray(gen["packed"])[f], fr.pack_bits(bits[None, :])[0])
        assert np.array_equal(np.asarray(gen["sc_packed"])[f], fr.pack_bits(sc[None, :])[0])     # scrambled, bit-exact
        assert rel_l2(np.asarray(gen["rx"])[:, f], want) < (1e-12 if precision == "fp64" else 3e-6)
    assert draw == "fixed" or len(seen) > 1
    # add_STO_CFO_frames on its own == add_STO then add_CFO of the oracle, per frame (negative shifts included)
    rng = np.random.default_rng(5)
    y = (rng.standard_normal((300, 4)) + 1j * rng.standard_normal((300, 4))).astype(
        np.complex128 if precision == "fp64" else np.complex64)
    sto, cfo = np.array([0, 17, -23, 299]), np.array([0.0, 12.25, -3.5, 30.49])
    got = np.asarray(ofdm.add_STO_CFO_frames(y, sto, cfo, 64))
    for f in range(4):
        want = oracle.add_CFO(oracle.add_STO(y[:, f].astype(np.complex128), sto[f]), cfo[f], 64)
        assert rel_l2(got[:, f], want) < (1e-12 if precision == "fp64" else 5e-7)   # phases up to ~900 rad in double
    assert np.array_equal(np.asarray(ofdm.add_STO_CFO_frames(y, sto, None, 64))[:, 2], oracle.add_STO(y[:, 2], -23))


@pytest.mark.parametrize("path", ["fast", "generic"])
@pytest.mark.parametrize("precision", ["fp64", "fp32"])
def test_rx_chain_descrambler_equals_oracle(ofdm, oracle, monkeypatch, path, precision):
    """ofdm_rx_plan_set_descrambler: the fused Task-5 RX descrambles every frame's demapped bits (register reset per frame,
    T5/Main_model_Task_5.m:257-274) before bits_out / BER.  fp64: bits and error counts identical to the oracle's chain +
    DeScrambler; fp32: the same up to boundary decisions (each flips up to 3 descrambled bits)."""
    from ofdm_course_amd import frames as fr
    if path == "generic":
        monkeypatch.setenv("OFDM_CHAIN_GENERIC", "1")
    cfg = fr.config_M() if path == "fast" else fr.config_small(nfft=256, n_carrier=64, comb=4, const="16QAM", n_symb=5)
    cfg.N_symb = 14 if path == "fast" else 5
    plan = fr.make_plan(cfg, ofdm, precision=precision)
    h, _ = oracle.get_MP_channel_resp(cfg.taps, cfg.Nfft)
    nfr = 6
    gen = plan.tx_frames(nfr, h=h, SNR=cfg.SNR_dB, seed=11, frame0=3, want_bits=True, Register=REG, noise_first=True)
    rx = np.asarray(gen["rx"])
    want = oracle.rx_chain_task5(rx.astype(np.complex128), cfg.Nfft, cfg.T_guard, cfg.N_carrier, cfg.pilotCarriers,
                                 cfg.dataCarriers, fr.pilot_column(cfg, ofdm), cfg.K, cfg.dominant_taps, cfg.Constellation,
                                 ref_bits=np.asarray(gen["bits"]), Register=REG)
    plan.set_descrambler(REG)
    out = ofdm.rx_chain_task5(plan, rx, ref_bits_packed=np.asarray(gen["packed"]))
    got_bits = fr.unpack_bits(np.asarray(out["bits"]), plan.frame_bits)
    errs = np.asarray(out["errors"]).astype(np.int64)
    assert np.array_equal(errs, np.count_nonzero(got_bits != np.asarray(gen["bits"]), axis=1))   # counter == its own bits
    pad = fr.unpack_bits(np.asarray(out["bits"]), plan.frame_bytes * 8)[:, plan.frame_bits:]
    assert not pad.any()                                                                         # padding stays zero
    if precision == "fp64":
        assert np.array_equal(got_bits, want["bits"]) and np.array_equal(errs, want["errors"])
    else:
        assert np.max(np.count_nonzero(got_bits != want["bits"], axis=1)) <= 6
    # descrambler off: the raw (scrambled) decisions, compared against the scrambled reference
    plan.set_descrambler(None)
    raw = ofdm.rx_chain_task5(plan, rx, ref_bits_packed=np.asarray(gen["sc_packed"]))
    raw_bits = fr.unpack_bits(np.asarray(raw["bits"]), plan.frame_bits)
    for f in range(nfr):
        assert np.array_equal(oracle.DeScrambler(REG, raw_bits[f])[0], got_bits[f])              # the .m loop itself


def test_tx_frames_device_flavour_roundtrip(ofdm):
    """Device flavour, clean channel: the chain decodes its own generator without a single bit error (MMSE mode: on a
    noiseless flat channel the reference's OMP re-picks atom 1 and its pinv split halves the tap, OMP_estimate.m:31-33);
    with the 6-tap channel at 20 dB the OMP chain's BER is the benchmark's.  With Scrambler + DeScrambler in the loop
    (the wave-per-frame kernel's DESCR build, several pack batches per frame) the error count of the same frames becomes
    the descrambled one: every decision error reaches three descrambled bits."""
    import torch
    from ofdm_course_amd import frames as fr
    cfg = fr.config_M()
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    dev = torch.device("cuda:0")
    clean = plan.tx_frames(1500, h=None, SNR=None, seed=3, device=dev)       # more than one 1024-frame chunk
    plan.set_mmse(np.array([1.0]), 60.0)
    out = ofdm.rx_chain_task5(plan, clean["rx"], ref_bits_packed=clean["packed"])
    assert int(out["errors"].sum().item()) == 0
    sc = plan.tx_frames(1500, h=None, SNR=None, seed=3, device=dev, Register=REG)
    plan.set_descrambler(REG)                                                 # MMSE mode + DeScrambler: four-wavefront stage
    out = ofdm.rx_chain_task5(plan, sc["rx"], ref_bits_packed=sc["packed"])
    assert int(out["errors"].sum().item()) == 0
    assert torch.equal(sc["packed"], clean["packed"]) and not torch.equal(sc["sc_packed"], sc["packed"])
    plan.set_descrambler(None)
    plan.set_mmse(None)
    data = fr.make_frames_device(cfg, ofdm, plan, 256, seed=3, device=dev)
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
    ber = out["errors"].sum().item() / (256 * plan.frame_bits)
    assert 0.04 < ber < 0.07
    # scrambled link, OMP mode (the wave kernel's DESCR instantiation): errors vs the scrambled reference without the
    # descrambler; with it, bits == DeScrambler of those raw decisions and the count is the descrambled one
    data = fr.make_frames_device(cfg, ofdm, plan, 256, seed=3, device=dev, Register=REG)
    raw = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["sc_packed"])
    plan.set_descrambler(REG)
    dsc = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
    plan.set_descrambler(None)
    e_raw, e_dsc = int(raw["errors"].sum().item()), int(dsc["errors"].sum().item())
    assert 0.04 < e_raw / (256 * plan.frame_bits) < 0.07 and 2.0 * e_raw < e_dsc <= 3 * e_raw
    rb = fr.unpack_bits(raw["bits"].cpu().numpy(), plan.frame_bits)
    db = fr.unpack_bits(dsc["bits"].cpu().numpy(), plan.frame_bits)
    reg = np.array(REG, dtype=np.uint8)
    for f in (0, 100, 255):
        ext = np.concatenate([reg[::-1], rb[f]])
        n = rb[f].size
        assert np.array_equal(db[f], rb[f] ^ ext[2:2 + n] ^ ext[1:1 + n])    # d[i] = s[i] ^ s[i-13] ^ s[i-14]
