"""CPU: the oracle's C twin agrees with the numpy oracle (two independent restatements of the
reference's Task-5 RX chain)."""
import os

import numpy as np

from oracle import ofdm_oracle as o
from oracle import ofdm_oracle_c as oc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_c_twin_matches_numpy_oracle_on_the_fixture():
    g = np.load(os.path.join(G, "task5_chain.npz"))
    D, _ = o.constellation_func("16QAM")
    r = oc.rx_chain_task5(g["rx"], int(g["nfft"]), int(g["tg"]), int(g["nc"]), g["pc"], g["dc"], g["pilots"],
                          int(g["nc"]) // 4, int(g["taps"]), D, ref_bits=g["bits_tx"], n_threads=2)
    assert np.array_equal(r["bits"], g["bits_rx"]) and np.array_equal(r["errors"], g["errors"])
    assert np.max(np.abs(r["H"] - g["H"])) < 1e-10
    assert [list(i) for i in r["index"]] == [list(i[i > 0]) for i in g["index"]]


def test_c_twin_config_m_two_frames():
    rng = np.random.default_rng(0)
    nfft, nc, comb, ns, tg = 2048, 512, 4, 14, 256
    pc, dc = o.pilot_layout_comb(nc, comb)
    D, bps = o.constellation_func("64QAM")
    amp = 2 * np.max(np.abs(D))
    pv = np.where(np.arange(len(pc)) % 2 == 0, amp, -amp).astype(complex)
    h, _ = o.get_MP_channel_resp(np.array([[0, 1], [4, .8], [10, .6], [15, .4], [21, .2], [25, .1]]), nfft)
    F = 2
    bits = rng.integers(0, 2, (F, len(dc) * ns * bps)).astype(np.uint8)
    rx = np.zeros(((nfft + tg) * ns, F), complex)
    for f in range(F):
        X = o.OFDM_map_carriers(o.mapping(bits[f], "64QAM")[0], ns, nfft, dc, pc, np.repeat(pv[:, None], ns, axis=1))
        rx[:, f], _ = o.Noise(20.0, o.apply_channel(o.OFDM_modulator(X, tg).ravel(order="F"), h), rng=rng)
    a = o.rx_chain_task5(rx, nfft, tg, nc, pc, dc, pv, 128, 6, "64QAM", ref_bits=bits)
    b = oc.rx_chain_task5(rx, nfft, tg, nc, pc, dc, pv, 128, 6, D, ref_bits=bits, n_threads=2)
    assert np.array_equal(a["bits"], b["bits"]) and np.array_equal(a["errors"], b["errors"])
    assert np.max(np.abs(a["H"] - b["H"])) < 1e-10
    assert [list(i) for i in a["index"]] == b["index"]
