"""C3 demodulator: aligned (STO a multiple of 16 samples = 128 bytes) against random STO -- what the misaligned loads cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
import numpy as np, torch
ofdm_ = bc.ofdm; ofdm_.init(0)
ofdm, dc, dev = bc.ofdm, bc.dc, bc.dev
Nfft, Tg, N_carrier, N_symb, const = 2048, 256, 800, 50, "64QAM"
allc, pil, dat = dc.layout_percent(Nfft, N_carrier, 15, tail=2)
d, bps = ofdm.constellation_func(const)
pv = dc.alternating_pilots(4 / 3 * float(np.max(np.abs(d))), len(pil), N_symb)
h, _ = ofdm.get_MP_channel_resp(np.array([[0, 1.0], [4, 0.6], [10, 0.3]]), Nfft)
plan = ofdm.RxPlan(Nfft, Tg, N_symb, N_carrier, pil, dat, pv[:, 0], int(np.ceil(N_carrier / 6)), 3, const, precision="fp32", device=0)
for td in ("random", 1024, 1031, 1040, "random"):
    gen = plan.tx_frames(4096, h=h, SNR=30.0, seed=9, device=dev, Time_Delay=td, Freq_Shift="random", noise_first=True)
    rxb = gen["rx"]
    for _ in range(3): ofdm.rx_chain_task4(plan, rxb, 1, 1, 1, ref_bits_packed=gen["packed"])
    plan.set_timing(True)
    k = []
    for _ in range(5):
        out = ofdm.rx_chain_task4(plan, rxb, 1, 1, 1, ref_bits_packed=gen["packed"])
        k.append(plan.last_stage_ms())
    plan.set_timing(False)
    tgp = out["tg_position"] if "tg_position" in out else None
    print(td, {n[:10]: round(float(np.mean([x[n] for x in k])), 4) for n in k[0]}, None if tgp is None else tgp[:4].tolist(), flush=True)
