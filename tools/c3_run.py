"""Runs the batched Task-4 receiver (ofdm_rx_chain_task4) at the C3 geometry (Nfft 2048, 64-QAM, 50-symbol frames, each frame
its own STO / CFO draw + 3-tap multipath) a few times: the command profiled by rocprofv3 for the C3 kernels.
usage: python tools/c3_run.py [frames] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ofdm_course_amd as ofdm
from ofdm_course_amd.drivers import common as dc
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ofdm.init(0)
dev = torch.device("cuda:0")
Nfft, Tg, N_carrier, N_symb, const = 2048, 256, 800, 50, "64QAM"
allc, pil, dat = dc.layout_percent(Nfft, N_carrier, 15, tail=2)
d, bps = ofdm.constellation_func(const)
pv = dc.alternating_pilots(4 / 3 * float(np.max(np.abs(d))), len(pil), N_symb)
h, _ = ofdm.get_MP_channel_resp(np.array([[0, 1.0], [4, 0.6], [10, 0.3]]), Nfft)
rng = np.random.default_rng(3)
plan = ofdm.RxPlan(Nfft, Tg, N_symb, N_carrier, pil, dat, pv[:, 0], int(np.ceil(N_carrier / 6)), 3, const, precision="fp32", device=0)
gen = plan.tx_frames(F, h=None, SNR=None, seed=9, device=dev)
rxb = torch.empty_like(gen["rx"].t().contiguous())
for f in range(F):
    yf = gen["rx"][:, f].contiguous()
    yf, _ = ofdm.Noise(30.0, yf, seed=9, stream=f)
    yf = ofdm.add_CFO(ofdm.add_STO(yf, int(rng.integers(0, Nfft + Tg + 1))), float(rng.integers(0, 31)) + rng.random() - 0.5, Nfft)
    rxb[f] = ofdm.apply_channel(yf, h)
rxb = rxb.t()
run = lambda: ofdm.rx_chain_task4(plan, rxb, 1, 1, 1, ref_bits_packed=gen["packed"])
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for _ in range(10):
        out = run()
    torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    out = run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
print({"frames": F, "ms": round(ms, 4), "sym_per_s": F * N_symb / ms * 1e3,
       "hbm_frac": (Nfft + Tg) * 8 * 3 * F * N_symb / (ms * 1e-3) / 8e12,
       "ok_frames": int((out["status"] >= 0).sum().item()), "errors": int(out["errors"].sum().item())})
