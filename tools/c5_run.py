"""Runs the fused RX chain at the C5 geometry (Nfft 8192, 256-QAM, 32-tap OMP, comb 4) a few times: the command profiled by
tools/pmc_cmd.sh / rocprofv3 for the C5 kernels.  usage: python tools/c5_run.py [frames] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ofdm_course_amd as ofdm
from ofdm_course_amd import frames as fr
F = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ofdm.init(0)
dev = torch.device("cuda:0")
cfg = fr.config_C5()
plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
data = fr.make_frames_device(cfg, ofdm, plan, F, seed=5, device=dev)
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < float(os.environ.get("C5_WARM_S", "0.5")):      # clocks settle after ~0.3 s of back-to-back launches
    for _ in range(10):
        ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
    torch.cuda.synchronize()
plan.set_timing(True)
k = []
for _ in range(reps):
    out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
    k.append(plan.last_kernel_ms())
plan.set_timing(False)
k = np.mean(np.array(k), axis=0)
nsym = F * cfg.N_symb
_, bps = ofdm.constellation_func(cfg.Constellation)
b_sym = (cfg.Nfft + cfg.T_guard) * 8 + 2 * len(cfg.dataCarriers) * bps / 8 + len(cfg.pilotCarriers) * 8 / cfg.N_symb
tot = float(np.sum(k))
print({"frames": F, "kernels_ms": [round(float(x), 4) for x in k], "sum_ms": round(tot, 4), "sym_per_s": nsym / tot * 1e3,
       "hbm_frac": b_sym * nsym / (tot * 1e-3) / 8e12, "errors": int(out["errors"].sum().item())})
