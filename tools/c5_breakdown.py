"""Per-stage times of the split-form chain at the C5 geometry (Nfft 8192, 256-QAM) against the number of OMP taps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ofdm_course_amd as ofdm
from ofdm_course_amd import frames as fr
ofdm.init(0)
dev = torch.device("cuda:0")
for taps in (2, 8, 16, 32):
    cfg = fr.config_C5()
    cfg.dominant_taps = taps
    const = f"taps {taps}"
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    data = fr.make_frames_device(cfg, ofdm, plan, 2048, seed=5, device=dev)
    for _ in range(3): ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
    plan.set_timing(True); k = []
    for _ in range(5):
        ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"]); k.append(plan.last_kernel_ms())
    plan.set_timing(False)
    print(const, np.round(np.mean(np.array(k), axis=0) * 1e3, 1), flush=True)
