"""Per-kernel time of the fast chain as a function of dominant_taps (config M, 8192 frames, fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ofdm_course_amd as ofdm
from ofdm_course_amd import frames as fr

ofdm.init(0)
dev = torch.device("cuda:0")
cfg = fr.config_M()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
data = fr.make_frames(cfg, ofdm, F, seed=1, precision="fp32", device=dev)
ref = torch.from_numpy(data["packed"]).to(dev)
for taps in (1, 2, 3, 4, 6, 8):
    cfg.dominant_taps = taps
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    for _ in range(3):
        ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
    plan.set_timing(True)
    k = []
    for _ in range(8):
        ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
        k.append(plan.last_kernel_ms())
    print(taps, np.round(np.mean(np.array(k), axis=0) * 1e3, 1), flush=True)
