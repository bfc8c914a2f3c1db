"""Per-frame overhead of rx_symbols_kernel: same number of OFDM symbols cut into frames of 4 / 7 / 14 / 28 symbols."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ofdm_course_amd as ofdm
from ofdm_course_amd import frames as fr

ofdm.init(0)
dev = torch.device("cuda:0")
total = 8192 * 14
for ns in (28, 14, 7, 4):
    cfg = fr.config_M()
    cfg.N_symb = ns
    F = total // ns
    data = fr.make_frames(cfg, ofdm, F, seed=1, precision="fp32", device=dev)
    ref = torch.from_numpy(data["packed"]).to(dev)
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    for _ in range(30):
        ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
    plan.set_timing(True)
    k = []
    for _ in range(10):
        ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
        k.append(plan.last_kernel_ms())
    k = np.mean(np.array(k), axis=0) * 1e3
    print(f"N_symb {ns:3d} frames {F:6d}: pilot+omp {k[0]:7.1f} us  symbols {k[2]:7.1f} us  -> {k[2] * 1e3 / F:8.2f} ns/frame, "
          f"{k[2] * 1e3 / (F * (ns - 1)):7.2f} ns per non-first symbol", flush=True)
    del data, ref, plan
    torch.cuda.empty_cache()
