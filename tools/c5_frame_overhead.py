"""C5 symbol stage: per-frame overhead from two frame lengths with the same number of transformed symbols."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
import ofdm_course_amd as ofdm
from ofdm_course_amd import frames as fr
ofdm.init(0)
dev = torch.device("cuda:0")
for F, ns in ((3072, 14), (1536, 27), (6144, 7), (3072, 14)):
    cfg = fr.config_C5()
    cfg.N_symb = ns
    plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
    data = fr.make_frames_device(cfg, ofdm, plan, F, seed=5, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(10):
            ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
        torch.cuda.synchronize()
    plan.set_timing(True)
    k = []
    for _ in range(10):
        ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
        k.append(plan.last_kernel_ms())
    k = np.mean(np.array(k), axis=0)
    print(F, ns, [round(float(x), 4) for x in k], "symbols through the symbol stage:", F * (ns - 1), flush=True)
    plan.close()
