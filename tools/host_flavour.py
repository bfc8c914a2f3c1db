"""PCIe-inclusive throughput of the host-pointer (MEX-style) flavour of the fused chain."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ofdm_course_amd as ofdm
from ofdm_course_amd import frames as fr
cfg = fr.config_M()
F = 1024
data = fr.make_frames(cfg, ofdm, F, seed=1, precision="fp32")
plan = fr.make_plan(cfg, ofdm, precision="fp32")
rx = np.asfortranarray(np.asarray(data["rx"]))
ofdm.rx_chain_task5(plan, rx, ref_bits_packed=data["packed"])
t0 = time.perf_counter()
for _ in range(5):
    out = ofdm.rx_chain_task5(plan, rx, ref_bits_packed=data["packed"])
dt = (time.perf_counter() - t0) / 5
print(f"host-pointer flavour (numpy in/out, staging through PCIe): {F} frames in {dt*1e3:.1f} ms = "
      f"{F*cfg.N_symb/dt/1e6:.2f} M OFDM symbols/s ({rx.nbytes/dt/1e9:.1f} GB/s host->device incl. Python marshalling)")
