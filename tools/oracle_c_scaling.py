import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ofdm_oracle as o, ofdm_oracle_c as oc
rng = np.random.default_rng(0)
nfft, nc, ns, tg = 2048, 512, 14, 256
pc, dc = o.pilot_layout_comb(nc, 4); D, bps = o.constellation_func("64QAM")
F = 512
rx = np.ascontiguousarray(rng.standard_normal((F, (nfft + tg) * ns)) + 1j * rng.standard_normal((F, (nfft + tg) * ns)))
pv = np.ones(128, complex)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for t in (1, 2, 4, 8, 16, 32):
    r = oc.rx_chain_task5(rx, nfft, tg, nc, pc, dc, pv, 128, 6, D, n_threads=t, frame_major=True)
    print(t, "threads", round(F * ns / r["seconds"]), "sym/s")
