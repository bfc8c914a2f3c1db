"""Per-step durations of the first steps after a synchronisation point (how a short timed region differs from a long one).
usage: python tools/step_profile.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ofdm_course_amd as ofdm
from ofdm_course_amd import frames as fr
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ofdm.init(0)
dev = torch.device("cuda:0")
cfg = fr.config_M()
F = 20480
data = fr.make_frames(cfg, ofdm, F, seed=1, precision="fp32", device=dev)
plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
ref = torch.from_numpy(data["packed"]).to(dev)
step = lambda: ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for _ in range(10): step()
    torch.cuda.synchronize()
for idle_ms in (0, 1, 10, 100):
    torch.cuda.synchronize()
    time.sleep(idle_ms / 1e3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ev[0].record()
    for i in range(K):
        step(); ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
    print("idle %3d ms before: first steps" % idle_ms, [round(x, 3) for x in d[:6]], "mean of rest %.3f" % (sum(d[6:]) / (K - 6)), "total/K %.3f" % (ev[0].elapsed_time(ev[K]) / K))
