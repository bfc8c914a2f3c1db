"""How long the part needs to reach its steady rate after an idle period: mean step time in windows of 50 steps.
usage: python tools/ramp_profile.py [idle_seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ofdm_course_amd as ofdm
from ofdm_course_amd import frames as fr
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
ofdm.init(0)
dev = torch.device("cuda:0")
cfg = fr.config_M()
data = fr.make_frames(cfg, ofdm, 20480, seed=1, precision="fp32", device=dev)
plan = fr.make_plan(cfg, ofdm, precision="fp32", device=0)
ref = torch.from_numpy(data["packed"]).to(dev)
step = lambda: ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=ref)
for rep in range(2):
    torch.cuda.synchronize()
    time.sleep(idle)
    out = []
    t_start = time.perf_counter()
    for w in range(60):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): step()
        e1.record(); torch.cuda.synchronize()
        out.append((round(time.perf_counter() - t_start, 2), round(e0.elapsed_time(e1) / 50, 4)))
    print("after %.0f s idle: (t, ms/step) " % idle, out[:12], "...", out[-3:])
