"""The pilot-count Monte-Carlo study of Task 5/Task5_part2.m on the device: wall time of the batched replay (one
ofdm_task5_part2_tile call per scenario) at the reference's size (Nfft 4096, N_carrier 1024, 57 combs x 100 channel draws) and of
the call-by-call replay on a few scenarios; with --random a random-pilot-mask study (dense dictionaries: MP / OMP correlations
on the matrix cores) at Nfft 1024 -- the command profiled for the MFMA counters.
usage: python tools/part2_run.py [--random] [--runs N] [--precision fp32|fp64] [--no-percall]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ofdm_course_amd as ofdm
from ofdm_course_amd.drivers import task5_part2 as drv

ap = argparse.ArgumentParser()
ap.add_argument("--random", action="store_true")
ap.add_argument("--runs", type=int, default=100)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--no-percall", action="store_true")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
ofdm.init(0)
if a.random:
    kw = dict(Nfft=1024, N_carrier=256, reg_pilot=0, Nps=[16, 32, 48, 64, 96, 128], monteCarloRuns=a.runs, SamplingRate=2e7, seed=21)    # seed 21: every mask keeps data carriers (pilot_step != 1)
else:
    kw = dict(monteCarloRuns=a.runs)                       # the committed script: Nfft 4096, N_carrier 1024, 57 combs, EPA
def timed(**k):
    torch.cuda.synchronize(); t = time.perf_counter(); r = drv.run(ofdm, **k); torch.cuda.synchronize()
    return time.perf_counter() - t, r
timed(batched=True, precision=a.precision, **kw)           # plans, operators, clocks
ts = [timed(batched=True, precision=a.precision, **kw) for _ in range(a.reps)]
tb, rb = min(ts, key=lambda x: x[0])
n_real = int(rb["_sums"]["runs"].sum())
out = {"study": "random masks" if a.random else "Task5_part2.m as committed", "precision": a.precision, "scenarios": len(rb["combs"]),
       "realisations": n_real, "batched_s": tb, "realisations_per_s": n_real / tb,
       "BER_first_last": [rb["BERs"][:, 0].tolist(), rb["BERs"][:, -1].tolist()]}
if not a.no_percall:
    sub = dict(kw)
    if a.random: sub["Nps"] = kw["Nps"][:2]
    else: sub["combs"] = [4, 16]
    sub["monteCarloRuns"] = min(a.runs, 10)
    tp, rp = timed(batched=False, **sub)
    out["per_call_s_per_realisation"] = tp / int(rp["_sums"]["runs"].sum())
    out["speedup_vs_per_call"] = out["per_call_s_per_realisation"] * n_real / tb
print(json.dumps(out))
