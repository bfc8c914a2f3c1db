"""Runs the fused chain in MMSE mode at the C4 geometry (Nfft 4096, comb 4, 64QAM, frames of 14): the command profiled by rocprofv3
for the C4 kernels.  usage: python tools/c4_run.py [frames] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ofdm_course_amd as ofdm
import bench_configs
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ofdm.init(0)
print(json.dumps(bench_configs.c4_batched(F, reps)))
