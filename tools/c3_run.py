"""Runs the batched Task-4 receiver (ofdm_rx_chain_task4) at the C3 geometry (Nfft 2048, 64-QAM, 50-symbol frames, each frame
its own STO / CFO draw + 3-tap multipath; frames from ONE ofdm_tx_frames_ex call) a few times: the command profiled by
rocprofv3 for the C3 kernels.  usage: python tools/c3_run.py [frames] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ofdm_course_amd as ofdm
import bench_configs
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ofdm.init(0)
print(json.dumps(bench_configs.c3_batched(F, reps)))
