"""Monte-Carlo BER(SNR) sweep of the fused Task-5 RX chain, sharded over the GPUs of one node (SURVEY.md 8e;
BASELINE config 5: Nfft 8192, 256-QAM, sparse 32-tap channel, OMP_estimate, 20 SNR points over 8 GPUs).

The reference's sweeps are loops over independent SNR points (T5/Main_model_Task_5.m:303-346,
T3/Main_model_Task_3.m:237-268).  Here every (snr_idx, batch_idx) tile is one unit: tiles are dealt round-robin
to the ranks (`sweep.tiles_for_rank`), each tile generates its frames on its own GPU (`ofdm_tx_frames`: payload -> TX ->
multipath -> Noise, Philox streams keyed by the tile, so the table does not depend on the GPU count; nothing is drawn
or packed on the host), runs `rx_chain_task5`, and adds
its error / bit counts.  One SUM all-reduce of the int64 counters ends the sweep -- no samples are exchanged.

    python -m ofdm_course_amd.drivers.sweep_ber --config C5 --batches 4 --frames-per-tile 64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m ofdm_course_amd.drivers.sweep_ber --config C5 --batches 16
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np


def run(config="C5", snrs=None, batches=2, frames_per_tile=32, precision="fp32", seed=7, estimator="omp",
        rank=0, world=1, device_index=0, backend="nccl"):
    """Returns (on every rank) the reduced table {"SNRs", "errors", "bits", "BER", ...}."""
    import torch
    import ofdm_course_amd as ofdm
    from ofdm_course_amd import frames as fr
    from ofdm_course_amd import sweep

    ofdm.init(device_index)
    dev = torch.device("cuda", device_index)
    cfg = {"C5": fr.config_C5, "M": fr.config_M}[config]()
    snrs = np.arange(0.0, 30.0, 1.5) if snrs is None else np.asarray(snrs, dtype=float)      # 20 points
    plan = fr.make_plan(cfg, ofdm, precision=precision, device=device_index)
    if estimator == "mmse":
        h, _ = ofdm.get_MP_channel_resp(cfg.taps, cfg.Nfft)
        hh = np.zeros(cfg.N_carrier, dtype=np.complex128)
        hh[: len(h)] = h
    counters = sweep.Counters(len(snrs))
    t0 = time.perf_counter()
    n_tiles = 0
    for si, bi in sweep.tiles_for_rank(len(snrs), batches, rank, world):
        cfg.SNR_dB = float(snrs[si])
        key, stream0 = sweep.tile_seed_stream(seed, si, bi, frames_per_tile)
        data = fr.make_frames_device(cfg, ofdm, plan, frames_per_tile, seed=key, device=dev, frame0=stream0)
        if estimator == "mmse":
            plan.set_mmse(hh, cfg.SNR_dB)
        out = ofdm.rx_chain_task5(plan, data["rx"], ref_bits_packed=data["packed"])
        counters.add(si, 0, int(out["errors"].sum().item()), frames_per_tile * plan.frame_bits)
        n_tiles += 1
    torch.cuda.synchronize()
    local_s = time.perf_counter() - t0
    total = sweep.all_reduce_counters(counters, device=dev if backend == "nccl" else None)
    return {"config": config, "estimator": estimator, "SNRs": snrs.tolist(), "errors": total.errors[:, 0].tolist(),
            "bits": total.bits[:, 0].tolist(), "BER": (total.errors[:, 0] / np.maximum(total.bits[:, 0], 1)).tolist(),
            "batches": batches, "frames_per_tile": frames_per_tile, "n_gpus": world, "tiles_this_rank": n_tiles,
            "seconds_this_rank": local_s, "dtype": "f32" if precision == "fp32" else "f64"}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", choices=["C5", "M"], default="C5")
    ap.add_argument("--batches", type=int, default=2, help="tiles per SNR point")
    ap.add_argument("--frames-per-tile", type=int, default=32)
    ap.add_argument("--snrs", type=float, nargs="*", default=None)
    ap.add_argument("--precision", choices=["fp32", "fp64"], default="fp32")
    ap.add_argument("--estimator", choices=["omp", "mmse"], default="omp")
    ap.add_argument("--backend", default="nccl", help="nccl = RCCL over xGMI; gloo for rehearsals")
    ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: every rank on this GPU")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from ofdm_course_amd import sweep
    rank, local_rank, world = sweep.dist_env()
    dev_index = local_rank if a.force_device is None else a.force_device
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(a.backend)
    res = run(a.config, a.snrs, a.batches, a.frames_per_tile, a.precision, estimator=a.estimator, rank=rank, world=world,
              device_index=dev_index, backend=a.backend)
    if rank == 0:
        text = json.dumps(res)
        if a.json:
            with open(a.json, "w") as f:
                f.write(text)
        else:
            print(text, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
