#!/usr/bin/env python3
"""Headline benchmark: OFDM symbols/s through the full Task-5 RX chain (config M: Nfft=2048,
Tg=256, N_carrier=512, comb 4, 64-QAM, OMP(6), frames of 14 symbols) on N MI355X GPUs.

A "step" = one pass of the fused RX chain (ofdm_rx_chain_task5) over this rank's resident batch of
synthetic frames (inputs already in HBM); the K timed steps end with the sweep's one SUM all-reduce of the
error counters (RCCL, only when N > 1).  Weak scaling: every rank owns `--frames` frames; global frame ids (and
therefore payload bits and noise) do not depend on the GPU count.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM, algorithmic
bytes of SURVEY.md section 8d / kernel time from HIP events on the launch stream) and `cpu_baseline`
(the oracle = CPU restatement of the .m reference, timed on this box's host cores on a bounded
sample of the same frames; NOT MATLAB).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_symbol(cfg, bps, csize):
    """RX samples read once + packed decided bits written + reference bits read
    + pilot column amortised over the frame.  Intermediates (X, H, equalised IQ) are not counted."""
    nd, npil = len(cfg.dataCarriers), len(cfg.pilotCarriers)
    return (cfg.Nfft + cfg.T_guard) * csize + 2 * nd * bps / 8.0 + npil * csize / cfg.N_symb


def cpu_baseline(cfg, rx_host, pilots, bits, seconds, n_threads):
    """Oracle chain (C twin of the numpy oracle: oracle/c/ofdm_oracle.c, OpenMP over frames) on the host
    cores of this box.  The same sample of frames is re-run until ~`seconds` CPU-seconds are spent."""
    from oracle import ofdm_oracle as oracle
    from oracle import ofdm_oracle_c as oracle_c
    D, _ = oracle.constellation_func(cfg.Constellation)
    args = (cfg.Nfft, cfg.T_guard, cfg.N_carrier, cfg.pilotCarriers, cfg.dataCarriers, pilots, cfg.K,
            cfg.dominant_taps, D)
    one = oracle_c.rx_chain_task5(rx_host[:64], *args, ref_bits=bits[:64], n_threads=1, frame_major=True)
    single = 64 * cfg.N_symb / one["seconds"]
    wall, passes, r = 0.0, 0, None
    while wall * n_threads < seconds and passes < 50:
        r = oracle_c.rx_chain_task5(rx_host, *args, ref_bits=bits[: rx_host.shape[0]], n_threads=n_threads,
                                    frame_major=True)
        wall += r["seconds"]
        passes += 1
    return dict(value=passes * rx_host.shape[0] * cfg.N_symb / wall, single=single, passes=passes, wall=wall,
                errors=r["errors"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed steps (2000 x 0.45 ms = 0.9 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--frames", type=int, default=20480,
                    help="frames resident per GPU and decoded per step (20480 = 5.3 GB of fp32 input; 8192 ... 24576 measured, the\n                    two launches of a step amortise over more frames: +3 %% from 8192 to 20480, flat beyond)")
    ap.add_argument("--precision", choices=["fp32", "fp64"], default="fp32")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-seconds (threads x wall) for the CPU baseline")
    ap.add_argument("--cpu-frames", type=int, default=2048)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: put every rank on this GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    dev_index = local_rank if args.force_device is None else args.force_device
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where the collective's tensors live

    import ofdm_course_amd as ofdm
    from ofdm_course_amd import frames as fr

    ofdm.init(dev_index)
    cfg = fr.config_M()
    _, bps = ofdm.constellation_func(cfg.Constellation)
    F = args.frames
    f0 = rank * F
    data = fr.make_frames(cfg, ofdm, F, seed=1, precision=args.precision, device=dev, frame0=f0)
    plan = fr.make_plan(cfg, ofdm, precision=args.precision, device=dev_index)
    ref = torch.from_numpy(data["packed"]).to(dev)
    rx = data["rx"]
    torch.cuda.synchronize()
    frame_bits = data["bits"].shape[1]
    def step():
        out = ofdm.rx_chain_task5(plan, rx, ref_bits_packed=ref)
        return out

    def reduce_counters(last_out):
        """The sweep's only collective: ONE SUM all-reduce of the int64 error / bit counters at its end
        (every step decodes the same resident batch, so the last step's counters are the batch's)."""
        c = torch.stack([last_out["errors"].sum().to(torch.int64),
                         torch.tensor(F * frame_bits, dtype=torch.int64, device=dev)]).to(cdev)
        if world > 1:
            dist.all_reduce(c)
        return c

    # one HIP event pair around the K timed steps, on the stream the library launches on (torch's current stream,
    # bound with ofdm_set_stream): a pair per step would put two barrier packets (~11 us) between steps
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    out = None
    for _ in range(args.warmup):
        out = step()
    reduce_counters(out)                       # warm the collective up as well
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        out = step()
    ev[1].record()
    counters = reduce_counters(out)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev[0].elapsed_time(ev[1]) / args.steps if args.steps else float("nan")

    # per-kernel durations (HIP events inside the library, on the launch stream), outside the timed region
    plan.set_timing(True)
    kms = []
    for _ in range(max(3, min(args.steps, 10))):
        step()
        kms.append(plan.last_kernel_ms())
    plan.set_timing(False)
    kms = np.mean(np.array(kms), axis=0)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    tot_err, tot_bits = int(counters[0].item()), int(counters[1].item())

    if rank == 0:
        sym_per_step = F * cfg.N_symb * world
        value = sym_per_step * args.steps / elapsed
        csize = 16 if args.precision == "fp64" else 8
        b_sym = algorithmic_bytes_per_symbol(cfg, bps, csize)
        achieved = b_sym * F * cfg.N_symb / (kernel_ms * 1e-3) / 1e9
        # comb pilots: symbol-1 transform + OMP run as ONE launch (the library reports 0 for the absent OMP launch)
        if float(kms[1]) == 0.0:
            knames, kvals = ["rx_pilot_omp_kernel", "rx_symbols_kernel"], [float(kms[0]), float(kms[2])]
        else:
            knames = ["rx_pilot_kernel", "omp_batch_kernel", "rx_symbols_kernel"]
            kvals = [float(k) for k in kms]
        res = {
            "metric": "OFDM sym/s full Task-5 RX (Nfft=2048, 64-QAM, OMP)",
            "value": value, "unit": "OFDM symbols/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f64", "data": "synthetic",
            "config": {"workload": "M: Nfft=2048 Tg=256 N_carrier=512 comb=4 (128 pilots, K=128) 64QAM OMP(6 taps) "
                                   "6-tap channel 20 dB, frames of 14 symbols",
                       "frames_per_gpu": F, "symbols_per_step": sym_per_step, "sharding": f"frames x{world}"},
            "ber": tot_err / max(tot_bits, 1),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "ofdm_rx_chain_task5 = " + " + ".join(knames) +
                                   " (all launches of one step; dominant: rx_symbols_kernel)",
                         "kernel_ms": kernel_ms, "bytes_per_symbol": b_sym},
            "kernels_ms": dict(zip(knames, kvals)),
        }
        # HBM bytes per step from the committed PMC passes (cannot be collected from inside this process): only
        # quoted when this run is the workload those passes profiled
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1", "traffic.json")
        tj = None
        if os.path.exists(tpath) and args.precision == "fp32":
            with open(tpath) as f:
                tj = json.load(f)
        if tj is not None and tj.get("frames", 8192) == F:
            res["roofline"]["traffic"] = tj["chain_bytes_per_step"]
            res["roofline"]["algorithmic_bytes"] = b_sym * F * cfg.N_symb
            res["roofline"]["traffic_source"] = ("profiles/round1/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                 "passes of this command (tools/pmc.sh), FETCH_SIZE doubled for gfx950")
        # dominant kernel alone: symbols 2..S of every frame + stash + bits out + reference bits in
        nd_, np_ = len(cfg.dataCarriers), len(cfg.pilotCarriers)
        b_dom = ((cfg.N_symb - 1) * (cfg.Nfft + cfg.T_guard) * csize + cfg.N_carrier * csize
                 + 2 * cfg.N_symb * nd_ * bps / 8.0) * F
        res["roofline_dominant"] = {"kernel": "rx_symbols_kernel", "bound": "hbm", "bytes_per_launch": b_dom,
                                    "achieved": b_dom / (float(kms[2]) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": b_dom / (float(kms[2]) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and not args.no_cpu:
            ncpu = min(args.cpu_frames, F)
            nthr = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            rx_host = np.ascontiguousarray(rx.t()[:ncpu].contiguous().cpu().numpy().astype(np.complex128))
            cb = cpu_baseline(cfg, rx_host, data["pilots"], data["bits"], args.cpu_seconds, nthr)
            gpu_errs = out["errors"][:ncpu].cpu().numpy().astype(np.int64)
            res["cpu_baseline"] = {
                "value": cb["value"], "unit": "OFDM symbols/s", "cores": nthr, "kind": "port",
                "single_thread_value": cb["single"],
                "sample": f"first {ncpu} of the {F} benchmark frames ({ncpu * cfg.N_symb} symbols) x {cb['passes']} passes, "
                          f"{cb['wall']:.2f} s wall on {nthr} OpenMP threads; C restatement of the .m reference "
                          f"(oracle/c/ofdm_oracle.c, float64), not MATLAB; host has {os.cpu_count()} cores"}
            res["ber_match"] = {"frames": int(ncpu), "gpu_errors": int(gpu_errs.sum()),
                                "oracle_errors": int(cb["errors"].sum()),
                                "max_abs_diff_per_frame": int(np.max(np.abs(gpu_errs - cb["errors"])))}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
