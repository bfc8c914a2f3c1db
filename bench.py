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


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: this parent never touches the GPU (torch is not even imported
    yet); it starts N fresh children -- one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment exactly as torch.distributed.run would set them -- relays rank 0's single JSON line and exits
    non-zero if any child failed.  The Monte-Carlo loop the ranks share out is the `parfor`-able one of
    Task5_part2.m:146-148."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rcs = [None] * n
    while any(c is None for c in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(c not in (None, 0) for c in rcs):          # a rank died: the others would wait in a collective for ever
            time.sleep(2.0)
            for r, p in enumerate(procs):
                if rcs[r] is None and p.poll() is None:
                    p.kill()                               # exactly the PIDs started above
                    rcs[r] = p.wait()
            break
        time.sleep(0.05)
    rcs = [p.wait() if c is None else c for p, c in zip(procs, rcs)]
    out0.seek(0)
    for line in out0.read().decode().splitlines():      # ONE JSON line on stdout; anything else rank 0 printed goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        sys.exit(1)


def host_threads():
    """CPUs this process may run on (the affinity mask, cut by a cgroup v2 CPU quota if one is set)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def f64_leg(ofdm, fr, cfg, dev, dev_index, F, steps, bps, check_frames):
    """Config M in the reference's own arithmetic (MATLAB computes in double: T5/OFDM_demodulator.m:5-8, T5/OMP_estimate.m:9):
    the same chain call on a float64 plan over F resident float64 frames (the wave-per-frame kernel is fp32 only, so the
    symbol stage is the four-wavefront rx_symbols_kernel<double>).  Roofline on SURVEY 8(d)'s fp64 figure (37 586 B/symbol).
    The frames come from the device generator in the reference's order (Noise -> conv); `check_frames` of them are decoded by
    the oracle's C twin as well (identical error counts are required of parity mode)."""
    import torch
    plan = fr.make_plan(cfg, ofdm, precision="fp64", device=dev_index)
    data = fr.make_frames_device(cfg, ofdm, plan, F, seed=1, device=dev, noise_first=True, want_bits=check_frames > 0)
    rx, ref = data["rx"], data["packed"]
    for _ in range(10):
        out = ofdm.rx_chain_task5(plan, rx, ref_bits_packed=ref)
    torch.cuda.synchronize()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(steps):
        out = ofdm.rx_chain_task5(plan, rx, ref_bits_packed=ref)
    ev[1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev[0].elapsed_time(ev[1]) / steps
    plan.set_timing(True)
    kms = []
    for _ in range(5):
        ofdm.rx_chain_task5(plan, rx, ref_bits_packed=ref)
        kms.append(plan.last_kernel_ms())
    plan.set_timing(False)
    kms = np.mean(np.array(kms), axis=0)
    nsym = F * cfg.N_symb
    b_sym = algorithmic_bytes_per_symbol(cfg, bps, 16)
    achieved = b_sym * nsym / (kernel_ms * 1e-3) / 1e9
    names = ["rx_pilot_omp_kernel<double>", "omp_batch_kernel<double>", "rx_symbols_kernel<double>"]
    r = {"value": nsym * steps / elapsed, "unit": "OFDM symbols/s", "dtype": "f64", "steps": steps, "frames_per_gpu": F,
         "ms_per_step": elapsed / steps * 1e3,
         "kernels_ms": {n: float(k) for n, k in zip(names, kms) if float(k) > 0.0},
         "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                      "kernel_ms": kernel_ms, "bytes_per_symbol": b_sym},
         "ber": float(out["errors"].sum().item()) / (F * plan.frame_bits)}
    if check_frames > 0:
        from oracle import ofdm_oracle as oracle
        from oracle import ofdm_oracle_c as oracle_c
        n = min(check_frames, F)
        D, _ = oracle.constellation_func(cfg.Constellation)
        rx_host = np.ascontiguousarray(rx.t()[:n].contiguous().cpu().numpy())
        want = oracle_c.rx_chain_task5(rx_host, cfg.Nfft, cfg.T_guard, cfg.N_carrier, cfg.pilotCarriers, cfg.dataCarriers,
                                       fr.pilot_column(cfg, ofdm), cfg.K, cfg.dominant_taps, D,
                                       ref_bits=data["bits"][:n].cpu().numpy(), n_threads=max(1, min(host_threads(), n // 4)),
                                       frame_major=True)
        got = out["errors"][:n].cpu().numpy().astype(np.int64)
        r["ber_match"] = {"frames": int(n), "gpu_errors": int(got.sum()), "oracle_errors": int(want["errors"].sum()),
                          "max_abs_diff_per_frame": int(np.max(np.abs(got - want["errors"])))}
    plan.close()
    return r


# Work-skipping ablation switches of rounds 1-2: gone from the shipped library (only the -DOFDM_DIAG build of tools/ has
# OFDM_WAVE_ABL); a benchmark must not be able to skip work by environment, so their mere presence is refused.
RESULT_CHANGING_ENV = ("OFDM_WAVE_ABL", "OFDM_CHAIN_SKIP")


def env_in_force():
    """Every OFDM_* variable set for this run (the library's path-selection switches)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("OFDM_")}


def refuse_result_changing_env():
    bad = [k for k in RESULT_CHANGING_ENV if k in os.environ]
    if bad and not os.environ.get("OFDM_BENCH_ALLOW_DIAG"):
        print(f"bench.py: refusing to run with work-skipping diagnostic variable(s) set: {bad} "
              "(they change results; unset them)", file=sys.stderr)
        sys.exit(3)
    if bad and "_diag" not in os.environ.get("OFDM_LIB_PATH", ""):
        print(f"bench.py: {bad} only exist in libofdm_mi355x_diag.so (OFDM_LIB_PATH)", file=sys.stderr)
        sys.exit(3)


def main():
    refuse_result_changing_env()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed steps (2000 x 0.45 ms = 0.9 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--frames", type=int, default=20480,
                    help="frames resident per GPU and decoded per step (20480 = 5.3 GB of fp32 input; 8192 ... 24576 measured, the\n                    two launches of a step amortise over more frames: +3 %% from 8192 to 20480, flat beyond)")
    ap.add_argument("--precision", choices=["fp32", "fp64"], default="fp32")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-seconds (threads x wall) for the CPU baseline")
    ap.add_argument("--cpu-frames", type=int, default=2048)
    ap.add_argument("--cpu-threads", type=int, default=16, help="second CPU figure next to 1 thread and all cores")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--prewarm-seconds", type=float, default=0.3, help="untimed settling time before the W warm-up steps")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: put every rank on this GPU")
    ap.add_argument("--no-f64", action="store_true", help="skip the reference-precision (float64) leg of config M")
    ap.add_argument("--f64-steps", type=int, default=150)
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2..C5 lines")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus, sys.argv[1:])
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    import torch
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
    # channel stages in the order of the reference's drivers: Noise(20 dB) on the TX signal, then conv(h)
    # (T5/Main_model_Task_5.m:106-127, T5/Task5_part2.m:134,:152) -- rounds 1-2 benchmarked conv -> Noise frames
    data = fr.make_frames(cfg, ofdm, F, seed=1, precision=args.precision, device=dev, frame0=f0, noise_first=True)
    plan = fr.make_plan(cfg, ofdm, precision=args.precision, device=dev_index)
    ref = torch.from_numpy(data["packed"]).to(dev)
    rx = data["rx"]
    torch.cuda.synchronize()
    frame_bits = data["bits"].shape[1]
    def step():
        out = ofdm.rx_chain_task5(plan, rx, ref_bits_packed=ref)
        return out

    counters_dev = torch.zeros(2, dtype=torch.int64, device=dev)
    counters_dev[1] = F * frame_bits

    def reduce_counters(last_out):
        """The sweep's only collective: ONE SUM all-reduce of the int64 error / bit counters at its end
        (every step decodes the same resident batch, so the last step's counters are the batch's).  Two small device
        kernels + the collective: nothing is allocated or copied from the host inside the timed region."""
        counters_dev[0] = last_out["errors"].sum()
        c = counters_dev if cdev == dev else counters_dev.to(cdev)
        if world > 1:
            dist.all_reduce(c)
        return c

    # one HIP event pair around the K timed steps, on the stream the library launches on (torch's current stream,
    # bound with ofdm_set_stream): a pair per step would put two barrier packets (~11 us) between steps
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    out = None
    # clock / TLB settling before the W warm-up steps: a fresh process reaches its steady rate only after ~0.2 s of
    # back-to-back launches (20 timed steps right after 5 warm-up steps read 6-9 % low); untimed, reported as prewarm_steps
    prewarm_steps = 0
    if args.prewarm_seconds > 0:
        # ONE uninterrupted run of launches (a loop that synchronises every few steps keeps the part in its bursty clock
        # state: the first ~20 steps of the timed region then read 10 % slow -- tools/step_profile.py): ten steps to learn
        # the step time, then as many as fill the settling time, nothing waited for until the timed region's own bracket
        t_pre = time.perf_counter()
        for _ in range(10):
            out = step()
        torch.cuda.synchronize()
        per_step = max((time.perf_counter() - t_pre) / 10, 1e-5)
        prewarm_steps = 10 + int(min(args.prewarm_seconds / per_step, 5000))
        for _ in range(prewarm_steps - 10):
            out = step()
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
    # per-frame error counts of the whole job in global frame order (outside the timed region): the digest must not
    # depend on how many ranks shared the frames out
    ferr = out["errors"].to(torch.int32).to(cdev).contiguous()
    if world > 1:
        parts = [torch.empty_like(ferr) for _ in range(world)]
        dist.all_gather(parts, ferr)
        ferr = torch.cat(parts)
    ferr = ferr.cpu().numpy()

    if rank == 0:
        sym_per_step = F * cfg.N_symb * world
        value = sym_per_step * args.steps / elapsed
        csize = 16 if args.precision == "fp64" else 8
        b_sym = algorithmic_bytes_per_symbol(cfg, bps, csize)
        achieved = b_sym * F * cfg.N_symb / (kernel_ms * 1e-3) / 1e9
        # comb pilots: symbol-1 transform + OMP run as ONE launch (the library reports 0 for the absent OMP launch)
        # the symbol stage of this geometry (Nfft 2048, fp32) is the one-wavefront-per-frame kernel unless switched off
        sym_kernel = ("rx_symbols_wave_kernel" if args.precision == "fp32" and not os.environ.get("OFDM_FAST_NO_WAVE")
                      else "rx_symbols_kernel")
        if float(kms[1]) == 0.0:
            knames, kvals = ["rx_pilot_omp_kernel", sym_kernel], [float(kms[0]), float(kms[2])]
        else:
            knames = ["rx_pilot_kernel", "omp_batch_kernel", sym_kernel]
            kvals = [float(k) for k in kms]
        res = {
            "metric": "OFDM sym/s full Task-5 RX (Nfft=2048, 64-QAM, OMP)",
            "value": value, "unit": "OFDM symbols/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "prewarm_steps": prewarm_steps,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f64", "data": "synthetic",
            "config": {"workload": "M: Nfft=2048 Tg=256 N_carrier=512 comb=4 (128 pilots, K=128) 64QAM OMP(6 taps) "
                                   "Noise(20 dB) -> 6-tap channel (the reference's order), frames of 14 symbols",
                       "frames_per_gpu": F, "symbols_per_step": sym_per_step, "sharding": f"frames x{world}"},
            "ber": tot_err / max(tot_bits, 1),
            "frame_errors": {"frames": int(ferr.size), "sum": int(ferr.sum()),
                             "sha1": __import__("hashlib").sha1(ferr.astype("<i4").tobytes()).hexdigest()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "ofdm_rx_chain_task5 = " + " + ".join(knames) +
                                   f" (all launches of one step; dominant: {sym_kernel})",
                         "kernel_ms": kernel_ms, "bytes_per_symbol": b_sym},
            "kernels_ms": dict(zip(knames, kvals)),
            "env": env_in_force(),
        }
        # HBM bytes per step from the committed PMC passes (cannot be collected from inside this process): only
        # quoted when this run is the workload those passes profiled
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        rounds = sorted(d for d in os.listdir(pdir) if os.path.exists(os.path.join(pdir, d, "traffic.json")))
        tpath = os.path.join(pdir, rounds[-1], "traffic.json") if rounds else ""
        tj = None
        if tpath and args.precision == "fp32":
            with open(tpath) as f:
                tj = json.load(f)
        if tj is not None and tj.get("frames", 8192) == F:
            res["roofline"]["traffic"] = tj["chain_bytes_per_step"]
            res["roofline"]["algorithmic_bytes"] = b_sym * F * cfg.N_symb
            res["roofline"]["traffic_source"] = (os.path.relpath(tpath, os.path.dirname(pdir)) + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                 "passes of this command (tools/pmc.sh), FETCH_SIZE doubled for gfx950")
        # dominant kernel alone: symbols 2..S of every frame + stash + bits out + reference bits in
        nd_, np_ = len(cfg.dataCarriers), len(cfg.pilotCarriers)
        b_dom = ((cfg.N_symb - 1) * (cfg.Nfft + cfg.T_guard) * csize + cfg.N_carrier * csize
                 + 2 * cfg.N_symb * nd_ * bps / 8.0) * F
        res["roofline_dominant"] = {"kernel": sym_kernel, "bound": "hbm", "bytes_per_launch": b_dom,
                                    "achieved": b_dom / (float(kms[2]) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": b_dom / (float(kms[2]) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and not args.no_cpu:
            ncpu = min(args.cpu_frames, F)
            nall = max(1, min(host_threads(), ncpu // 4))
            nthr = max(1, min(args.cpu_threads, nall))
            rx_host = np.ascontiguousarray(rx.t()[:ncpu].contiguous().cpu().numpy().astype(np.complex128))
            cb = cpu_baseline(cfg, rx_host, data["pilots"], data["bits"], args.cpu_seconds, nall)
            cb16 = cb if nthr == nall else cpu_baseline(cfg, rx_host, data["pilots"], data["bits"],
                                                        args.cpu_seconds / 2, nthr)
            gpu_errs = out["errors"][:ncpu].cpu().numpy().astype(np.int64)
            res["cpu_baseline"] = {
                "value": cb["value"], "unit": "OFDM symbols/s", "cores": nall, "kind": "port",
                "single_thread_value": cb["single"],
                "threads": {"1": cb["single"], str(nthr): cb16["value"], str(nall): cb["value"]},
                "host_cpu_count": os.cpu_count(), "usable_cpus": host_threads(),
                "sample": f"first {ncpu} of the {F} benchmark frames ({ncpu * cfg.N_symb} symbols) x {cb['passes']} passes, "
                          f"{cb['wall']:.2f} s wall on {nall} OpenMP threads = every CPU this process may use "
                          f"(os.cpu_count() = {os.cpu_count()}, affinity / cgroup quota = {host_threads()}); also 1 thread "
                          f"(64 frames) and {nthr} threads; C restatement of the .m reference "
                          f"(oracle/c/ofdm_oracle.c, float64), not MATLAB"}
            res["ber_match"] = {"frames": int(ncpu), "gpu_errors": int(gpu_errs.sum()),
                                "oracle_errors": int(cb["errors"].sum()),
                                "max_abs_diff_per_frame": int(np.max(np.abs(gpu_errs - cb["errors"])))}
        if world == 1 and not args.no_f64 and args.precision == "fp32":
            del data, rx, ref, out, plan
            torch.cuda.empty_cache()
            res["f64"] = f64_leg(ofdm, fr, cfg, dev, dev_index, F, args.f64_steps, bps, 0 if args.no_cpu else 256)
        if world == 1 and not args.no_secondary:
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs
            res["secondary"] = bench_configs.secondary_all()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
