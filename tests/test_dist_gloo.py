"""CPU: the N>1 path of the sweep (round-robin tiles + one SUM all-reduce of the counters) with
world_size-2 gloo processes.  On the GPU node the same code runs with backend "nccl" (= RCCL)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tile_result(snr_idx, batch_idx):
    """Deterministic stand-in for one (SNR, batch) tile's (bit errors, bits)."""
    rng = np.random.default_rng([snr_idx, batch_idx])
    return int(rng.integers(0, 1000)), 32256 * 8


def _worker(rank, world, port, n_snr, n_batches, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ofdm_course_amd import sweep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = sweep.Counters(n_snr)
    for s, b in sweep.tiles_for_rank(n_snr, n_batches, rank, world):
        e, n = _tile_result(s, b)
        c.add(s, 0, e, n)
    tot = sweep.all_reduce_counters(c)
    q.put((rank, tot.errors.copy(), tot.bits.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sweep_matches_single_process():
    n_snr, n_batches, world = 20, 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_snr, n_batches, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_e = np.zeros((n_snr, 1), np.int64)
    want_n = np.zeros((n_snr, 1), np.int64)
    for s in range(n_snr):
        for b in range(n_batches):
            e, n = _tile_result(s, b)
            want_e[s, 0] += e
            want_n[s, 0] += n
    for _, e, n in res:                 # every rank holds the global totals, bit-identical
        assert np.array_equal(e, want_e) and np.array_equal(n, want_n)


def test_bench_parent_reports_a_failed_rank_instead_of_hanging():
    """bench.py's own launcher (`python bench.py --gpus 2`): on this GPU-less host both ranks fail at
    torch.cuda.set_device; the parent must relay that as a non-zero exit code quickly, without any GPU call itself."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--frames", "8", "--backend", "gloo", "--no-cpu"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0 and r.stdout.strip().startswith("{")
    else:
        assert r.returncode == 1 and "ranks failed" in r.stderr, (r.returncode, r.stderr[-500:])
