"""Monte-Carlo sweep sharding over the GPUs of one node.

Units are (snr_idx, batch_idx) tiles -- in reference terms the (kk, jj) pairs of
T5/Task5_part2.m:46,:148 or the SNR loop iterations of T5/Main_model_Task_5.m:305.  No unit reads
another unit's data, so ranks never exchange samples: the only collective is one SUM all-reduce of
the error / bit counters (integers, hence bit-identical for any reduction order) at the end.
One process per GPU; torch.distributed backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np


def flatten_tiles(n_snr: int, n_batches: int) -> List[Tuple[int, int]]:
    """Tile list in (batch-major, snr-minor) order so consecutive tiles have different SNRs."""
    return [(s, b) for b in range(n_batches) for s in range(n_snr)]


def tiles_for_rank(n_snr: int, n_batches: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """Round-robin deal of the flattened tile list (never shard by SNR alone: 20 SNR points do not
    divide by 8 GPUs)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    return flatten_tiles(n_snr, n_batches)[rank::world]


def frame_range_for_rank(frames_per_rank: int, rank: int) -> Tuple[int, int]:
    """Weak-scaling shard of the benchmark: rank r owns global frames [r*F, (r+1)*F)."""
    return rank * frames_per_rank, (rank + 1) * frames_per_rank


def tile_seed_stream(base_seed: int, snr_idx: int, batch_idx: int, frames_per_tile: int) -> Tuple[int, int]:
    """(Philox key, first stream id) of a tile: depends on the tile only, never on the rank, so the
    sweep's result is independent of the GPU count."""
    return int(base_seed) + 1000003 * int(snr_idx), int(batch_idx) * int(frames_per_tile)


class Counters:
    """bit_errors / bits_total per (snr, estimator); int64 so the all-reduce is exact."""

    def __init__(self, n_snr: int, n_est: int = 1):
        self.errors = np.zeros((n_snr, n_est), dtype=np.int64)
        self.bits = np.zeros((n_snr, n_est), dtype=np.int64)

    def add(self, snr_idx: int, est_idx: int, n_errors: int, n_bits: int):
        self.errors[snr_idx, est_idx] += int(n_errors)
        self.bits[snr_idx, est_idx] += int(n_bits)

    def ber(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            return self.errors / self.bits


def dist_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def all_reduce_counters(counters: Counters, device=None) -> Counters:
    """SUM all-reduce of the counters over the default process group (no-op when not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return counters
    t = torch.from_numpy(np.stack([counters.errors, counters.bits]).astype(np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu().numpy()
    out = Counters(*counters.errors.shape)
    out.errors[:] = t[0]
    out.bits[:] = t[1]
    return out
