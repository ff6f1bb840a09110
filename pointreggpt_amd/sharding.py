"""Scene-index sharding across the GPUs of a node: scenes are independent, so rank r of W takes a
disjoint slice and nothing is exchanged on the data path.  The reference leaves this to the user's -start/-stop
flags (generate_dataset.py:16-25); here it is derived from RANK / WORLD_SIZE so one torchrun launch covers a range."""
from __future__ import annotations

import os
from typing import List, Tuple


def rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(start: int, stop: int, rank: int, world: int, batch: int = 1) -> Tuple[int, int]:
    """Contiguous block partition of [start, stop) in units of `batch` scenes (keeps the reference's batch-skip
    resume logic valid per rank: sd:2371-2381): rank r gets blocks r*q + min(r, rem) ... ."""
    n = max(0, stop - start)
    nblocks = (n + batch - 1) // batch
    q, rem = divmod(nblocks, world)
    b0 = rank * q + min(rank, rem)
    b1 = b0 + q + (1 if rank < rem else 0)
    return start + min(n, b0 * batch), start + min(n, b1 * batch)


def num_to_groups(num: int, divisor: int) -> List[int]:
    """[divisor]*k + [remainder] (sd:538-544)."""
    groups, rem = divmod(num, divisor)
    return [divisor] * groups + ([rem] if rem > 0 else [])


def job_seed(timeout_s: float = 120.0) -> int:
    """ONE fresh 63-bit seed per launch, the same on every rank.

    * `PRG_JOB_SEED` overrides (reproducing a logged run).
    * single process: `secrets.randbits(63)`.
    * WORLD_SIZE > 1: rank 0 DRAWS the seed with `secrets.randbits` and PUBLISHES it through a c10d TCPStore that it hosts on
      MASTER_ADDR:MASTER_PORT (torchrun exports both to every rank of every node, and this program opens no process group of
      its own on that port); the other ranks read it.  Nothing is derived from launcher ids or pids, so the value is fresh on
      every launch — including launches with a fixed `--rdzv_id` and port — and identical across nodes; an elastic restart
      re-launches the ranks and therefore draws a NEW seed (pass `--noise_seed` / PRG_JOB_SEED to pin one across restarts).
    Raises when the store cannot be reached: a silent per-rank seed would break the shard-invariance of the noise keys."""
    import secrets
    if os.environ.get("PRG_JOB_SEED"):
        return int(os.environ["PRG_JOB_SEED"]) & ((1 << 63) - 1)
    rank, world, _ = rank_world()
    if world <= 1:
        return secrets.randbits(63)
    from datetime import timedelta

    import torch.distributed as dist
    host, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT")
    if not port:
        raise RuntimeError("WORLD_SIZE > 1 without MASTER_PORT: pass --noise_seed (or PRG_JOB_SEED) so that all ranks agree on one seed")
    store = dist.TCPStore(host, int(port), world, is_master=(rank == 0), timeout=timedelta(seconds=timeout_s), wait_for_workers=False)
    if rank == 0:
        store.set("prg_job_seed", str(secrets.randbits(63)))
    seed = int(store.get("prg_job_seed").decode())
    store.add("prg_job_seed_readers", 1)
    if rank == 0:                      # keep the store alive until every rank has read it
        import time
        t0 = time.time()
        while int(store.add("prg_job_seed_readers", 0)) < world and time.time() - t0 < timeout_s:
            time.sleep(0.01)
    return seed


def batch_pose_seed(job_seed_value: int, first_scene_index: int, sample_index: int) -> int:
    """Seed of numpy's legacy global stream for one batch's `random_sample_pose` draw (sd:417-443): a function of the
    job seed and the batch's first scene only, so re-sharding or resuming the same job reproduces the same poses."""
    import numpy as np
    return int(np.random.SeedSequence([int(job_seed_value) & 0xFFFFFFFFFFFFFFFF, int(first_scene_index),
                                       int(sample_index), 0x706F7365]).generate_state(1, np.uint32)[0])
