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


def job_seed() -> int:
    """A fresh 63-bit seed that is THE SAME on every rank of one launch (no collective needed): derived from the
    launcher's run id when there is one (torchrun exports TORCHELASTIC_RUN_ID and MASTER_ADDR/PORT identically to all ranks;
    `--standalone` run ids are random per launch), from fresh entropy in a single-process run.  PRG_JOB_SEED overrides."""
    import hashlib
    import secrets
    if os.environ.get("PRG_JOB_SEED"):
        return int(os.environ["PRG_JOB_SEED"]) & ((1 << 63) - 1)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        key = "|".join(os.environ.get(k, "") for k in ("TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT",
                                                        "TORCHELASTIC_RESTART_COUNT"))
        run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
        if run_id in ("", "none"):
            # a fixed run id ('none' is torchrun's default): mix in the launcher's pid, which all ranks share as their parent
            key += "|ppid=%d" % os.getppid()
        return int.from_bytes(hashlib.sha256(key.encode()).digest()[:8], "little") & ((1 << 63) - 1)
    return secrets.randbits(63)


def batch_pose_seed(job_seed_value: int, first_scene_index: int, sample_index: int) -> int:
    """Seed of numpy's legacy global stream for one batch's `random_sample_pose` draw (sd:417-443): a function of the
    job seed and the batch's first scene only, so re-sharding or resuming the same job reproduces the same poses."""
    import numpy as np
    return int(np.random.SeedSequence([int(job_seed_value) & 0xFFFFFFFFFFFFFFFF, int(first_scene_index),
                                       int(sample_index), 0x706F7365]).generate_state(1, np.uint32)[0])
