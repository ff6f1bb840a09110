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
