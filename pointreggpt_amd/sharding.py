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
    store = None
    if rank == 0:
        try:
            store = dist.TCPStore(host, int(port), world, is_master=True, timeout=timedelta(seconds=timeout_s), wait_for_workers=False)
        except RuntimeError:       # (DistNetworkError, EADDRINUSE) the port is already served — torchrun's agent store: be its client
            store = None
    if store is None:
        store = dist.TCPStore(host, int(port), world, is_master=False, timeout=timedelta(seconds=timeout_s), wait_for_workers=False)
    # Under torchrun (torch >= 2.x, use_agent_store) MASTER_PORT is the AGENT's store, which outlives worker restarts: rank 0's
    # bind then fails quietly and every rank is a client of a store that may still hold the previous attempt's keys (ADVICE
    # round 4).  Both keys are therefore namespaced per attempt, and rank 0 publishes with compare_set so that a key that
    # already exists (a stale attempt with the same restart count cannot, a second job_seed() call in this process can) is
    # never half-overwritten: every rank — rank 0 included — returns the value the store holds.
    attempt = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    run = os.environ.get("TORCHELASTIC_RUN_ID", "")
    key = f"prg_job_seed/{run}/{attempt}/{_job_seed_calls[0]}"
    _job_seed_calls[0] += 1
    if rank == 0:
        store.compare_set(key, "", str(secrets.randbits(63)))
    seed = int(store.get(key).decode())
    store.add(key + "/readers", 1)
    if rank == 0:                      # keep the store alive until every rank has read it
        import time
        t0 = time.time()
        while int(store.add(key + "/readers", 0)) < world and time.time() - t0 < timeout_s:
            time.sleep(0.01)
    return seed


_job_seed_calls = [0]      # job_seed() calls in this process (all ranks call it the same number of times)


def batch_pose_seed(job_seed_value: int, first_scene_index: int, sample_index: int) -> int:
    """Seed of numpy's legacy global stream for one batch's `random_sample_pose` draw (sd:417-443): a function of the
    job seed and the batch's first scene only, so re-sharding or resuming the same job reproduces the same poses."""
    import numpy as np
    return int(np.random.SeedSequence([int(job_seed_value) & 0xFFFFFFFFFFFFFFFF, int(first_scene_index),
                                       int(sample_index), 0x706F7365]).generate_state(1, np.uint32)[0])


# ---------------------------------------------------------------------------------------------------------------------
# CPU placement of a rank (round 5, SURVEY 8e: the >= 0.97 scaling target is lost on the HOST side, not on the interconnect)
# ---------------------------------------------------------------------------------------------------------------------
def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_local_cpus(device_index: int) -> Tuple[List[int], int]:
    """(CPUs of the NUMA node the GPU hangs off, that node's id) from sysfs — `/sys/bus/pci/devices/<bdf>/local_cpulist` and
    `numa_node` of the HIP device's PCI function — or ([], -1) when the topology is not exposed (containers, single-node hosts
    report numa_node = -1 and the full CPU list)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        node = int(open(base + "/numa_node").read().strip())
        cpus = _parse_cpulist(open(base + "/local_cpulist").read())
        return cpus, node
    except Exception:      # noqa: BLE001 — placement is best effort; the caller falls back to an even split
        return [], -1


def rank_cpu_set(local_rank: int, local_world: int, allowed: List[int], gpu_cpus: List[int], peers_on_node: int | None = None,
                 index_on_node: int | None = None) -> List[int]:
    """The CPUs rank `local_rank` of `local_world` should run on: the GPU's NUMA-local CPUs (those the process may use), cut
    evenly among the `peers_on_node` ranks whose GPUs share that node (index_on_node = this rank's position among them); without
    topology, an even contiguous cut of the allowed set.  Pure function (tested on CPU)."""
    allowed = sorted(allowed)
    local = [c for c in allowed if c in set(gpu_cpus)]
    if local and peers_on_node and index_on_node is not None and len(local) >= peers_on_node:
        pool, k, n = local, index_on_node, peers_on_node
    else:
        pool, k, n = allowed, local_rank, max(1, local_world)
    if len(pool) < n:                # fewer CPUs than ranks: share everything
        return pool
    q, r = divmod(len(pool), n)
    lo = k * q + min(k, r)
    return pool[lo:lo + q + (1 if k < r else 0)]


def pin_rank_cpus(local_rank: int, local_world: int, device_index: int | None = None) -> dict:
    """Pin THIS process (its existing threads and every thread it starts afterwards: the lane threads, the C++ writer pool) to its share of the host:
    the cores of the NUMA node its GPU is attached to, divided among the ranks on that node.  One rank issues ~1000 graph
    launches per batch from each lane thread and runs cpu_count / world / 2 writer threads; unpinned, eight ranks' threads
    migrate across both sockets of a 2-socket host.  `PRG_NO_AFFINITY=1` opts out; a single rank is left alone.
    Returns what was done (bench.py prints it)."""
    info = {"pinned": False}
    if os.environ.get("PRG_NO_AFFINITY", "0") not in ("", "0") or local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        info["reason"] = "PRG_NO_AFFINITY" if os.environ.get("PRG_NO_AFFINITY", "0") not in ("", "0") else "single rank"
        return info
    allowed = sorted(os.sched_getaffinity(0))
    dev = local_rank if device_index is None else device_index
    gpu_cpus, node = gpu_local_cpus(dev)
    peers = idx = None
    if gpu_cpus and node >= 0:
        # ranks are one per GPU in device order: count the devices on this GPU's node to find this rank's share of it
        # (more ranks than devices — the gloo rehearsal — : rank r sits on device r % ndev, as bench.py / generate_dataset.py place it)
        try:
            import torch
            ndev = max(1, torch.cuda.device_count())
            dev_nodes = [gpu_local_cpus(d)[1] for d in range(ndev)]
            same = [r for r in range(local_world) if dev_nodes[r % ndev] == node]
            if local_rank in same:
                peers, idx = len(same), same.index(local_rank)
        except Exception:      # noqa: BLE001
            peers = idx = None
    cpus = rank_cpu_set(local_rank, local_world, allowed, gpu_cpus, peers, idx)
    if cpus:
        # Reading the GPU's PCI address initialised the HIP runtime, whose helper threads already exist, and sched_setaffinity(0)
        # only moves the CALLING thread: every thread of the process is moved, then torch's intra-op pool is
        # sized for the share (it was sized for the whole host at import).
        threads = 0
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")]
        except OSError:
            tids = []
        for tid in tids:
            try:
                os.sched_setaffinity(tid, cpus)
                threads += 1
            except OSError:        # a thread that exited meanwhile
                pass
        os.sched_setaffinity(0, cpus)
        try:
            import torch
            torch.set_num_threads(max(1, len(cpus)))
        except Exception:      # noqa: BLE001
            pass
        # rank_cpu_budget() tells a mask THIS function narrowed (the rank's private share) from one a container / taskset narrowed
        # (shared by every rank of the job) by this marker; exported so that worker processes started later see it too
        os.environ["PRG_PINNED_CPUS"] = str(len(cpus))
        info.update(pinned=True, cpus=len(cpus), first_cpu=cpus[0], last_cpu=cpus[-1], numa_node=node, threads_moved=threads,
                    source="sysfs local_cpulist" if peers else "even split of the allowed CPUs")
    return info
