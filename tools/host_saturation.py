#!/usr/bin/env python
"""Host-side 8x saturation rehearsal: can ONE host feed eight ranks at the headline rate?

No 8-GPU box is needed: what a rank's HOST does per batch does not depend on which GPU produced the tensors.  So

  1. RECORD (one GPU batch, the real networks): Generator.generate runs one batch of B synthetic scenes through the real
     pipeline (bf16, 1000-step ancestral DDNM by default); the two networks' outputs are captured — the MaskUnet
     keep-probabilities before and after sampling and the sampler's images;
  2. REPLAY (GPU idle): R processes (default 8), each placed by sharding.pin_rank_cpus like a rank of the real job, run the
     UNMODIFIED Generator.generate loop — scene synthesis, scene directories, memory-cloud crop, z-buffer / mask / float64
     unprojection launches (microseconds), device -> host copies, two lane threads, the C++ WriterPool with every file of the
     reference layout (2 PLY + 5 PNG + 2 text per pair) — with the two networks replaced by stubs that return the recorded
     tensors at once; then generate_gt over their scenes (gg:105-175) and, on rank 0, gather_gt (gg:177-188).  Scene indices
     are mapped modulo B onto the recorded batch so that geometry and network outputs stay consistent (the files of scene k
     and scene k + B are identical; rank 0 checks its first scene's generated cloud byte for byte against the recording run).

The lanes never wait for a GPU here, so the measured rate is the host's CAPACITY: sustained pairs/s of all ranks together,
CPU seconds per pair per process, bytes/s.  Target: >= 1.3 x (8 x 15 pairs/s = 120 pairs/s).

  python tools/host_saturation.py [--ranks 8] [--batches 5] [--batch 64] [--size 128] [--out gpurun_out/host_saturation.json]
"""
import argparse
import hashlib
import json
import os
import resource
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--ranks", type=int, default=8)
    p.add_argument("--batches", type=int, default=20, help="batches per rank in the replay")
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--size", type=int, default=128)
    p.add_argument("--lanes", type=int, default=2)
    p.add_argument("--record-steps", type=int, default=None, help="DDIM steps of the recording run (default: 1000-step ancestral)")
    p.add_argument("--target", type=float, default=120.0, help="pairs/s the host must sustain (8 x 15)")
    p.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "host_saturation_8x.json"))
    p.add_argument("--worker", default=None, help=argparse.SUPPRESS)      # internal: work directory of a replay rank
    return p.parse_args()


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def make_replay_generator(B, seed, **kw):
    from pointreggpt_amd.generator import Generator

    class ReplayGenerator(Generator):
        """Scene index modulo the recorded batch: every batch has the recorded batch's geometry."""

        def _scene_inputs(self, abs_idx, info_train, scene_dir):
            return super()._scene_inputs(abs_idx % B, info_train, scene_dir)

        def _poses(self, idxs, sample_idx, pose_seed=None):
            return super()._poses([i % B for i in idxs], sample_idx, pose_seed)

    return ReplayGenerator(synthetic_seed=seed, **kw)


def record(a, work):
    """One real batch on the GPU; returns the path of the recording and what the recording run wrote for scene 0."""
    import torch
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import MaskUnet, Unet
    B, S = a.batch, a.size
    unet = Unet(64, dtype="bf16").init_synthetic(seed=1, calibrated=True)
    mask = MaskUnet(64, dtype="bf16").init_synthetic(seed=2, calibrated=True)
    diff = GaussianDiffusion(unet, image_size=S, timesteps=1000, sampling_timesteps=a.record_steps)
    rec = {"probs": []}

    class RecModel:
        image_size = S

        def sample(self, **kw):
            out = diff.sample(**kw)
            rec["images"] = out.clone()
            return out

    def rec_mask(x):
        p = mask(x)
        rec["probs"].append(p.clone())
        return p

    gen = make_replay_generator(B, 0, diffusion_model=RecModel(), folder=None, batch_size=B, samples_folder=os.path.join(work, "record", "data"))
    t0 = time.perf_counter()
    gen.generate(0, B, 1, depth_correction=rec_mask, noise_seed=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    path = os.path.join(work, "recording.npz")
    np.savez(path, images=rec["images"].cpu().numpy(), prob1=rec["probs"][0].cpu().numpy(), prob2=rec["probs"][1].cpu().numpy())
    ref_file = os.path.join(work, "record", "data", "scene-000000", "sample-000001.cloud.ply")
    info = {"seconds_one_real_batch_incl_setup": dt, "transitions": len(diff.step_table()),
            "points_kept_fraction": float((rec["probs"][1] > 0.99).float().mean()),
            "scene0_generated_cloud_sha256": sha(ref_file), "scene0_generated_cloud_bytes": os.path.getsize(ref_file)}
    diff.close(); unet.close(); mask.close()
    return path, info


def worker(a):
    """One replay rank: pinned like a real rank, two lanes, stub networks, every file written, then generate_gt."""
    work = a.worker
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from pointreggpt_amd import sharding
    import torch
    ndev = max(1, torch.cuda.device_count())
    aff = sharding.pin_rank_cpus(rank, world, device_index=rank % ndev)
    torch.cuda.set_device(rank % ndev)
    dev = torch.device("cuda", rank % ndev)
    from pointreggpt_amd.generator import gather_gt, generate_gt
    B, S = a.batch, a.size
    r = np.load(os.path.join(work, "recording.npz"))
    images, prob1, prob2 = (torch.from_numpy(r[k]).to(dev) for k in ("images", "prob1", "prob2"))

    class StubModel:                      # GaussianDiffusion.sample: the recorded images, at once (a fresh tensor: the caller masks it)
        image_size = S

        def sample(self, **kw):
            return images.clone()

    class StubMask:                       # MaskUnet: called on the reprojection, then on the sampled images
        def __init__(self):
            self.n = 0

        def __call__(self, x):
            self.n += 1
            return (prob1 if self.n % 2 == 1 else prob2).clone()

    root = os.path.join(work, f"rank{rank}")
    gen = make_replay_generator(B, 0, diffusion_model=StubModel(), folder=None, batch_size=B, samples_folder=os.path.join(root, "ds", "data"))
    lanes = [(StubModel(), StubMask()) for _ in range(1, a.lanes)]
    first = rank * a.batches * B
    stop = first + a.batches * B
    st = {}
    # warm-up outside the clock: one batch per lane (library load, pool start, first-touch of the page cache)
    gen.generate(10_000_000 + first, 10_000_000 + first + a.lanes * B, 1, depth_correction=StubMask(), noise_seed=0, lanes=lanes, stats=st)
    shutil.rmtree(os.path.join(root, "ds", "data"), ignore_errors=True)
    os.makedirs(os.path.join(root, "ds", "data"), exist_ok=True)
    open(os.path.join(work, f"ready{rank}"), "w").close()
    while not os.path.exists(os.path.join(work, "go")):
        time.sleep(0.001)
    ru0, t0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
    gen.generate(first, stop, 1, depth_correction=StubMask(), noise_seed=0, lanes=lanes, stats=st)
    torch.cuda.synchronize()
    t1, ru1 = time.perf_counter(), resource.getrusage(resource.RUSAGE_SELF)
    generate_gt("ds", first, stop, 2, root=root)
    t2, ru2 = time.perf_counter(), resource.getrusage(resource.RUSAGE_SELF)
    data = os.path.join(root, "ds", "data")
    nbytes = sum(os.path.getsize(os.path.join(d, f)) for d, _s, fs in os.walk(data) for f in fs)
    nfiles = sum(len(fs) for _d, _s, fs in os.walk(data))
    gt_lines = sum(sum(1 for _ in open(os.path.join(data, s, "gt.log"))) for s in os.listdir(data) if os.path.exists(os.path.join(data, s, "gt.log")))
    out = {"rank": rank, "pairs": a.batches * B, "generate_s": t1 - t0, "gt_s": t2 - t1, "t_start": t0, "t_gen_end": t1, "t_end": t2,
           "cpu_s_generate": (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime),
           "cpu_s_gt": (ru2.ru_utime + ru2.ru_stime) - (ru1.ru_utime + ru1.ru_stime),
           "bytes_written": nbytes, "files_written": nfiles, "gt_lines": gt_lines, "writer_threads": st.get("writer_threads"),
           "lanes": st.get("lanes"), "affinity": aff, "cpus_allowed": len(os.sched_getaffinity(0)),
           "first_scene_cloud_sha256": sha(os.path.join(data, "scene-{:0>6d}".format(first), "sample-000001.cloud.ply"))}
    json.dump(out, open(os.path.join(work, f"result{rank}.json"), "w"))
    if rank == 0:
        # gather_gt over rank 0's scenes only (each rank's dataset lives in its own directory here; in the product all ranks
        # share one tree and rank 0 concatenates every scene log in index order)
        gather_gt("ds", first, stop, root=root)


def main():
    a = parse()
    if a.worker:
        worker(a)
        return
    import torch
    assert torch.cuda.is_available(), "needs a HIP device (recording run + the geometry launches of the replay)"
    work = tempfile.mkdtemp(prefix="prg_hostsat_")
    try:
        t_rec0 = time.perf_counter()
        _path, rec_info = record(a, work)
        rec_s = time.perf_counter() - t_rec0
        env = {k: v for k, v in os.environ.items() if k not in ("PRG_NO_AFFINITY", "PRG_PINNED_CPUS")}
        procs = []
        for r in range(a.ranks):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.ranks), LOCAL_WORLD_SIZE=str(a.ranks))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", work, "--batches", str(a.batches), "--batch", str(a.batch),
                                           "--size", str(a.size), "--lanes", str(a.lanes)], env=e))
        t_wait = time.time()
        while not all(os.path.exists(os.path.join(work, f"ready{r}")) for r in range(a.ranks)):
            if any(p.poll() not in (None, 0) for p in procs) or time.time() - t_wait > 900:
                raise RuntimeError("a replay rank failed before the start line")
            time.sleep(0.01)
        open(os.path.join(work, "go"), "w").close()
        rcs = [p.wait() for p in procs]
        assert not any(rcs), rcs
        res = [json.load(open(os.path.join(work, f"result{r}.json"))) for r in range(a.ranks)]
        pairs = sum(x["pairs"] for x in res)
        t0 = min(x["t_start"] for x in res)          # CLOCK_MONOTONIC is system-wide on Linux: comparable across processes
        gen_wall = max(x["t_gen_end"] for x in res) - t0
        all_wall = max(x["t_end"] for x in res) - t0
        out = {
            "what": "host-side saturation rehearsal: R ranks' UNMODIFIED Generator.generate loops (pinned by sharding.pin_rank_cpus, 2 lanes, C++ "
                    "WriterPool, every file of the reference layout) + generate_gt, with the two networks replaced by stubs that return one "
                    "recorded real batch's outputs at once (tools/host_saturation.py); the lanes never wait for a GPU: this is host CAPACITY",
            "host": {"cpu_count": os.cpu_count(), "cpus_allowed": len(os.sched_getaffinity(0)), "hip_devices": torch.cuda.device_count()},
            "ranks": a.ranks, "lanes_per_rank": a.lanes, "batch": a.batch, "image_size": a.size, "batches_per_rank": a.batches, "pairs": pairs,
            "recording": dict(rec_info, seconds_total=rec_s),
            "replay_matches_recording": all(x["first_scene_cloud_sha256"] == rec_info["scene0_generated_cloud_sha256"] for x in res),
            "sustained_pairs_per_s_generate": pairs / gen_wall,
            "sustained_pairs_per_s_generate_plus_gt": pairs / all_wall,
            "wall_s": {"generate_max_over_ranks": gen_wall, "generate_plus_gt_max_over_ranks": all_wall,
                       "generate_per_rank": [round(x["generate_s"], 3) for x in res], "gt_per_rank": [round(x["gt_s"], 3) for x in res]},
            "cpu_seconds_per_pair_per_process": {"generate": sum(x["cpu_s_generate"] for x in res) / pairs, "generate_gt": sum(x["cpu_s_gt"] for x in res) / pairs},
            "cores_busy_mean_during_generate": sum(x["cpu_s_generate"] for x in res) / gen_wall,
            "bytes_per_pair": sum(x["bytes_written"] for x in res) / pairs, "files_per_pair": sum(x["files_written"] for x in res) / pairs,
            "write_rate_MBps": sum(x["bytes_written"] for x in res) / gen_wall / 1e6,
            "gt_lines": sum(x["gt_lines"] for x in res),
            "target_pairs_per_s": a.target, "headroom_over_target": pairs / all_wall / a.target,
            "per_rank": [{k: x[k] for k in ("rank", "generate_s", "gt_s", "cpu_s_generate", "writer_threads", "lanes", "cpus_allowed", "affinity")} for x in res],
        }
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
        print(json.dumps({k: out[k] for k in ("host", "ranks", "pairs", "replay_matches_recording", "sustained_pairs_per_s_generate",
                                              "sustained_pairs_per_s_generate_plus_gt", "cpu_seconds_per_pair_per_process",
                                              "cores_busy_mean_during_generate", "write_rate_MBps", "headroom_over_target")}, indent=1))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
