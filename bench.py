#!/usr/bin/env python
"""Benchmark of the generative data path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

One *step* = one batch of B synthetic scene pairs through the hot path (BASELINE.json configs[1]/[2]):
z-buffer SE(3) reprojection -> MaskUnet -> DDNM condition -> T-step sampler over the conditional U-Net ->
MaskUnet -> float64 unprojection, everything resident in HBM when the clock starts.  Scene pairs are
independent, so ranks take disjoint scene indices and never communicate on the data path (weak scaling);
`value` = pairs produced by all ranks / max-over-ranks wall time.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNET_GFLOP = {64: 14.744, 128: 58.976, 256: 236.282}      # per image, SURVEY.md §8d / BASELINE.md §2
MASK_GFLOP = {64: 14.787, 128: 59.173, 256: 237.096}
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "mxfp8": 5000.0}   # dense, MI355X_MICROARCH.md chip table


def conv_sources_hash():
    import hashlib
    h = hashlib.sha256()
    for f in ("conv_ws.hip", "conv_c64.hip", "conv_w256.hip", "conv.hip", "conv.h", "common.h"):
        h.update(open(os.path.join(ROOT, "pointreggpt_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1, help="timed batches per rank")
    p.add_argument("--warmup", type=int, default=1, help="untimed batches per rank")
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--size", type=int, default=128)
    p.add_argument("--timesteps", type=int, default=1000)
    p.add_argument("--sampling-steps", type=int, default=None, help="< timesteps selects DDIM (default: ancestral DDNM)")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "mxfp8"])
    p.add_argument("--dim", type=int, default=64)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--sampler-only", action="store_true", help="configs[1]: p_sample_loop only (no geometry / MaskUnet)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--profile-transitions", type=int, default=40)
    p.add_argument("--no-e2e-files", action="store_true", help="skip the generate_dataset leg (files on disk)")
    p.add_argument("--e2e-batches", type=int, default=2)
    p.add_argument("--no-drift", action="store_true", help="skip the bf16-vs-fp32 end-to-end drift measurement")
    p.add_argument("--drift-scenes", type=int, default=4)
    return p.parse_args()


def cpu_baseline(size, dim):
    """The reference's arithmetic on the host cores: the oracle's p_sample (torch-CPU / oneDNN, fp32) on a bounded
    sample — a thread-count sweep, then 12 timed transitions at batch 4 — extrapolated to pairs/s for the full chain."""
    from oracle import diffusion as OD
    from oracle import unet as OU
    from pointreggpt_amd import weights as W
    cores = os.cpu_count() or 1
    B, T = 4, 1000
    p = W.synth_state_dict(W.unet_config(dim), 0)
    sch = OD.schedule(T)
    den = lambda x, t, c: OU.unet_forward(p, x, t, c)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B, 1, size, size), generator=g)
    pc = torch.tensor([[151.5, 152.1, size / 2 + 0.5, size / 2]] * B)
    cond = torch.cat([torch.rand((B, 1, size, size), generator=g) * 2 - 1,
                      (torch.rand((B, 1, size, size), generator=g) > 0.5).float() * 2 - 1], 1)
    nz = torch.randn((B, 1, size, size), generator=g)

    def transition(i):
        t0 = time.perf_counter()
        OD.p_sample(sch, den, x, 999 - i, pc, cond, nz)
        return time.perf_counter() - t0

    # oneDNN does not scale to every core of a big host at this batch: pick the fastest thread count first
    best, used = None, 1
    for th in [c for c in (8, 16, 32, 64, 128) if c <= cores] or [cores]:
        torch.set_num_threads(th)
        transition(0)
        d = transition(1)
        if best is None or d < best:
            best, used = d, th
        if d > 2.5 * best:
            break
    torch.set_num_threads(used)
    n = 12                       # ~6 s of timed CPU work after the ~10 s thread sweep
    dt = sum(transition(2 + i) for i in range(n)) / n
    cores = used
    pairs_per_s = B / (dt * T)
    return {"value": pairs_per_s, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"oracle p_sample (torch-CPU fp32, {used} threads = fastest of a sweep on a {os.cpu_count()}-core host), "
                      f"batch {B}, {size}x{size}, {n} timed transitions = {dt:.3f} s/transition, extrapolated x{T} "
                      f"transitions; MaskUnet/geometry (0.2% of the work) omitted"}


def mem_rooflines(G, bt, S, B, pr):
    """HIP-event bandwidth of the memory-bound kernels of one pair (north_star: 'coalesced HBM loads ... evidenced by
    HBM-GB/s'): algorithmic bytes per pixel (fp32 images, float64 points) / average duration of
    `reps` back-to-back C-ABI calls into preallocated outputs on torch's current stream (the stream they are launched on)."""
    import ctypes as C
    from pointreggpt_amd import _lib
    lib = _lib.load()
    reps = 100
    npx = B * S * S
    dev = bt["depth"].device
    depth, K, pose = bt["depth"].contiguous(), bt["K"].contiguous(), bt["pose"].contiguous()
    rpj = torch.empty_like(depth)
    hit = torch.empty((B, 1, S, S), dtype=torch.uint8, device=dev)
    xyz = torch.empty((B, S * S, 3), dtype=torch.float64, device=dev)
    valid = torch.empty((B, S * S), dtype=torch.uint8, device=dev)
    aug = torch.empty((B, 3, S, S), dtype=torch.float32, device=dev)
    prob = torch.rand((B, 1, S, S), device=dev)
    d2 = torch.empty_like(depth)
    h2 = torch.empty_like(hit)
    cond = torch.empty((B, 2, S, S), dtype=torch.float32, device=dev)
    P, st = _lib.ptr, _lib.stream_ptr()
    cases = {
        "reproject_zbuffer (unproject + SE(3) + atomicMin z-buffer + resolve)":
            (9, lambda: lib.prg_reproject_zbuffer(P(depth), P(K), P(pose), P(rpj), P(hit), B, S, S, 10.0, 0.0, 10.0, 0.1, st)),
        "unproject_f64 (+ inverse pose)":
            (4 + 24 + 1, lambda: lib.prg_unproject_f64(P(rpj), P(K), P(pose), P(xyz), P(valid), B, S, S, 10.0, 0.5, 10.0, st)),
        "depth_augment": (4 + 12, lambda: lib.prg_depth_augment(P(rpj), P(aug), B, S, S, st)),
        "apply_mask (+ condition assembly)":
            (4 + 4 + 1 + 4 + 1 + 8, lambda: lib.prg_apply_mask(P(prob), P(rpj), P(hit), 0.5, P(d2), P(h2), P(cond), B, S, S, st)),
    }
    out = {}
    for name, (bpp, fn) in cases.items():
        _lib.check(fn())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out[name] = {"bytes_per_px": bpp, "avg_us": us, "GBps": bpp * npx / (us * 1e-6) / 1e9,
                     "note": f"{reps} back-to-back C-ABI calls, torch events on the launch stream (includes launch gaps)"}
    if pr.get("step_launches"):
        us = pr["step_ms"] * 1e3 / pr["step_launches"]
        out["sampler_step (x0, DDNM replace, posterior/DDIM, Philox noise)"] = {
            "bytes_per_px": 20, "avg_us": us, "GBps": 20 * npx / (us * 1e-6) / 1e9,
            "note": "HIP events inside the library around every launch of the profiled transitions (eager launches)"}
    for v in out.values():
        v["frac_of_8TBps"] = v["GBps"] / 8000.0
    return {"bound": "hbm", "peak_GBps": 8000.0, "pixels_per_launch": npx, "kernels": out,
            "reading": "1 Mpx per launch = 9-30 MB per call: microsecond launches, latency- not bandwidth-bound; together "
                       "< 0.01 % of a 1000-step pair"}


def e2e_files(a, unet, mask, diff, rank, world, B, S):
    """generate_dataset.py's own loop (Generator.generate, synthetic scenes) for `e2e_batches` batches INCLUDING every file
    of the reference's layout (2 PLY + 5 PNG + 2 text files per pair): host post-processing runs on the library's C++
    writer pool while the GPU samples the next batch.  Returns pairs on disk / wall time of this rank."""
    import shutil
    import tempfile
    from pointreggpt_amd.generator import Generator
    root = tempfile.mkdtemp(prefix=f"prg_e2e_r{rank}_")
    try:
        gen = Generator(diff, None, batch_size=B, samples_folder=os.path.join(root, "data"), synthetic_seed=a.seed)
        first = 10_000_000 + rank * (a.e2e_batches + 1) * B
        st = {}
        gen.generate(first, first + B, 1, depth_correction=mask, stats=st)          # warm-up batch (graph capture, pool start)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gen.generate(first + B, first + (a.e2e_batches + 1) * B, 1, depth_correction=mask, stats=st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nfiles = sum(len(f) for _d, _s, f in os.walk(os.path.join(root, "data")))
        nbytes = sum(os.path.getsize(os.path.join(d, f)) for d, _s, fs in os.walk(os.path.join(root, "data")) for f in fs)
        return {"pairs": a.e2e_batches * B, "seconds": dt, "files_written": nfiles, "bytes_written": nbytes,
                "writer_threads": st.get("writer_threads"), "dir": "tmpfs/tmp (deleted)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def bf16_drift(a, G, synthetic, S, n_trans_rows):
    """How far the throughput mode (bf16) lands from the parity mode (fp32 HIP, pinned to the reference by the golden
    tests) on the SAME scenes, Philox keys and full transition table: depth in metres over in-painted pixels and the
    point-XYZ difference of the pixels both runs keep."""
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import MaskUnet, Unet
    n = a.drift_scenes
    idx = list(range(20_000_000, 20_000_000 + n))
    depth, K, pose = synthetic.synth_batch(a.seed, idx, S)
    dev = torch.device("cuda", torch.cuda.current_device())
    d_depth, d_K, d_pose = (torch.from_numpy(x).to(dev) for x in (depth, K, pose))
    seeds = [synthetic.noise_seed(a.seed, j) for j in idx]
    res = {}
    for dt in ("fp32", "bf16"):
        unet = Unet(a.dim, dtype=dt).init_synthetic(seed=1)
        mask = MaskUnet(a.dim, dtype=dt).init_synthetic(seed=2, final_bias=6.0)
        diff = GaussianDiffusion(unet, image_size=S, timesteps=a.timesteps, sampling_timesteps=a.sampling_steps)
        rpj, hit = G.reproject_tensor(d_depth, d_K, d_pose, clip=(0, 10), depth_unit=10.0, out_scale=0.1)
        _, hit_c, cond = G.apply_mask(mask(rpj), rpj, hit, 0.99)
        img = diff.sample(param_cond=G.param_vector(d_K), img_cond=cond, seeds=seeds)
        out, _, _ = G.apply_mask(mask(img), img, None, 0.99, want_cond=False)
        xyz, valid = G.unproject_f64(out, d_K, d_pose)
        res[dt] = (img.cpu(), hit_c.cpu(), xyz.cpu(), valid.cpu())
        diff.close(); unet.close(); mask.close()
    (i32, k32, x32, v32), (i16, k16, x16, v16) = res["fp32"], res["bf16"]
    same_known = bool(torch.equal(k32, k16))
    free = ~(k32 | k16)
    dd = (i32 - i16).abs()[free] * 10.0
    both = v32 & v16
    dx = (x32 - x16).abs().max(dim=-1).values[both]
    return {"scenes": n, "image_size": S, "transitions": n_trans_rows, "reference": "this library's fp32 parity mode, same Philox keys",
            "known_mask_identical": same_known, "inpainted_fraction": float(free.float().mean()),
            "inpainted_depth_m": {"max": float(dd.max()) if dd.numel() else 0.0, "mean": float(dd.mean()) if dd.numel() else 0.0,
                                  "median": float(dd.median()) if dd.numel() else 0.0},
            "xyz_m_points_kept_by_both": {"max": float(dx.max()) if dx.numel() else 0.0, "mean": float(dx.mean()) if dx.numel() else 0.0},
            "kept_by_only_one_fraction": float((v32 ^ v16).float().mean()),
            "saturated_fraction_fp32": float(((i32 <= 0) | (i32 >= 1)).float().mean()),
            "note": "synthetic (random) weights: most in-painted pixels end on the [-1,1] clamp in both modes (median 0); the "
                    "rest sit in a chain that amplifies perturbations (tests/golden/G12b: half-ulp perturbations of the network "
                    "output already move a 50-step result by 0.5-1.2e-4 m), so the maximum is not a precision statement",
            "known_pixels": "bit-identical to the condition in both modes (DDNM replacement)"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a HIP device (the product has no CPU path)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from pointreggpt_amd import geometry as G
    from pointreggpt_amd import synthetic
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import MaskUnet, Unet

    B, S = a.batch, a.size
    dev = torch.device("cuda", local)
    unet = Unet(a.dim, dtype=a.dtype).init_synthetic(seed=1)
    mask = None if a.sampler_only else MaskUnet(a.dim, dtype=a.dtype).init_synthetic(seed=2, final_bias=6.0)
    diff = GaussianDiffusion(unet, image_size=S, timesteps=a.timesteps, sampling_timesteps=a.sampling_steps)
    n_trans = len(diff.step_table())

    total_batches = a.warmup + a.steps
    # rank r owns scene indices r*total*B ... ; inputs are synthesised and uploaded BEFORE the clock starts
    batches = []
    for i in range(total_batches):
        first = (rank * total_batches + i) * B
        idx = list(range(first, first + B))
        depth, K, pose = synthetic.synth_batch(a.seed, idx, S)
        batches.append(dict(idx=idx, depth=torch.from_numpy(depth).to(dev), K=torch.from_numpy(K).to(dev),
                            pose=torch.from_numpy(pose).to(dev),
                            seeds=[synthetic.noise_seed(a.seed, j) for j in idx]))

    def one_batch(bt):
        pc = G.param_vector(bt["K"])
        if a.sampler_only:
            rpj, hit = G.reproject_tensor(bt["depth"], bt["K"], bt["pose"], clip=(0, 10), depth_unit=10.0, out_scale=0.1)
            _, _, cond = G.apply_mask(torch.ones_like(rpj), rpj, hit, 0.5)
            return diff.sample(param_cond=pc, img_cond=cond, seeds=bt["seeds"])
        rpj, hit = G.reproject_tensor(bt["depth"], bt["K"], bt["pose"], clip=(0, 10), depth_unit=10.0, out_scale=0.1)
        _, _, cond = G.apply_mask(mask(rpj), rpj, hit, 0.99)
        img = diff.sample(param_cond=pc, img_cond=cond, seeds=bt["seeds"])
        out, _, _ = G.apply_mask(mask(img), img, None, 0.99, want_cond=False)
        xyz, valid = G.unproject_f64(out, bt["K"], bt["pose"])
        return xyz

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_batch(batches[i])
    barrier()
    t0 = time.perf_counter()
    for i in range(a.warmup, total_batches):
        one_batch(batches[i])
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    pairs = world * a.steps * B
    value = pairs / dt
    tflop_pair = (n_trans * UNET_GFLOP.get(S, 58.976 * (S / 128) ** 2) +
                  (0 if a.sampler_only else 2 * MASK_GFLOP.get(S, 59.173 * (S / 128) ** 2))) / 1e3

    res = {
        "metric": "generated point-cloud pairs/sec (node), {0}x{0} depth, {1}-step {2}".format(
            S, n_trans, "DDNM" if not diff.is_ddim_sampling else "DDIM (DDNM replacement, eta=1)"),
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": ("configs[1]: DDNM p_sample_loop only" if a.sampler_only else
                                "configs[2]: full pipeline (SE(3) z-buffer reproject + MaskUnet + DDNM sampler + MaskUnet + f64 unproject)"),
                   "batch_per_gpu": B, "image_size": S, "transitions": n_trans,
                   "sampler": "ddim" if diff.is_ddim_sampling else "ancestral-ddnm", "unet_dim": a.dim,
                   "noise": "on-device Philox4x32-10 keyed per scene", "weights": "synthetic (deterministic initialiser)",
                   "parallelism": f"scene-sharded x{world}, no collectives",
                   "hipgraph": "one captured transition (U-Net + update), replayed per step",
                   "tflop_per_pair": tflop_pair},
        "end_to_end_mfma_frac": value / world * tflop_pair / MFMA_PEAK_TFLOPS[a.dtype],
    }

    if rank == 0 and not a.no_roofline:
        # dominant kernel = conv_igemm_kernel (implicit-GEMM convolution).  Its launches are timed live with HIP events
        # on the sampler's own stream over `profile_transitions` transitions of the same workload (eager launches).
        nprof = min(a.profile_transitions, n_trans)
        pdiff = GaussianDiffusion(unet, image_size=S, timesteps=a.timesteps, sampling_timesteps=a.sampling_steps)
        rows = pdiff.step_table()[:nprof]
        pdiff.step_table = lambda: rows
        bt = batches[-1]
        pc = G.param_vector(bt["K"])
        rpj, hit = G.reproject_tensor(bt["depth"], bt["K"], bt["pose"], clip=(0, 10), depth_unit=10.0, out_scale=0.1)
        _, _, cond = G.apply_mask(torch.ones_like(rpj), rpj, hit, 0.5)
        pdiff.sample(param_cond=pc, img_cond=cond, seeds=bt["seeds"], profile=True)
        torch.cuda.synchronize()
        pr = pdiff.last_profile(B)
        ach = pr["conv_flops"] / (pr["conv_ms"] * 1e-3) / 1e12
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "conv_hbm_traffic.json")
        if os.path.exists(tpath) and B == 64 and S == 128 and a.dtype == "bf16":
            # HBM bytes per conv launch from the committed rocprofv3 PMC passes of this same workload (bench.py cannot
            # run the profiler on itself); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.  The file
            # records the hash of the conv kernel sources it was measured on: a stale measurement is refused (null).
            tj = json.load(open(tpath))
            if tj.get("kernel_sources_sha256") == conv_sources_hash():
                traffic, traffic_src = tj["hbm_bytes_per_launch"], "profiles/conv_hbm_traffic.json (sources hash matches)"
            else:
                traffic_src = "profiles/conv_hbm_traffic.json is stale (conv kernel sources changed since the PMC passes): not reported"
        res["roofline"] = {
            "kernel": "MFMA convolutions: conv3x3_ws_kernel (3x3, wave-specialised persistent) + conv_igemm_kernel (1x1 / 4x4s2)",
            "bound": "mfma", "achieved": ach,
            "peak": MFMA_PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS[a.dtype],
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (FETCH_SIZE*2 + WRITE_SIZE)", "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": pr.get("conv_bytes", 0.0) / max(1, pr["conv_launches"]),
            "peak_measured_random_operands": 1670.0 if a.dtype == "bf16" else None,   # tools/micro/mfma_power.hip, power-limited
            "launches": pr["conv_launches"], "avg_launch_us": pr["conv_ms"] * 1e3 / max(1, pr["conv_launches"]),
            "flop_per_launch": pr["conv_flops"] / max(1, pr["conv_launches"]),
            "share_of_step_time": pr["conv_ms"] / pr["total_ms"],
            "measured": f"HIP events around every conv launch, {nprof} transitions, batch {B}",
        }
        if not a.sampler_only:
            res["roofline_mem"] = mem_rooflines(G, bt, S, B, pr)
        pdiff.close()
    if not a.no_e2e_files and not a.sampler_only:
        # every rank runs its own shard of the generate_dataset loop; aggregate like the headline metric
        e2e = e2e_files(a, unet, mask, diff, rank, world, B, S)
        dt_e = e2e["seconds"]
        if dist is not None:
            tt = torch.tensor([dt_e], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_e = float(tt.item())
        e2e.update(value=world * e2e["pairs"] / dt_e, unit="pairs/s on disk (whole job)", seconds_max_over_ranks=dt_e,
                   vs_device_only=world * e2e["pairs"] / dt_e / value,
                   what="Generator.generate --synthetic: memory-cloud z-buffer + MaskUnet + sampler + MaskUnet + unprojection + "
                        "crop / 0.025 voxel grid / PLY + PNG + text files through the C++ writer pool, overlapped with the next batch")
        res["e2e_files"] = e2e
    if rank == 0 and not a.no_drift and a.dtype == "bf16" and not a.sampler_only:
        res["bf16_drift"] = bf16_drift(a, G, synthetic, S, n_trans)
    if rank == 0 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(S, a.dim)

    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
