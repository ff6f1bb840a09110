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
# dense, MI355X_MICROARCH.md chip table.  f16x3: the f16 matrix pipe (2.5 PFLOP/s) executes THREE MFMA FLOPs per algorithmic
# FLOP (hi*hi + hi*lo + lo*hi), so algorithmic TFLOP/s are priced against 2500 / 3
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "mxfp8": 5000.0, "f16x3": 2500.0 / 3.0}
CONV_CLASS = ("MFMA convolutions: conv3x3_w256_kernel (256-pixel x 128-channel tiles, also Downsample) + conv3x3_c64_kernel "
              "(64 -> 64, weights-stationary) + conv3x3_ws_kernel (128-pixel wave-specialised tiles) + conv_igemm_kernel (1x1); "
              "the second conv of every ResnetBlock reads an f16 tensor through a packed-f16 GroupNorm+SiLU prologue and contracts "
              "f16 operands (v_mfma_f32_32x32x16_f16, the bf16 instruction's rate; 'h16', DESIGN 4.7); Upsample convs as four 2x2-tap "
              "sub-pixel convolutions (4/9 of the MACs, DESIGN 4.8)")


def conv_sources_hash():
    import hashlib
    h = hashlib.sha256()
    for f in ("conv_ws.hip", "conv_c64.hip", "conv_w256.hip", "conv.hip", "conv.h", "common.h"):
        h.update(open(os.path.join(ROOT, "pointreggpt_amd", "csrc", f), "rb").read())
    return h.hexdigest()


EVIDENCE = {"profiles": "profiles/README.md lists every file with its command and a one-line reading: rocprofv3 --kernel-trace --stats "
                        "summaries per precision mode, the PMC passes, the per-launch listing of one evaluation, the driver-command "
                        "bench line + sidecar, the GPU test log",
            "ab_records": "profiles/*_ab_* compare alternating runs of the same binary on the same box (box to box the same binary "
                          "spreads +-4 %)",
            "power": "profiles/r06_power_clock_mfma_busy_per_mode.json (socket power, shader clock, pairs per joule, MFMA-busy per mode)",
            "mfma_sustained": "profiles/r06_mfma_sustained_rates_micro.txt (register-only MFMA loops at the power cap: bf16 1.82-1.86 PFLOP/s on random "
                              "operands, f16 1.67-1.70, constant operands 2.49)"}
LINE_LIMIT = 4096          # bytes of the contract line (the driver keeps a bounded stdout tail and parses its last line)
SIDECAR = "bench_full.json"


def write_sidecar(res):
    """Everything bench.py measured (per-kernel tables, per-rank rows, drift chains, memory-bound rooflines, prose) as one JSON
    document next to bench.py, and in gpurun_out/ when that directory exists (gpurun merges it back); returns the path written."""
    path = os.path.join(ROOT, SIDECAR)
    txt = json.dumps(res, indent=1)
    for p in (path, os.path.join(ROOT, "gpurun_out", SIDECAR)):
        try:
            if os.path.isdir(os.path.dirname(p)):
                with open(p, "w") as f:
                    f.write(txt + "\n")
        except OSError as e:          # a read-only checkout must not cost the contract line
            print(f"bench.py: could not write {p}: {e}", file=sys.stderr)
    return SIDECAR


def _r(x, n=4):
    """Round floats to n significant digits for the contract line (the sidecar keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    if isinstance(x, str) and len(x) > 200:
        return x[:197] + "..."
    return x


def contract_line(res, sidecar):
    """The ONE stdout line: the contract's scalar keys, `config`, a short `roofline` (dominant kernel class), `cpu_baseline`,
    and compact `parity_mode` / `configs4` / `e2e_files` objects.  Built key by key (never by pruning the full document), and
    checked against LINE_LIMIT so that no future leg can push it past what the driver parses."""
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data")}
    c = res["config"]
    out["config"] = {k: _r(c[k]) for k in ("workload", "batch_per_gpu", "image_size", "transitions", "sampler", "unet_dim", "streams",
                                           "parallelism", "hipgraph", "tflop_per_pair") if k in c}
    out["end_to_end_mfma_frac"] = _r(res.get("end_to_end_mfma_frac"))
    rf = res.get("roofline")
    if rf:
        out["roofline"] = {"kernel": "MFMA conv class (conv3x3_w256 + conv3x3_c64 + conv3x3_ws + conv_igemm), HIP events per launch"}
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "launches", "avg_launch_us",
                  "executed_frac", "share_of_step_time"):
            out["roofline"][k] = _r(rf.get(k), 5)
        out["roofline"]["frac"] = rf["frac"]              # exactly achieved / peak of the full-precision values
        out["roofline"]["achieved"] = rf["achieved"]
    cb = res.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": f"oracle p_sample (torch-CPU fp32, {cb['cores']} threads = fastest of a sweep), batch 4, 12 timed transitions x1000"}
        hf = cb.get("host_filled")
        if isinstance(hf, dict) and "value" in hf:
            out["cpu_baseline"]["host_filled"] = {"value": _r(hf["value"]), "cores": hf["cores"]}
    pm = res.get("parity_mode")
    if pm:
        tol = pm.get("tolerance", {})
        o = {"north_star_xyz_m": 1e-4}
        for key, src in (("fp32", "fp32"), ("f16x3", "f16x3"), ("f16x3_256", "f16x3_256_ddim250")):
            leg = pm.get(src)
            if not leg:
                continue
            o[key] = {"pairs_per_s": _r(leg["pairs_per_s"]), "lanes": leg.get("streams"),
                      "frac": _r(leg.get("roofline", {}).get("frac")), "worst_xyz_m": _r(tol.get("worst_xyz_m", {}).get(key), 3)}
            if "one_lane" in leg:
                o[key]["one_lane_pairs_per_s"] = _r(leg["one_lane"]["pairs_per_s"])
        for k in ("f16x3_vs_fp32_one_lane", "headline_vs_fp32"):
            if k in pm:
                o[k] = _r(pm[k])
        out["parity_mode"] = o
    c4 = res.get("configs4")
    if c4:
        out["configs4"] = {"value": _r(c4["value"]), "unit": c4["unit"], "dtype": c4["dtype"], "batch": c4["config"]["batch_per_gpu"],
                           "frac": _r(c4.get("roofline", {}).get("frac")), "peak": c4.get("roofline", {}).get("peak"),
                           "bf16_same_shape": _r(c4.get("bf16_same_shape", {}).get("value")),
                           "mxfp8_over_bf16": _r(c4.get("mxfp8_over_bf16_same_shape")),
                           "mx_flop_fraction": _r(c4.get("mx_flop_fraction")), "mx_cap_measured": c4.get("mx_cap_measured")}
    e = res.get("e2e_files")
    if e:
        out["e2e_files"] = {"value": _r(e["value"]), "unit": "pairs/s on disk", "vs_device_only": _r(e["vs_device_only"]),
                            "gt_log_pairs_per_s": _r(e.get("gt_log", {}).get("pairs_per_s"))}
    rm = res.get("roofline_mem")
    if rm and "streaming" in rm:
        out["roofline_mem"] = {k: _r(v.get("frac_of_8TBps"), 3) for k, v in rm["streaming"].get("kernels", {}).items()}
    if "per_rank" in res:
        pr = res["per_rank"]
        out["per_rank_timed_s"] = {"min": _r(min(p["timed_s"] for p in pr)), "max": _r(max(p["timed_s"] for p in pr))}
    out["full"] = sidecar
    line = json.dumps(out, separators=(",", ":"))
    # last resort, in the order the driver needs the keys least (never reached with today's legs: asserted by the tests)
    for k in ("roofline_mem", "e2e_files", "per_rank_timed_s", "configs4", "parity_mode"):
        if len(line) < LINE_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) < LINE_LIMIT, len(line)
    return line


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4, help="timed batches per rank")
    p.add_argument("--warmup", type=int, default=2, help="untimed batches per rank")
    p.add_argument("--batch", type=int, default=64)
    p.add_argument("--streams", type=int, default=2, help="independent pipelines (HIP stream + host thread each) sharing the GPU")
    p.add_argument("--dynamic-lanes", action="store_true", help="lanes pull the timed batches from a shared counter instead of owning every n-th batch (for "
                   "lane counts that do not divide the step count; measured equal within noise at 2 / 3 / 4 lanes: profiles/r06_lanes_sweep_bf16.txt)")
    p.add_argument("--size", type=int, default=128)
    p.add_argument("--timesteps", type=int, default=1000)
    p.add_argument("--sampling-steps", type=int, default=None, help="< timesteps selects DDIM (default: ancestral DDNM)")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "mxfp8", "f16x3"])
    p.add_argument("--dim", type=int, default=64)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--sampler-only", action="store_true", help="configs[1]: p_sample_loop only (no geometry / MaskUnet)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--profile-transitions", type=int, default=40)
    p.add_argument("--no-e2e-files", action="store_true", help="skip the generate_dataset leg (files on disk)")
    p.add_argument("--e2e-batches", type=int, default=4)
    p.add_argument("--no-drift", action="store_true", help="skip the drift-vs-reference legs (fixtures G19 / G20)")
    p.add_argument("--no-configs4", action="store_true", help="skip the configs[4] leg (256x256, 250-step DDIM, mxfp8)")
    p.add_argument("--c4-batch", type=int, default=16)
    p.add_argument("--c4-steps", type=int, default=3, help="timed batches of the configs[4] leg")
    p.add_argument("--uncalibrated", action="store_true", help="round-1/2 synthetic weights (98 %% of in-painted pixels saturate)")
    p.add_argument("--no-parity-mode", action="store_true", help="skip the parity_mode legs (fp32 and f16x3 at the headline shape)")
    p.add_argument("--parity-transitions", type=int, default=100, help="timed ancestral transitions of the parity_mode legs (extrapolated to 1000)")
    p.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)   # internal: one host_filled worker with N threads
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
    res = {"value": pairs_per_s, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": f"oracle p_sample (torch-CPU fp32, {used} threads = fastest of a sweep on a {os.cpu_count()}-core host), "
                     f"batch {B}, {size}x{size}, {n} timed transitions = {dt:.3f} s/transition, extrapolated x{T} "
                     f"transitions; MaskUnet/geometry (0.2% of the work) omitted"}
    res["host_filled"] = cpu_host_filled(size, dim)
    return res


def _cpu_worker(threads, size, dim):
    """One host_filled worker: `threads` oneDNN threads, batch 4, 1 warm-up + 4 timed transitions; prints seconds per transition."""
    from oracle import diffusion as OD
    from oracle import unet as OU
    from pointreggpt_amd import weights as W
    torch.set_num_threads(threads)
    B = 4
    p = W.synth_state_dict(W.unet_config(dim), 0)
    sch = OD.schedule(1000)
    den = lambda x, t, c: OU.unet_forward(p, x, t, c)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B, 1, size, size), generator=g)
    pc = torch.tensor([[151.5, 152.1, size / 2 + 0.5, size / 2]] * B)
    cond = torch.cat([torch.rand((B, 1, size, size), generator=g) * 2 - 1, (torch.rand((B, 1, size, size), generator=g) > 0.5).float() * 2 - 1], 1)
    nz = torch.randn((B, 1, size, size), generator=g)
    OD.p_sample(sch, den, x, 999, pc, cond, nz)
    t0 = time.perf_counter()
    for i in range(4):
        OD.p_sample(sch, den, x, 998 - i, pc, cond, nz)
    print("CPU_WORKER_SECONDS_PER_TRANSITION %.6f" % ((time.perf_counter() - t0) / 4), flush=True)


def cpu_host_filled(size, dim):
    """The same oracle transition with the WHOLE host busy: cpu_count // 16 concurrent 16-thread workers (one oneDNN process does
    not scale past ~16 threads at this batch), aggregate pairs/s (round-3 VERDICT item 9)."""
    import subprocess
    cores = os.cpu_count() or 1
    threads = min(16, cores)
    nw = max(1, cores // threads)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(threads), "--size", str(size), "--dim", str(dim)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for _ in range(nw)]
    dts = []
    for pr in procs:
        out, _ = pr.communicate(timeout=600)
        for line in out.splitlines():
            if line.startswith("CPU_WORKER_SECONDS_PER_TRANSITION"):
                dts.append(float(line.split()[1]))
    if len(dts) != nw:
        return {"error": f"{nw - len(dts)} of {nw} workers failed"}
    agg = sum(4 / (dt * 1000) for dt in dts)
    return {"value": agg, "unit": "pairs/s", "cores": nw * threads, "workers": nw, "threads_per_worker": threads,
            "seconds_per_transition_mean": float(np.mean(dts)), "wall_seconds": time.perf_counter() - t0,
            "sample": f"{nw} concurrent oracle workers x {threads} threads, batch 4 each, {size}x{size}, 4 timed transitions after 1 "
                      f"warm-up, extrapolated x1000 transitions"}


def per_kernel_table(pr, peak_tflops, transitions, top=8):
    """The per-shape table of the profiled conv launches (prg_sampler_get_profile_shapes: HIP events around every launch inside
    the library): the `top` shapes by time, each with launches per evaluation, microseconds per launch, algorithmic GFLOP per
    launch and the fraction of the MFMA peak it runs at (evidence item 6)."""
    rows = sorted(pr.get("conv_shapes", []), key=lambda r: -r["ms"])
    tot = sum(r["ms"] for r in rows) or 1.0
    out = []
    for r in rows[:top]:
        us = r["ms"] * 1e3 / max(1, r["launches"])
        gf = r["flops"] / max(1, r["launches"]) / 1e9
        name = "{}{}x{} {}->{} @{}x{}{}{}".format("up+" if r["ups"] else "", r["k"], r["k"], r["cin"], r["cout"], r["hout"], r["wout"],
                                                  " s2" if r["stride"] == 2 else "", (" two-source" if r["two_source"] else "") + (" +prologue" if r["prologue"] else ""))
        out.append({"conv": name, "launches_per_evaluation": r["launches"] / max(1, transitions), "avg_us": us, "gflop_per_launch": gf,
                    "tflops": gf / us * 1e3 if us else None, "frac": gf / us * 1e3 / peak_tflops if us else None,
                    "executed_over_algorithmic": r["flops_executed"] / r["flops"] if r["flops"] else None, "share_of_conv_time": r["ms"] / tot})
    return out


def _mem_cases(lib, _lib, depth, K, pose, B, S, reps):
    """HIP-event microseconds per launch of the memory-bound kernels on a (B, 1, S, S) batch: `reps` back-to-back C-ABI calls into
    preallocated outputs on torch's current stream (the stream they are launched on)."""
    import ctypes as C
    dev = depth.device
    npx = B * S * S
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
        "reproject_zbuffer": (9, lambda: lib.prg_reproject_zbuffer(P(depth), P(K), P(pose), P(rpj), P(hit), B, S, S, 10.0, 0.0, 10.0, 0.1, st)),
        "unproject_f64": (4 + 24 + 1, lambda: lib.prg_unproject_f64(P(rpj), P(K), P(pose), P(xyz), P(valid), B, S, S, 10.0, 0.5, 10.0, st)),
        "depth_augment": (4 + 12, lambda: lib.prg_depth_augment(P(rpj), P(aug), B, S, S, st)),
        "apply_mask": (4 + 4 + 1 + 4 + 1 + 8, lambda: lib.prg_apply_mask(P(prob), P(rpj), P(hit), 0.5, P(d2), P(h2), P(cond), B, S, S, st)),
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
        out[name] = {"bytes_per_px": bpp, "avg_us": us, "GBps": bpp * npx / (us * 1e-6) / 1e9}
    # atomicMin traffic of the z-buffer: one atomic per source pixel that is valid and lands in the frame (= at least the hit count)
    out["reproject_zbuffer"]["valid_source_px_fraction"] = float(((depth * 10.0 > 0.0) & (depth * 10.0 < 10.0)).float().mean())
    out["reproject_zbuffer"]["hit_px_fraction"] = float(hit.float().mean())
    # the sampler's transition update alone (prg_debug_sampler_step: x0, DDNM replace, posterior, Philox noise), ancestral row
    x = torch.randn((B, S * S), device=dev)
    u = torch.randn((B, S * S), device=dev)
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device=dev)
    row = _lib.StepC(t=500, clip_pred=2, c_x0=0.01, c_x=0.98, c_eps=0.0, sigma=0.05, sqrt_recip=1.2, sqrt_recipm1=0.7)
    us = C.c_float()
    _lib.check(lib.prg_debug_sampler_step(P(x), P(u), P(cond), P(seeds), C.byref(row), B, S * S, reps, C.byref(us), st))
    out["sampler_step"] = {"bytes_per_px": 20, "avg_us": us.value, "GBps": 20 * npx / (us.value * 1e-6) / 1e9}
    for v in out.values():
        v["frac_of_8TBps"] = v["GBps"] / 8000.0
    return out


def mem_rooflines(G, bt, S, B, pr, stream_batch=1024):
    """Bandwidth of the memory-bound kernels of one pair (north_star: 'coalesced HBM loads ... evidenced by HBM-GB/s'): algorithmic
    bytes per pixel (fp32 images, float64 points) / HIP-event duration, at two sizes: the benchmarked batch
    (1 Mpx per launch: microsecond launches, latency-bound) and `stream_batch` scenes (16 Mpx per launch: what the kernels stream at)."""
    from pointreggpt_amd import _lib, synthetic
    lib = _lib.load()
    small = _mem_cases(lib, _lib, bt["depth"].contiguous(), bt["K"].contiguous(), bt["pose"].contiguous(), B, S, 100)
    if pr.get("step_launches"):
        us = pr["step_ms"] * 1e3 / pr["step_launches"]
        small["sampler_step_in_chain"] = {"bytes_per_px": 20, "avg_us": us, "GBps": 20 * B * S * S / (us * 1e-6) / 1e9,
                                          "frac_of_8TBps": 20 * B * S * S / (us * 1e-6) / 1e9 / 8000.0,
                                          "note": "HIP events inside the library around every launch of the profiled transitions"}
    res = {"bound": "hbm", "peak_GBps": 8000.0, "achievable_GBps": 6300.0,
           "bytes_per_px": "reproject 4 in + 4 z-buffer + 1 mask; unproject_f64 4 in + 24 xyz + 1 valid; depth_augment 4 in + 12 out; "
                           "apply_mask 4 + 4 + 1 in, 4 + 1 + 8 out; sampler_step 4 x + 4 u + 8 cond in, 4 x out (noise generated on chip)",
           "benchmarked_batch": {"pixels_per_launch": B * S * S, "kernels": small,
                                 "reading": "1 Mpx per launch = 9-30 MB per call: microsecond launches, latency- not bandwidth-bound; "
                                            "together < 0.01 % of a 1000-step pair"}}
    nb = int(stream_batch)
    reps = (nb + B - 1) // B
    cat = lambda t: t.repeat(reps, *([1] * (t.dim() - 1)))[:nb].contiguous()
    big = _mem_cases(lib, _lib, cat(bt["depth"]), cat(bt["K"]), cat(bt["pose"]), nb, S, 20)
    res["streaming"] = {"pixels_per_launch": nb * S * S, "kernels": big,
                        "reading": f"{nb} scenes per launch ({nb * S * S / 2**20:.0f} Mpx): the same kernels streaming; fractions are of the 8 TB/s "
                                   "peak (MI355X_MICROARCH.md gives ~6.3 TB/s as the achievable copy rate)"}
    return res


def e2e_files(a, unet, mask, diff, rank, world, B, S, lanes=None):
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
        nl = 1 + len(lanes or [])
        # warm-up: one batch per lane (graph capture, pool start)
        gen.generate(first - (nl - 1) * B, first + B, 1, depth_correction=mask, stats=st, lanes=lanes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gen.generate(first + B, first + (a.e2e_batches + 1) * B, 1, depth_correction=mask, stats=st, lanes=lanes)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nfiles = sum(len(f) for _d, _s, f in os.walk(os.path.join(root, "data")))
        nbytes = sum(os.path.getsize(os.path.join(d, f)) for d, _s, fs in os.walk(os.path.join(root, "data")) for f in fs)
        # generate_gt.py on the same scenes (generate_gt.py:105-188): read both PLYs, 0.025 voxel grids (C++), all pairs in one
        # prg_overlap_counts launch, per-scene gt.log, metadata/gt.log
        from pointreggpt_amd.generator import gather_gt, generate_gt
        t1 = time.perf_counter()
        generate_gt("", first + B, first + (a.e2e_batches + 1) * B, 2, root=root)
        gather_gt("", first + B, first + (a.e2e_batches + 1) * B, root=root)
        dt_gt = time.perf_counter() - t1
        gt_path = os.path.join(root, "metadata", "gt.log")
        n_lines = sum(1 for _ in open(gt_path)) if os.path.exists(gt_path) else 0
        return {"pairs": a.e2e_batches * B, "seconds": dt, "files_written": nfiles, "bytes_written": nbytes,
                "writer_threads": st.get("writer_threads"), "lanes": st.get("lanes"), "dir": "tmpfs/tmp (deleted)",
                "gt_log": {"seconds": dt_gt, "pairs_per_s": a.e2e_batches * B / dt_gt, "lines": n_lines,
                           "what": "generate_gt + gather_gt over the timed scenes, after generation (not overlapped)"}}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def drift_vs_reference(dtypes, dim):
    """Distance of the library's precision modes from the REAL reference on chains of real length: the committed fixtures
    G20 (250-step DDIM @128x128), G19 (1000-step ancestral @64x64) and G21 (250-step DDIM @256x256: configs[4]'s chain), all
    produced by the reference itself on the
    calibrated synthetic denoiser (tools/make_goldens.py), re-run here through the C-ABI on the fixtures' stored condition
    and regenerated noise.  Metres (normalised depth x 10); in-painted pixels only (known pixels are bit-exact)."""
    import hashlib
    from pointreggpt_amd import geometry as G
    from pointreggpt_amd import weights as W
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import Unet
    gold = os.path.join(ROOT, "tests", "golden")
    g0 = np.load(os.path.join(gold, "G0_host_tables.npz"))
    out = {}
    for name, S, steps, tab, rep_batch in (("G22_chain1000_ancestral_128", 128, None, "anc1000", 64), ("G20_ddim250_128", 128, 250, "ddim250", 64),
                                           ("G19_chain1000_ancestral_64", 64, None, "anc1000", 64), ("G21_ddim250_256", 256, 250, "ddim250", 16),
                                           ("G21b_ddim250_256", 256, 250, "ddim250", 16)):
        path = os.path.join(gold, name + ".npz")
        if not os.path.exists(path):
            out[name] = "fixture missing"
            continue
        g = np.load(path)
        st = torch.random.get_rng_state()
        torch.manual_seed(int(g["noise_seed"]))
        nz = torch.stack([torch.randn((1, 1, S, S)) for _ in range(int(g["n_draws"]))])
        torch.random.set_rng_state(st)
        nz = nz.reshape(-1, 1, 1, S, S)
        if hashlib.sha256(nz.numpy().tobytes()).digest() != bytes(bytearray(g["noise_sha256"].tolist())):
            out[name] = "torch.randn does not reproduce the fixture's noise on this build"
            continue
        known = (g["img_cond"][:, 1:2] + 1) * 0.5 > 0.5
        ref = g["sampled"]
        v_ref = ((ref[0, 0] * 10 > 0.5) & (ref[0, 0] * 10 < 10)).reshape(-1)
        res = {"chain": f"{steps or 1000}-step {'DDIM' if steps else 'ancestral DDNM'} @{S}x{S}, dim {dim}, calibrated synthetic weights; "
                        f"the fixture's scene replicated x{rep_batch} (benchmarked launch shapes), slot 0 compared",
               "reference_spread_1_vs_8_threads_xyz_m": float(g["xyz_spread_1_vs_8_threads_m"]),
               "reference_to_float64_twin_xyz_m": float(g["xyz_ref_to_exact_m"]),
               "inpainted_fraction": float(g["inpainted_fraction"]),
               "saturated_fraction_reference": float(g["saturated_fraction_inpainted"])}
        for dt in dtypes:
            net = Unet(dim, dtype=dt).load_state_dict(W.synth_state_dict(W.unet_config(dim), int(g["wseed"]), calibrated=True))
            net.set_time_freqs(g0[f"freqs_dim{dim}"])
            d = GaussianDiffusion(net, image_size=S, timesteps=1000, sampling_timesteps=steps)
            rows = d.step_table()
            for r, v in zip(rows, g0[tab + "_rows"]):      # the fixtures' host's float32 coefficient table
                for j, k in enumerate(("c_x0", "c_x", "c_eps", "sigma", "sqrt_recip", "sqrt_recipm1")):
                    r[k] = float(v[j])
            d.step_table = lambda rows=rows: rows
            # the scene replicated over the benchmarked batch: the launches then have the benchmarked shapes and take the
            # benchmarked kernels (the 256-pixel / MX kernels only run where a launch fills the chip); slot 0 is compared
            rb = rep_batch if dt in ("bf16", "mxfp8") else min(8, rep_batch)     # the float32-storage modes: 8 slots (time)
            nzb = nz.cuda().expand(-1, rb, -1, -1, -1).contiguous()
            img_b = d.sample(param_cond=torch.from_numpy(g["pc"]).cuda().repeat(rb, 1),
                             img_cond=torch.from_numpy(g["img_cond"]).cuda().repeat(rb, 1, 1, 1), noise=nzb)
            slot_invariant = bool(torch.equal(img_b[0], img_b[rb - 1]))
            img_d = img_b[:1].contiguous()
            del nzb, img_b
            cloud = G.point_clouds(img_d, torch.from_numpy(g["K"]).cuda(), torch.from_numpy(g["pose"]).cuda())[0]
            img = img_d.cpu().numpy()
            dd = np.abs(img.astype(np.float64) - ref)[~known] * 10.0
            v_hip = ((img[0, 0] * 10 > 0.5) & (img[0, 0] * 10 < 10)).reshape(-1)
            ch = np.full((v_hip.size, 3), np.nan); ch[v_hip] = cloud
            cr = np.full((v_ref.size, 3), np.nan); cr[v_ref] = g["cloud"]
            both = v_hip & v_ref
            res[dt] = {"known_pixels_bit_exact": bool(np.array_equal(img[known], ref[known])), "batch_slot_invariant": slot_invariant,
                       "inpainted_depth_m": {"max": float(dd.max()), "mean": float(dd.mean()), "median": float(np.median(dd))},
                       "xyz_linf_m": float(np.abs(ch[both] - cr[both]).max()),
                       "valid_mask_identical": bool(np.array_equal(v_hip, v_ref)),
                       "saturated_fraction": float(((img <= 0) | (img >= 1))[~known].mean())}
            d.close(); net.close()
        out[name] = res
    return out


def configs4_leg(a, G, synthetic, rank, dt="mxfp8"):
    """BASELINE configs[4] on one GPU: 256x256 depth, 250-step DDIM (eta = 1, DDNM replacement: the setting
    generate_dataset.py:34-49 ships), MX-fp8 operands for the 3x3 convolutions (v_mfma_scale_f32_32x32x64_f8f6f4), full
    pipeline, B = 16 (= 64 x 128x128 pixels per batch).  Own roofline against the 5 PFLOP/s dense MX-fp8 peak."""
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import MaskUnet, Unet
    B, S, T, steps = a.c4_batch, 256, 1000, 250
    dev = torch.device("cuda", torch.cuda.current_device())
    unet = Unet(a.dim, dtype=dt).init_synthetic(seed=1, calibrated=True)
    mask = MaskUnet(a.dim, dtype=dt).init_synthetic(seed=2, calibrated=True)
    diff = GaussianDiffusion(unet, image_size=S, timesteps=T, sampling_timesteps=steps)
    n_trans = len(diff.step_table())
    batches = []
    for i in range(1 + a.c4_steps):
        first = 30_000_000 + (rank * (1 + a.c4_steps) + i) * B
        idx = list(range(first, first + B))
        depth, K, pose = synthetic.synth_batch(a.seed, idx, S)
        batches.append(dict(depth=torch.from_numpy(depth).to(dev), K=torch.from_numpy(K).to(dev), pose=torch.from_numpy(pose).to(dev),
                            seeds=[synthetic.noise_seed(a.seed, j) for j in idx]))

    def one(bt):
        rpj, hit = G.reproject_tensor(bt["depth"], bt["K"], bt["pose"], clip=(0, 10), depth_unit=10.0, out_scale=0.1)
        _, hit_c, cond = G.apply_mask(mask(rpj), rpj, hit, 0.99)
        img = diff.sample(param_cond=G.param_vector(bt["K"]), img_cond=cond, seeds=bt["seeds"])
        out, _, _ = G.apply_mask(mask(img), img, None, 0.99, want_cond=False)
        return G.unproject_f64(out, bt["K"], bt["pose"]), img, hit_c

    one(batches[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for bt in batches[1:]:
        _, img, hit_c = one(bt)
    torch.cuda.synchronize()
    dtm = time.perf_counter() - t0
    value = a.c4_steps * B / dtm
    tflop_pair = (n_trans * UNET_GFLOP[S] + 2 * MASK_GFLOP[S]) / 1e3
    free = ~hit_c.bool()
    res = {"metric": f"generated point-cloud pairs/sec (one GPU), {S}x{S} depth, {n_trans}-step DDIM (DDNM replacement, eta=1)",
           "value": value, "unit": "pairs/s", "n_gpus": 1, "steps": a.c4_steps, "warmup": 1, "ms_per_step": dtm / a.c4_steps * 1e3,
           "dtype": dt, "data": "synthetic",
           "config": {"workload": "configs[4]: 256x256, 250-step DDIM, MX-fp8 U-Net operands, full pipeline" if dt == "mxfp8" else
                                  f"configs[4]'s shape in {dt} (same-shape comparator): 256x256, 250-step DDIM, full pipeline", "batch_per_gpu": B,
                      "image_size": S, "transitions": n_trans, "unet_dim": a.dim, "tflop_per_pair": tflop_pair,
                      "weights": "synthetic, calibrated head", "operand_format": ("OCP MX e4m3 + E8M0 per 32 channels on the 3x3 "
                      "convolutions with Cout % 128 == 0; 64-channel convolutions, attention and 1x1 convs in bf16") if dt == "mxfp8" else dt},
           "end_to_end_mfma_frac": value * tflop_pair / MFMA_PEAK_TFLOPS[dt],
           "saturated_fraction_inpainted": float(((img <= 0) | (img >= 1))[free].float().mean()) if bool(free.any()) else None}
    if not a.no_roofline:
        pdiff = GaussianDiffusion(unet, image_size=S, timesteps=T, sampling_timesteps=steps)
        nprof = min(20, n_trans)
        rows = pdiff.step_table()[:nprof]
        pdiff.step_table = lambda: rows
        bt = batches[-1]
        rpj, hit = G.reproject_tensor(bt["depth"], bt["K"], bt["pose"], clip=(0, 10), depth_unit=10.0, out_scale=0.1)
        _, _, cond = G.apply_mask(torch.ones_like(rpj), rpj, hit, 0.5)
        pdiff.sample(param_cond=G.param_vector(bt["K"]), img_cond=cond, seeds=bt["seeds"], profile=True)
        torch.cuda.synchronize()
        pr = pdiff.last_profile(B)
        ach = pr["conv_flops"] / (pr["conv_ms"] * 1e-3) / 1e12
        res["roofline"] = {"kernel": CONV_CLASS + (" (MX-fp8 operands on the w256mx launches)" if dt == "mxfp8" else ""), "bound": "mfma", "achieved": ach,
                           "peak": MFMA_PEAK_TFLOPS[dt], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS[dt], "traffic": None,
                           "launches": pr["conv_launches"], "avg_launch_us": pr["conv_ms"] * 1e3 / max(1, pr["conv_launches"]),
                           "flop_per_launch": pr["conv_flops"] / max(1, pr["conv_launches"]),
                           "share_of_step_time": pr["conv_ms"] / pr["total_ms"],
                           "measured": f"HIP events around every conv launch, {nprof} transitions, batch {B}"}
        if dt == "mxfp8":
            # share of the conv class's algorithmic FLOPs (and of its time) whose launches ran on the scale-MFMA with MX-fp8
            # operands; the rest (64-channel convs, 1x1 convs, sub-pixel Upsamples) ran on bf16 / f16 operands
            sh = pr.get("conv_shapes", [])
            res["mx_flop_fraction"] = sum(r["flops"] for r in sh if r["mx"]) / max(1.0, sum(r["flops"] for r in sh))
            res["mx_time_fraction"] = sum(r["ms"] for r in sh if r["mx"]) / max(1e-9, sum(r["ms"] for r in sh))
        pdiff.close()
    diff.close(); unet.close(); mask.close()
    return res


def parity_mode_leg(a, G, synthetic, rank, dt, shape=None, lanes=1):
    """The headline workload (full pipeline, B x S x S, ancestral DDNM) in a float32-storage precision mode: `fp32` = the parity
    mode (exact-f32 MFMA, float64 partial sums) and `f16x3` = the same storage and normalisation arithmetic with every
    convolution as three f16 MFMAs on hi/lo-split operands.  `parity_transitions` ancestral transitions are timed (the first
    rows of the 1000-step table: same per-transition work) and the sampler's share is extrapolated x(1000 / transitions).
    `lanes` > 1: that many independent pipelines (own handles, workspaces, graphs, one host thread + HIP stream each) run one timed
    batch each AT THE SAME TIME, as the headline leg's `--streams` does; pairs/s counts every lane's batch over the common wall time."""
    import threading
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import MaskUnet, Unet
    B, S, T = a.batch, a.size, a.timesteps
    steps = None
    if shape is not None:                      # (batch, image size, DDIM steps): the shipped setting, generate_dataset.py:34-49
        B, S, steps = shape
    dev = torch.device("cuda", torch.cuda.current_device())
    lanes = max(1, int(lanes))
    pipes = []
    for k in range(lanes):
        u = Unet(a.dim, dtype=dt).init_synthetic(seed=1, calibrated=True)
        m = MaskUnet(a.dim, dtype=dt).init_synthetic(seed=2, calibrated=True)
        d = GaussianDiffusion(u, image_size=S, timesteps=T, sampling_timesteps=steps)
        pipes.append(dict(unet=u, mask=m, diff=d, stream=torch.cuda.Stream(device=dev) if lanes > 1 else None,
                          ev=[torch.cuda.Event(enable_timing=True) for _ in range(2)]))
    unet, mask, diff = pipes[0]["unet"], pipes[0]["mask"], pipes[0]["diff"]
    full = len(diff.step_table())
    nt = min(a.parity_transitions, full)
    rows = diff.step_table()[:nt]
    for pp in pipes:
        pp["diff"].step_table = lambda: rows
    batches = []
    for i in range(2 * lanes):
        first = 40_000_000 + (rank * 2 * lanes + i) * B
        idx = list(range(first, first + B))
        depth, K, pose = synthetic.synth_batch(a.seed, idx, S)
        batches.append(dict(depth=torch.from_numpy(depth).to(dev), K=torch.from_numpy(K).to(dev), pose=torch.from_numpy(pose).to(dev),
                            seeds=[synthetic.noise_seed(a.seed, j) for j in idx]))

    def one(bt, pp, timed=False):
        rpj, hit = G.reproject_tensor(bt["depth"], bt["K"], bt["pose"], clip=(0, 10), depth_unit=10.0, out_scale=0.1)
        _, _, cond = G.apply_mask(pp["mask"](rpj), rpj, hit, 0.99)
        if timed:
            pp["ev"][0].record()
        img = pp["diff"].sample(param_cond=G.param_vector(bt["K"]), img_cond=cond, seeds=bt["seeds"])
        if timed:
            pp["ev"][1].record()
        out, _, _ = G.apply_mask(pp["mask"](img), img, None, 0.99, want_cond=False)
        return G.unproject_f64(out, bt["K"], bt["pose"])

    def run(first, timed):
        if lanes == 1:
            one(batches[first], pipes[0], timed)
            return
        errs = []

        def worker(k):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(pipes[k]["stream"]):
                    one(batches[first + k], pipes[k], timed)
            except BaseException as e:      # noqa: BLE001 — re-raised on the main thread
                errs.append(e)

        th = [threading.Thread(target=worker, args=(k,)) for k in range(lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    run(0, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(lanes, True)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    t_s = max(pp["ev"][0].elapsed_time(pp["ev"][1]) for pp in pipes) * 1e-3      # (the lanes' samplers run side by side)
    t_full = (t_all - t_s) + t_s * (full / nt)
    tflop_pair = (full * UNET_GFLOP.get(S, 58.976 * (S / 128) ** 2) + 2 * MASK_GFLOP.get(S, 59.173 * (S / 128) ** 2)) / 1e3
    res = {"dtype": dt, "pairs_per_s": lanes * B / t_full, "unit": "pairs/s", "batch": B, "image_size": S,
           "sampler": "ddim" if steps else "ancestral-ddnm",
           "timed_transitions": nt, "extrapolated_to": full, "seconds_timed": t_all, "seconds_sampler_timed": t_s,
           "ms_per_transition": t_s / nt * 1e3 / lanes, "streams": lanes, "tflop_per_pair": tflop_pair,
           "how": f"full pipeline on {lanes} lane(s) (one batch of {B} each, side by side), one warm-up batch per lane, one timed batch "
                  f"per lane of {nt} transitions (hipGraph replay); pairs/s = lanes x B / (non-sampler time + sampler time x {full}/{nt}); "
                  f"ms_per_transition = per batch-transition of the whole device (sampler time / transitions / lanes)"}
    if not a.no_roofline:
        npr = min(10, nt)
        prow = rows[:npr]
        pdiff = GaussianDiffusion(unet, image_size=S, timesteps=T, sampling_timesteps=steps)
        pdiff.step_table = lambda: prow
        bt = batches[1]
        rpj, hit = G.reproject_tensor(bt["depth"], bt["K"], bt["pose"], clip=(0, 10), depth_unit=10.0, out_scale=0.1)
        _, _, cond = G.apply_mask(torch.ones_like(rpj), rpj, hit, 0.5)
        pdiff.sample(param_cond=G.param_vector(bt["K"]), img_cond=cond, seeds=bt["seeds"], profile=True)
        torch.cuda.synchronize()
        pr = pdiff.last_profile(B)
        ach = pr["conv_flops"] / (pr["conv_ms"] * 1e-3) / 1e12
        res["roofline"] = {"kernel": ("conv3x3_halo_kernel<float> + conv_igemm_kernel<float> (v_mfma_f32_32x32x2_f32, float64-summed partials)" if dt == "fp32"
                                      else "conv3x3_split_ws_kernel (Cout % 128 = 0) + conv3x3_split_p64_kernel (Cout = 64, persistent) + conv_igemm_split_kernel (1x1, 4x4/s2): "
                                           "3 x v_mfma_f32_32x32x16_f16 per product tile on hi/lo-split operands"),
                           "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS[dt], "unit": "TFLOP/s (algorithmic)",
                           "frac": ach / MFMA_PEAK_TFLOPS[dt], "traffic": None, "launches": pr["conv_launches"],
                           "avg_launch_us": pr["conv_ms"] * 1e3 / max(1, pr["conv_launches"]),
                           "share_of_step_time": pr["conv_ms"] / pr["total_ms"],
                           "measured": f"HIP events around every conv launch, {npr} transitions, batch {B}",
                           "peak_note": ("157.3 TFLOP/s dense f32 MFMA" if dt == "fp32" else
                                         "2500 / 3 TFLOP/s: the f16 pipe executes three MFMA FLOPs per algorithmic FLOP")}
        pdiff.close()
    for pp in pipes:
        pp["diff"].close(); pp["unet"].close(); pp["mask"].close()
    return res


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU, LOCAL_RANK = k) on 127.0.0.1 and
    wait for them; rank 0 inherits stdout and prints the JSON line.  Fails loudly when the box has fewer devices than ranks
    (PRG_BENCH_BACKEND=gloo is the rehearsal mode: ranks share devices, barrier / MAX on the host)."""
    import socket
    import subprocess
    n = a.gpus
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < n and os.environ.get("PRG_BENCH_BACKEND", "nccl") != "gloo":
        print(f"bench.py --gpus {n}: only {ndev} HIP device(s) visible (set PRG_BENCH_BACKEND=gloo to rehearse with shared devices)", file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for k in range(n):
        env = dict(os.environ, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if k == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    sys.exit(next((rc for rc in rcs if rc), 0))


def main():
    a = parse()
    if a.cpu_worker:
        _cpu_worker(a.cpu_worker, a.size, a.dim)
        return
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, a.gpus):
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a HIP device (the product has no CPU path)", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU (the driver's launch).  PRG_BENCH_BACKEND=gloo is a rehearsal aid for boxes with fewer GPUs than ranks:
    # the ranks then share devices (local rank modulo the device count) and the barrier / MAX reduction run on the host.
    backend = os.environ.get("PRG_BENCH_BACKEND", "nccl")
    local_rank = local
    local = local % torch.cuda.device_count() if backend == "gloo" else local
    t_setup0 = time.perf_counter()
    # CPU placement (round 5): this rank's lane threads and writer pool stay on the cores of its GPU's NUMA node (its share of
    # them); done before any thread exists so that everything started later inherits the mask.  PRG_NO_AFFINITY=1 opts out.
    from pointreggpt_amd import sharding
    affinity = sharding.pin_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), device_index=local)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (gloo prints a connection banner on STDOUT: keep the contract's one JSON line clean by lending it stderr meanwhile)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend, rank=rank, world_size=world)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    red_dev = torch.device("cpu") if backend == "gloo" else torch.device("cuda", local)

    from pointreggpt_amd import geometry as G
    from pointreggpt_amd import synthetic
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import MaskUnet, Unet

    B, S = a.batch, a.size
    dev = torch.device("cuda", local)
    unet = Unet(a.dim, dtype=a.dtype).init_synthetic(seed=1, calibrated=not a.uncalibrated)
    mask = None if a.sampler_only else MaskUnet(a.dim, dtype=a.dtype).init_synthetic(seed=2, calibrated=True)
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

    # `--streams N` (default 2): N independent pipelines (own network handles, workspaces, sampler graphs), one host thread
    # and one HIP stream each, take the batches round-robin.  Every batch is still a B-scene launch of the same kernels; what
    # changes is that one pipeline's launch boundaries, ramps and drains (139 per evaluation) are filled by the other's
    # kernels instead of idling the chip.  N = 1 is the strictly serial loop of rounds 1-2.
    pipes = [dict(unet=unet, mask=mask, diff=diff, stream=None)]
    for k in range(1, max(1, a.streams)):
        u2 = Unet(a.dim, dtype=a.dtype).init_synthetic(seed=1, calibrated=not a.uncalibrated)
        m2 = None if a.sampler_only else MaskUnet(a.dim, dtype=a.dtype).init_synthetic(seed=2, calibrated=True)
        pipes.append(dict(unet=u2, mask=m2, stream=None,
                          diff=GaussianDiffusion(u2, image_size=S, timesteps=a.timesteps, sampling_timesteps=a.sampling_steps)))
    if len(pipes) > 1:
        for pp in pipes:
            pp["stream"] = torch.cuda.Stream(device=dev)

    def one_batch(bt, pp=None):
        pp = pp or pipes[0]
        mask_, diff_ = pp["mask"], pp["diff"]
        pc = G.param_vector(bt["K"])
        if a.sampler_only:
            rpj, hit = G.reproject_tensor(bt["depth"], bt["K"], bt["pose"], clip=(0, 10), depth_unit=10.0, out_scale=0.1)
            _, _, cond = G.apply_mask(torch.ones_like(rpj), rpj, hit, 0.5)
            return diff_.sample(param_cond=pc, img_cond=cond, seeds=bt["seeds"])
        rpj, hit = G.reproject_tensor(bt["depth"], bt["K"], bt["pose"], clip=(0, 10), depth_unit=10.0, out_scale=0.1)
        _, hit_c, cond = G.apply_mask(mask_(rpj), rpj, hit, 0.99)
        img = diff_.sample(param_cond=pc, img_cond=cond, seeds=bt["seeds"])
        out, _, _ = G.apply_mask(mask_(img), img, None, 0.99, want_cond=False)
        xyz, valid = G.unproject_f64(out, bt["K"], bt["pose"])
        last.update(img=img, known=hit_c, valid=valid)          # (references only: read after the clock stops)
        return xyz

    last = {}

    def run_batches(lo, hi, dynamic=False):
        if len(pipes) == 1:
            for i in range(lo, hi):
                one_batch(batches[i])
            return
        import threading
        errs = []

        # `dynamic`: the lanes pull the next batch from one shared counter instead of owning every n-th batch — with a batch count
        # that is not a multiple of the lane count (the driver's 20 steps on 3 lanes) no lane idles through a whole last round
        nxt, lock = [lo], threading.Lock()

        def take():
            with lock:
                i = nxt[0]
                nxt[0] += 1
            return i if i < hi else None

        def worker(k):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(pipes[k]["stream"]):
                    if dynamic:
                        i = take()
                        while i is not None:
                            one_batch(batches[i], pipes[k])
                            i = take()
                    else:
                        for i in range(lo + k, hi, len(pipes)):
                            one_batch(batches[i], pipes[k])
            except BaseException as e:      # noqa: BLE001 — re-raised on the main thread
                errs.append(e)

        th = [threading.Thread(target=worker, args=(k,)) for k in range(len(pipes))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if len(pipes) > a.warmup and total_batches:
        # setup, not a step: pipelines the W warm-up steps would not reach allocate their workspace and capture their graph
        for k in range(a.warmup, len(pipes)):
            with torch.cuda.stream(pipes[k]["stream"]):
                one_batch(batches[0], pipes[k])
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup0       # process group + weight synthesis / upload + input synthesis + lane graph capture
    t_w0 = time.perf_counter()
    run_batches(0, a.warmup)
    torch.cuda.synchronize()
    warmup_s = time.perf_counter() - t_w0
    barrier()
    t0 = time.perf_counter()
    run_batches(a.warmup, total_batches, dynamic=a.dynamic_lanes)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0              # this rank's own time (before the closing barrier)
    barrier()
    dt = time.perf_counter() - t0
    free_b, total_b = torch.cuda.mem_get_info()
    per_rank = None
    if dist is not None:
        tt = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # per-rank attribution of a slow barrier: every rank's setup / warm-up / own timed seconds and
        # the device memory in use on its GPU, gathered to rank 0
        mine = torch.tensor([setup_s, warmup_s, dt_own, float(total_b - free_b), float(affinity.get("cpus", 0)),
                             float(affinity.get("numa_node", -1))], device=red_dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "setup_s": float(v[0]), "warmup_s": float(v[1]), "timed_s": float(v[2]),
                     "device_bytes_in_use": int(v[3]), "pinned_cpus": int(v[4]), "numa_node": int(v[5])} for r, v in enumerate(allr)]

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
                   "noise": "on-device Philox4x32-10 keyed per scene",
                   "weights": "synthetic (deterministic initialiser" + (")" if a.uncalibrated else ", calibrated 1x1 head: x0 predictions inside (-1, 1))"),
                   "parallelism": f"scene-sharded x{world}, no collectives",
                   "hipgraph": "one captured transition (U-Net + update), replayed per step",
                   "streams": len(pipes),
                   "tflop_per_pair": tflop_pair},
        "end_to_end_mfma_frac": value / world * tflop_pair / MFMA_PEAK_TFLOPS[a.dtype],
        "setup": {"setup_s": setup_s, "warmup_s": warmup_s, "timed_s_this_rank": dt_own,
                  "what": "setup = process group + weight synthesis / upload + input synthesis + lane workspaces and graph capture (untimed)"},
        "device_memory": {"in_use_bytes": int(total_b - free_b), "total_bytes": int(total_b),
                          "what": "hipMemGetInfo on this rank's device after the timed loop (all ranks sharing the device included)"},
        "cpu_affinity": affinity,
    }
    if per_rank is not None:
        res["per_rank"] = per_rank
    if last:
        # the timed workload is not degenerate: how many in-painted pixels of the last timed batch ended on the [0,1] clamp,
        # and how many pixels survive the second depth correction into the clouds
        img, known = last["img"], last["known"].bool()
        free = ~known
        res["workload"] = {"saturated_fraction_inpainted": float(((img <= 0) | (img >= 1))[free].float().mean()),
                           "inpainted_fraction": float(free.float().mean()),
                           "inpainted_depth_m": {"mean": float(img[free].mean()) * 10, "std": float(img[free].std()) * 10},
                           "points_kept_fraction": float(last["valid"].float().mean())}

    if rank == 0 and not a.no_roofline:
        # dominant kernel class = the MFMA convolutions (w256 + c64 + ws + igemm).  Their launches are timed live with HIP
        # events on the sampler's own stream over `profile_transitions` transitions of the same workload (eager launches).
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
            "kernel": CONV_CLASS,
            "bound": "mfma", "achieved": ach,
            "peak": MFMA_PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS[a.dtype],
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (FETCH_SIZE*2 + WRITE_SIZE)", "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": pr.get("conv_bytes", 0.0) / max(1, pr["conv_launches"]),
            "peak_measured_random_operands": 1670.0 if a.dtype == "bf16" else None,   # tools/micro/mfma_power.hip, power-limited
            "launches": pr["conv_launches"], "avg_launch_us": pr["conv_ms"] * 1e3 / max(1, pr["conv_launches"]),
            "flop_per_launch": pr["conv_flops"] / max(1, pr["conv_launches"]),
            # `achieved` / `frac` are ALGORITHMIC (the reference operator's 2 * MAC count); the three Upsample convs
            # run as four 2 x 2-tap sub-pixel convolutions (4 / 9 of their MACs): what the matrix pipe actually executed
            "executed_achieved": pr.get("conv_flops_executed", pr["conv_flops"]) / (pr["conv_ms"] * 1e-3) / 1e12,
            "executed_frac": pr.get("conv_flops_executed", pr["conv_flops"]) / (pr["conv_ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[a.dtype],
            "executed_over_algorithmic_flops": pr.get("conv_flops_executed", pr["conv_flops"]) / max(1.0, pr["conv_flops"]),
            "share_of_step_time": pr["conv_ms"] / pr["total_ms"],
            "measured": f"HIP events around every conv launch, {nprof} transitions, batch {B}",
        }
        res["roofline"]["per_kernel"] = per_kernel_table(pr, MFMA_PEAK_TFLOPS[a.dtype], nprof)
        if not a.sampler_only:
            res["roofline_mem"] = mem_rooflines(G, bt, S, B, pr, stream_batch=1024 * (128 * 128) // (S * S) if B >= 16 else 4 * B)
        pdiff.close()
    if not a.no_e2e_files and not a.sampler_only:
        # every rank runs its own shard of the generate_dataset loop; aggregate like the headline metric
        e2e = e2e_files(a, unet, mask, diff, rank, world, B, S, lanes=[(pp["diff"], pp["mask"]) for pp in pipes[1:]])
        dt_e = e2e["seconds"]
        if dist is not None:
            tt = torch.tensor([dt_e], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_e = float(tt.item())
        e2e.update(value=world * e2e["pairs"] / dt_e, unit="pairs/s on disk (whole job)", seconds_max_over_ranks=dt_e,
                   vs_device_only=world * e2e["pairs"] / dt_e / value,
                   what="Generator.generate --synthetic: memory-cloud z-buffer + MaskUnet + sampler + MaskUnet + unprojection + "
                        "crop / 0.025 voxel grid / PLY + PNG + text files through the C++ writer pool, overlapped with the next batch")
        res["e2e_files"] = e2e
    # the single-GPU diagnostics (drift against the fixtures, the configs[4] leg, the CPU baseline) belong to the N = 1 line
    # only: at N > 1 the other ranks would sit in the closing barrier while rank 0 runs them
    if rank == 0 and world == 1 and not a.no_drift and a.dim == 64:
        res["drift_vs_reference"] = drift_vs_reference(sorted({a.dtype, "bf16", "mxfp8", "f16x3", "fp32"}), a.dim)
    if rank == 0 and world == 1 and not a.no_parity_mode and not a.sampler_only and a.dtype == "bf16":
        # the modes that hold the north-star tolerance (1e-4 m point-XYZ against the reference: tests/test_gpu_f16x3.py,
        # test_long_chain_fp32_north_star), at the headline shape: what "correct" costs next to the bf16 headline
        # fp32 on one lane (as in rounds 3-4); f16x3 on the headline leg's `--streams` lanes, its one-lane figure kept beside it
        pm = {"fp32": parity_mode_leg(a, G, synthetic, rank, "fp32"),
              "f16x3": parity_mode_leg(a, G, synthetic, rank, "f16x3", lanes=max(1, a.streams))}
        if a.streams > 1:
            one_lane = parity_mode_leg(a, G, synthetic, rank, "f16x3")
            pm["f16x3"]["one_lane"] = {k: one_lane[k] for k in ("pairs_per_s", "ms_per_transition", "streams")}
            if "roofline" in one_lane and "roofline" not in pm["f16x3"]:
                pm["f16x3"]["roofline"] = one_lane["roofline"]
        # equal lane counts: one-lane f16x3 over one-lane fp32; the mixed figure is labelled as such
        f1 = pm["f16x3"].get("one_lane", pm["f16x3"])["pairs_per_s"]
        pm["f16x3_vs_fp32_one_lane"] = f1 / pm["fp32"]["pairs_per_s"]
        pm["f16x3_lanes_vs_fp32_one_lane"] = pm["f16x3"]["pairs_per_s"] / pm["fp32"]["pairs_per_s"]
        # ... and at the shipped setting (256x256, 250-step DDIM: configs[4]'s shape, generate_dataset.py:34-49) in the f16x3 mode
        pm["f16x3_256_ddim250"] = parity_mode_leg(a, G, synthetic, rank, "f16x3", shape=(a.c4_batch, 256, 250), lanes=max(1, a.streams))
        pm["headline_vs_fp32"] = value / pm["fp32"]["pairs_per_s"]
        # measured in THIS run (drift_vs_reference above), not quoted: point-XYZ L-infinity of each chain against the reference
        dv = res.get("drift_vs_reference", {})
        per = {dt: {k: v[dt]["xyz_linf_m"] for k, v in dv.items() if isinstance(v, dict) and dt in v} for dt in ("fp32", "f16x3", "bf16")}
        cal128 = ("G22_chain1000_ancestral_128", "G20_ddim250_128", "G19_chain1000_ancestral_64")
        worst = lambda dt, names: max([per[dt][k] for k in names if k in per[dt]], default=None)
        pm["tolerance"] = {"north_star_m": 1e-4, "metric": "point-XYZ L-infinity vs the reference (m), measured in this run",
                           "per_chain": per,
                           # the calibrated chains of each leg's image size (G21 = the saturating 256x256 stress chain: bounded by
                           # the reference's own distance from its float64 twin, tests/test_gpu_parity.py)
                           "worst_xyz_m": {"fp32": worst("fp32", cal128), "f16x3": worst("f16x3", cal128),
                                           "f16x3_256": worst("f16x3", ("G21b_ddim250_256",)), "bf16": worst("bf16", cal128)},
                           "stress_chain_G21_xyz_m": {dt: per[dt].get("G21_ddim250_256") for dt in per},
                           "labelled_parity": "fp32 (tests/test_gpu_parity.py::test_long_chain_fp32_north_star) and f16x3 "
                                              "(tests/test_gpu_f16x3.py) on every calibrated chain"}
        res["parity_mode"] = pm
    if rank == 0 and world == 1 and not a.no_configs4 and not a.sampler_only:
        res["configs4"] = configs4_leg(a, G, synthetic, rank)
        # the same shape, same call, same box in bf16: whether fp8 operands buy anything on their own config
        cmp_ = configs4_leg(a, G, synthetic, rank, dt="bf16")
        res["configs4"]["bf16_same_shape"] = {k: cmp_[k] for k in ("value", "unit", "ms_per_step", "dtype", "end_to_end_mfma_frac") if k in cmp_}
        if "roofline" in cmp_:
            res["configs4"]["bf16_same_shape"]["roofline"] = {k: cmp_["roofline"][k] for k in ("achieved", "peak", "frac", "avg_launch_us", "launches")}
        res["configs4"]["mxfp8_over_bf16_same_shape"] = res["configs4"]["value"] / cmp_["value"]
        # the measured ceiling of MX operands on this network (profiles/r05_mxcap_activations_in_memory_cap.txt)
        res["configs4"]["mx_cap_measured"] = "<=1.15x bf16 (profiles/r05_mxcap_activations_in_memory_cap.txt)"
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(S, a.dim)

    if rank == 0:
        res["evidence"] = EVIDENCE
        # the driver parses the LAST stdout line: a short contract line (< 4 KB, asserted here and in the tests); every table,
        # per-rank row, drift chain and prose string lives in the sidecar
        side = write_sidecar(res)
        line = contract_line(res, side)
        print(json.dumps(res), file=sys.stderr, flush=True)
        print(line, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
