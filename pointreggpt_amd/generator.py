"""`Generator.generate` (sd:2250-2694) on the HIP hot path: same sequence, same on-disk layout, batched geometry.

Per batch of scenes: source depth frame -> scene "memory" cloud (crop box) -> sample-000000.cloud.ply; for each sample:
random pose -> z-buffer reprojection of the memory clouds (ONE launch for the ragged batch) -> depth correction ->
DDNM condition -> diffusion sampler -> depth correction -> float64 unprojection into the common frame -> fragment
-> crop / voxel / sample-000001.cloud.ply, plus the pose / intrinsic / png side files the reference writes.

Input side: `--synthetic` scenes (pointreggpt_amd.synthetic; what tests and benchmarks use — neither box has 3DMatch)
or the reference's real-data layout (train_info.pkl + `<cloud>.info.txt` + 3DMatch RGB-D frames, sd:2352-2459).
The real-data decode path mirrors torchvision Resize(NEAREST)/CenterCrop/ToTensor with PIL and is parity-unpinned.

Sharding: scenes are independent; under torchrun every rank takes a contiguous block of the requested range
(pointreggpt_amd.sharding) — the reference leaves this to manual -start/-stop.
sd = /root/reference/denoising_diffusion_pytorch/successive_ddnm_diffusion.py
"""
from __future__ import annotations

import os
import pickle
import shutil
import sys
import threading
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch

from . import geometry as G
from . import postprocess as PP
from . import synthetic
from .sharding import num_to_groups

PAIRS_PER_LAP = 20642   # real pairs per lap of the scene index; src/tgt swap on odd laps (sd:2397-2410)


class Generator:
    def __init__(self, diffusion_model, folder: Optional[str], *, batch_size=16, samples_folder="./samples",
                 results_folder="./results", synthetic_seed: Optional[int] = None, device="cuda", **_ignored):
        self.model = diffusion_model                  # pointreggpt_amd.diffusion.GaussianDiffusion
        self.folder = folder
        self.batch_size = int(batch_size)
        self.image_size = diffusion_model.image_size
        self.samples_folder = Path(samples_folder)
        self.samples_folder.mkdir(parents=True, exist_ok=True)
        self.results_folder = Path(results_folder)
        self.synthetic_seed = synthetic_seed
        self.device = torch.device(device)
        # device ops: the HIP library (default; raises without a GPU) or, ONLY when asked for with device="cpu", the
        # prg_cpu_* twins (BASELINE configs[0]: the plumbing run that needs no GPU)
        if self.device.type == "cpu":
            from . import cpu as _cpu
            self.G = _cpu.ops
        else:
            self.G = G

    # -- weights (sd:2307-2324): the generator samples from the EMA copy ------------------------------------------
    def load(self, milestone, unet=None):
        from .weights import unet_state_from_checkpoint
        data = torch.load(str(self.results_folder / f"model-{milestone}.pt"), map_location="cpu")
        net = unet if unet is not None else self.model.model
        net.load_state_dict(unet_state_from_checkpoint(data, net.cfg))

    # -- input side ------------------------------------------------------------------------------------------------
    def _real_scene(self, abs_idx: int, info_train, scene_dir: Path):
        """Source frame of scene `abs_idx` from the 3DMatch layout (sd:2397-2459).  Returns depth (S,S) f32 in
        10 m units and K (3,3) f32."""
        from PIL import Image
        S = self.image_size
        lap_odd = (abs_idx // PAIRS_PER_LAP) % 2 == 1
        rel = info_train["tgt" if lap_odd else "src"][abs_idx % PAIRS_PER_LAP]
        info_path = os.path.join("./dataset/indoor/data", rel).replace(".pth", ".info.txt")
        with open(info_path) as f:
            scene_name, seq_name, f0, _f1 = f.readline().rstrip("\n").split()
        root = os.path.join(self.folder, scene_name)
        K = G.intrinsic_transform(np.loadtxt(os.path.join(root, "camera-intrinsics.txt")), resize=S,
                                  centercrop=S).astype(np.float32)
        img = Image.open(os.path.join(root, seq_name, "frame-{:0>6d}.depth.png".format(int(f0))))
        w, h = img.size                                   # Resize(S): short side -> S (sd:2356-2361)
        if w <= h:
            nw, nh = S, int(S * h / w)
        else:
            nw, nh = int(S * w / h), S
        img = img.resize((nw, nh), Image.NEAREST)
        left, top = int(round((nw - S) / 2.0)), int(round((nh - S) / 2.0))
        img = img.crop((left, top, left + S, top + S))
        depth = np.asarray(img).astype(np.float32) * np.float32(1e-4)      # uint16 mm -> 10 m units
        depth[depth > 1] = 0
        return depth, K

    def _scene_inputs(self, abs_idx: int, info_train, scene_dir: Path):
        if self.synthetic_seed is not None:
            depth, K, _pose = synthetic.synth_scene(self.synthetic_seed, abs_idx, self.image_size)
            return depth, K
        return self._real_scene(abs_idx, info_train, scene_dir)

    def _poses(self, idxs: List[int], sample_idx: int, pose_seed: Optional[int] = None) -> np.ndarray:
        if self.synthetic_seed is None:
            if pose_seed is not None:          # reproducible and shard-invariant: a LOCAL legacy stream per (job, batch, sample)
                from .sharding import batch_pose_seed                      # (the process-wide numpy state is left alone)
                rs = np.random.RandomState(batch_pose_seed(pose_seed, idxs[0], sample_idx))
                return G.random_sample_pose(len(idxs), rng=rs).astype(np.float32)
            with Generator._pose_lock:         # numpy's legacy GLOBAL stream, as the reference (single lane only)
                return G.random_sample_pose(len(idxs)).astype(np.float32)
        out = []
        for i in idxs:                                                      # shard-invariant per-scene stream
            rng = synthetic.scene_rng(self.synthetic_seed, i)
            rng.random(4096 + 64 * sample_idx)                              # decorrelate from the depth draw
            out.append(synthetic.synth_pose(rng))
        return np.stack(out)

    # -- the pipeline ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, start_scene_index, stop_scene_index, num_samples, memory_voxel_size=0.002,
                 save_voxel_size=0.025, has_refine_step=False, depth_correction=None, mask_threshold=0.99,
                 noise_seed: Optional[int] = None, progress: bool = False, writer_threads: int = 0, stats: Optional[dict] = None,
                 seed_poses: Optional[bool] = None, lanes: Optional[list] = None):
        """Same sequence as sd:2363-2694.  File output is asynchronous: every batch's clouds / images / text files are
        handed to the library's C++ writer pool (crop, voxel grid, PLY / PNG encoding on worker threads) and are
        produced while the GPU samples the next batch.  A batch's resume marker — the generated cloud of its LAST scene
        (sd:2371-2381) — is submitted only after everything else of that batch is on disk, so an interrupted run never
        skips an incomplete batch.

        ``lanes``: extra (GaussianDiffusion, MaskUnet) pairs holding THE SAME weights on their own library handles.  With
        n lanes (this object's model + ``depth_correction`` is lane 0) the batches are dealt round-robin to n host threads,
        each on its own HIP stream: a batch is still one B-scene launch sequence, but the launch boundaries, ramps and
        drains of one lane (139 kernels per U-Net evaluation) are filled by the other lane's kernels (+7 % pairs/s measured
        at B = 64, 128x128).  A scene's files do not depend on the lane that produced it.
        ``seed_poses`` (real-data input only): draw the poses from a local legacy stream per (job seed, batch, sample) so that
        a run can be re-sharded / resumed reproducibly; False = numpy's global stream like the reference (single lane only).
        Default: True when the caller passes a `noise_seed` (ANY value, 0 included: `None` is the "no seed" sentinel) or lanes,
        False (the reference's unseeded behaviour) otherwise."""
        if seed_poses is None:
            seed_poses = noise_seed is not None or (lanes is not None and len(lanes) > 0)
        noise_seed = 0 if noise_seed is None else int(noise_seed)
        info_train = None
        if self.synthetic_seed is None:
            with open("./dataset/indoor/metadata/train_info.pkl", "rb") as f:
                info_train = pickle.load(f)
        batches = []
        first = start_scene_index
        for batch in num_to_groups(stop_scene_index - start_scene_index, self.batch_size):
            idxs = list(range(first, first + batch))
            first += batch
            # resume: skip a batch whose last scene already has its generated cloud (sd:2371-2381)
            done = self.samples_folder / "scene-{:0>6d}/sample-{:0>6d}.cloud.ply".format(idxs[-1], num_samples // 2)
            if done.is_file():
                print("Skip completed scene {:0>6d} - {:0>6d}.".format(idxs[0], idxs[-1]))
                continue
            batches.append(idxs)
        pairs = [(self.model, depth_correction)] + list(lanes or [])
        if self.device.type == "cpu":
            pairs = pairs[:1]                 # lanes are HIP streams
        if self.synthetic_seed is None and not seed_poses:
            pairs = pairs[:1]                 # one global pose stream cannot be shared by concurrent lanes
        pairs = pairs[:max(1, len(batches))]
        kw = dict(num_samples=num_samples, memory_voxel_size=memory_voxel_size, save_voxel_size=save_voxel_size,
                  has_refine_step=has_refine_step, mask_threshold=mask_threshold, noise_seed=noise_seed, progress=progress,
                  seed_poses=seed_poses, info_train=info_train)
        n = len(pairs)
        wt = writer_threads if writer_threads <= 0 or n == 1 else max(1, writer_threads // n)
        results = [None] * n
        if n == 1:
            results[0] = self._lane(batches, pairs[0][0], pairs[0][1], wt, 1, **kw)
        else:
            import threading
            errs = []
            stop = threading.Event()          # set by the first failing lane: the others stop at their next batch
            dev = self.device if self.device.index is not None else torch.device("cuda", torch.cuda.current_device())

            def work(k):
                try:
                    torch.cuda.set_device(dev)
                    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                        results[k] = self._lane(batches[k::n], pairs[k][0], pairs[k][1], wt, n, stop=stop, **kw)
                        torch.cuda.current_stream().synchronize()
                except BaseException as e:      # noqa: BLE001 — re-raised on the calling thread
                    errs.append((k, e))
                    stop.set()

            threads = [threading.Thread(target=work, args=(k,)) for k in range(n)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errs:
                for k, e in errs[1:]:         # nothing is dropped: later lanes' errors are reported, the first is raised
                    print("lane {} also failed: {!r}".format(k, e), file=sys.stderr)
                raise errs[0][1]
            if stop.is_set():
                raise RuntimeError("a lane stopped early without an error")
        if stats is not None:
            stats.update(pairs=sum(r[0] for r in results), writer_jobs=sum(r[1] for r in results),
                         writer_threads=sum(r[2] for r in results), lanes=n)

    _pose_lock = threading.Lock()      # guards numpy's process-wide legacy stream (seed_poses=False)

    def _lane(self, batches, model, depth_correction, writer_threads, n_lanes, *, num_samples, memory_voxel_size,
              save_voxel_size, has_refine_step, mask_threshold, noise_seed, progress, seed_poses, info_train, stop=None):
        """One lane's share of the batches (all of them with a single lane), on the calling thread's current stream."""
        S, dev = self.image_size, self.device
        if writer_threads <= 0 and n_lanes > 1:
            writer_threads = max(2, min(16, PP.rank_cpu_budget() // 2) // n_lanes)
        with PP.WriterPool(writer_threads) as pool:
            marker_prev = None                # deferred submission of the previous batch's resume marker
            n_pairs = 0
            marker_index = num_samples // 2   # the cloud file the resume check looks for (0 or 1; never written beyond that)

            def flush_marker():
                nonlocal marker_prev
                if marker_prev is not None:
                    pool.wait()               # the previous batch had a whole GPU batch time to finish: no stall in practice
                    marker_prev()
                    marker_prev = None

            for idxs in batches:
                if stop is not None and stop.is_set():
                    break                      # another lane failed: do not sample this lane's whole share first
                batch = len(idxs)
                K = np.zeros((batch, 3, 3), dtype=np.float32)
                depth0 = np.zeros((batch, 1, S, S), dtype=np.float32)
                sdirs = []
                for j, idx in enumerate(idxs):
                    sdir = self.samples_folder / "scene-{:0>6d}".format(idx)
                    if sdir.exists():
                        shutil.rmtree(str(sdir), ignore_errors=True)
                    sdir.mkdir(parents=True, exist_ok=True)
                    sdirs.append(sdir)
                    depth0[j, 0], K[j] = self._scene_inputs(idx, info_train, sdir)
                K_dev = torch.from_numpy(K).to(dev)
                # the source frames of the whole batch in ONE unprojection (sd:2479-2483 does it per scene)
                frames = self.G.point_clouds(torch.from_numpy(depth0).to(dev), K_dev, None)
                memory: List[np.ndarray] = []
                marker_cur = None
                for j in range(batch):
                    scene_pc = PP.crop_aabb(frames[j].astype(np.float32)).astype(np.float32)   # the scene "memory" (sd:2484-2490)
                    memory.append(scene_pc)
                    pool.text(str(sdirs[j] / "camera-intrinsics.txt"), K[j])
                    pool.image01(str(sdirs[j] / "sample-{:0>6d}.image.png".format(0)), depth0[j, 0])
                    job = (lambda path=str(sdirs[j] / "sample-{:0>6d}.cloud.ply".format(0)), pc=scene_pc:
                           pool.cloud(path, pc, None, crop=False, voxel=save_voxel_size))
                    if j == batch - 1 and marker_index == 0:
                        marker_cur = job              # this batch's resume marker: written after everything else
                    else:
                        job()
                param_cond = self.G.param_vector(K_dev)
                fragments: List[Optional[np.ndarray]] = [None] * batch
                poses0 = None
                for sample_idx in range(num_samples):
                    pose = self._poses(idxs, sample_idx, noise_seed if seed_poses else None)
                    if sample_idx == 0:
                        poses0 = pose
                    pose_dev = torch.from_numpy(pose).to(dev)
                    rpj, hit = self.G.project_clouds(memory, pose, K, S, dev, depth_scale=0.1)
                    prob = depth_correction(rpj)
                    rpj_c, hit_c, cond = self.G.apply_mask(prob, rpj, hit, mask_threshold)
                    seeds = [synthetic.noise_seed(noise_seed, i, sample_idx) for i in idxs]
                    images = model.sample(param_cond=param_cond, img_cond=cond, seeds=seeds,
                                          has_refine_step=has_refine_step)
                    prob2 = depth_correction(images)
                    images, _, _ = self.G.apply_mask(prob2, images, None, mask_threshold, want_cond=False)
                    xyz, valid = self.G.unproject_f64(images, K_dev, pose_dev)      # common frame, float64 (sd:2623-2628)
                    # one blocking copy per tensor: the host waits here for the GPU while the pool writes the previous batch
                    rpj_host, crt_host, img_host = rpj.cpu().numpy(), rpj_c.cpu().numpy(), images.cpu().numpy()
                    xyz_host, valid_host = xyz.cpu().numpy(), valid.cpu().numpy()
                    flush_marker()
                    last_sample = sample_idx == num_samples - 1
                    for j, idx in enumerate(idxs):
                        sdir = sdirs[j]
                        pool.image01(str(sdir / "reprojected.image.png"), rpj_host[j])
                        pool.text(str(sdir / "sample-{:0>6d}.pose.txt".format(sample_idx + 1)), np.linalg.inv(pose[j]))
                        pool.image01(str(sdir / "corrected.image.png"), crt_host[j])
                        pool.image01(str(sdir / "sample-{:0>6d}.image.png".format(sample_idx + 1)), img_host[j])
                        pool.depth16(str(sdir / "sample-{:0>6d}.depth.png".format(sample_idx + 1)), img_host[j])
                        if num_samples == 1:
                            frag, fvalid = xyz_host[j], valid_host[j]                  # compaction happens in the worker
                        else:
                            pc = xyz_host[j][valid_host[j]]
                            fragments[j] = pc if sample_idx == 0 else np.concatenate([fragments[j], pc], axis=0)
                            frag, fvalid = fragments[j], None
                            if not last_sample:                                         # memory update (sd:2661-2680)
                                merged = np.concatenate([memory[j], pc], axis=0)
                                memory[j] = PP.native_voxel_down_sample(merged, memory_voxel_size).astype(np.float32)
                        if last_sample:                                                 # sd:2640-2658
                            path = str(sdir / "sample-{:0>6d}.cloud.ply".format(1))
                            job = (lambda path=path, frag=frag, fvalid=fvalid, T=poses0[j].astype(np.float64):
                                   pool.cloud(path, frag, fvalid, pre=T, crop=True, voxel=save_voxel_size, post=np.linalg.inv(T)))
                            if j == batch - 1 and marker_index == 1:
                                marker_cur = job
                            else:
                                job()
                    if progress:
                        print("batch {:0>6d}-{:0>6d}: sample {}/{}".format(idxs[0], idxs[-1], sample_idx + 1, num_samples))
                n_pairs += batch
                flush_marker()                # (only reached with a pending marker when the sample loop did not run)
                marker_prev = marker_cur
            flush_marker()
            jobs = pool.wait()
            return n_pairs, jobs, pool.threads


def generate_gt(dataset_name: str, start_scene_index: int, stop_scene_index: int, num_samples: int,
                root: str = ".", overlap: str = "hip", scenes_per_launch: int = 512) -> None:
    """generate_gt.py:105-175 — per scene, every pair of .cloud.ply -> overlap ratios -> scene gt.log.

    The reference queries a KD-tree once per point in a Python loop (the wall-clock tail of a 10k-scene dataset); here the
    clouds of up to `scenes_per_launch` scenes are voxel-down-sampled in C++ and all their pairs go through ONE
    prg_overlap_counts launch.  `overlap='numpy-spec'` selects the numpy specification in postprocess.py instead — a test
    hook for boxes without a GPU, never chosen implicitly."""
    from itertools import combinations
    if overlap not in ("hip", "numpy-spec"):
        raise ValueError("overlap must be 'hip' or 'numpy-spec'")
    data = Path(root) / dataset_name / "data"
    todo = []                                   # (scene_name, gt_path, [(s, t, src, tgt), ...])
    for scene_idx in range(start_scene_index, stop_scene_index):
        scene_name = "scene-{:0>6d}".format(scene_idx)
        sdir = data / scene_name
        gt_path = sdir / "gt.log"
        if gt_path.exists():
            print("scene gt log has existed, skip over it")
            continue
        cand = []
        for s, t in combinations(range(num_samples), 2):
            ps, pt = sdir / "sample-{:0>6d}.cloud.ply".format(s), sdir / "sample-{:0>6d}.cloud.ply".format(t)
            if not ps.exists() or not pt.exists():
                continue
            src, tgt = PP.read_ply(str(ps)), PP.read_ply(str(pt))
            if len(src) < 1000 or len(tgt) < 1000:
                continue
            cand.append((s, t, src, tgt))
        todo.append((scene_name, gt_path, cand))
        if len(todo) >= scenes_per_launch:
            _finish_gt(todo, overlap)
            todo = []
    if todo:
        _finish_gt(todo, overlap)


def _finish_gt(todo, overlap: str) -> None:
    flat = [(src, tgt) for _n, _p, cand in todo for (_s, _t, src, tgt) in cand]
    if overlap == "hip":
        ratios = PP.overlap_ratios_hip(flat)
    else:
        ratios = [PP.compute_overlap_ratio(a, b) for a, b in flat]
    k = 0
    for scene_name, gt_path, cand in todo:
        lines = []
        for (s, t, _src, _tgt) in cand:
            o_s, o_t = ratios[k]
            k += 1
            if np.isnan(o_s) or np.isnan(o_t) or (o_s < 0.1 and o_t < 0.1):
                continue
            lines.append("{}\t{}\t{}\t{:.4f}\t{:.4f}\n".format(scene_name, s, t, o_s, o_t))
        gt_path.parent.mkdir(exist_ok=True)
        with open(gt_path, "w") as f:
            f.writelines(lines)


def gather_gt(dataset_name: str, start_index: int, stop_index: int, root: str = ".") -> None:
    """generate_gt.py:177-188 — concatenate the scene logs in index order into metadata/gt.log (replacing it)."""
    final = Path(root) / dataset_name / "metadata" / "gt.log"
    final.parent.mkdir(parents=True, exist_ok=True)
    if final.exists():
        os.remove(str(final))
    with open(final, "ab") as out:
        for scene_idx in range(start_index, stop_index):
            p = Path(root) / dataset_name / "data" / "scene-{:0>6d}".format(scene_idx) / "gt.log"
            if p.exists():
                out.write(p.read_bytes())
