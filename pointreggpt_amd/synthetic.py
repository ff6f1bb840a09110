"""Synthetic scenes for benchmarks and parity tests (neither box has 3DMatch data or released weights).

Per scene index ``i`` everything is drawn from ``numpy.random.Generator(PCG64(SeedSequence([base_seed, i])))``
so a scene is the same whichever rank, shard or batch position produces it.

  * source depth (S,S) float32, unit 1.0 == 10 m: a tilted plane with 2–4 nearer boxes, values within
    [0.08, 0.35] (0.8–3.5 m: inside the generator's crop box, reference sd:2348), 5 % random holes (0),
    anything > 1 zeroed as the reference does to real frames (sd:2459);
  * intrinsics: one of the six 3DMatch K's with the reference's probabilities (sd:358-368) pushed through
    the Resize(S)+CenterCrop(S) transform (sd:47-119, restated in geometry.intrinsic_transform);
  * pose: the distribution of ``random_sample_pose`` (sd:417-443) but from the per-scene generator.

sd = /root/reference/denoising_diffusion_pytorch/successive_ddnm_diffusion.py
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

from . import geometry as G


def scene_rng(base_seed: int, scene_index: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(base_seed), int(scene_index)])))


def synth_depth(rng: np.random.Generator, S: int) -> np.ndarray:
    yy, xx = np.meshgrid(np.linspace(-1, 1, S), np.linspace(-1, 1, S), indexing="ij")
    base = rng.uniform(0.22, 0.30)
    d = base + rng.uniform(-0.04, 0.04) * xx + rng.uniform(-0.04, 0.04) * yy
    for _ in range(int(rng.integers(2, 5))):
        h, w = rng.integers(S // 8, S // 2, size=2)
        y0, x0 = rng.integers(0, S - h), rng.integers(0, S - w)
        d[y0:y0 + h, x0:x0 + w] = rng.uniform(0.10, base - 0.02) + 0.01 * xx[y0:y0 + h, x0:x0 + w]
    d = np.clip(d, 0.08, 0.35).astype(np.float32)
    d[rng.random((S, S)) < 0.05] = 0.0
    d[d > 1] = 0.0
    return d


def synth_pose(rng: np.random.Generator, center=(0.0, 0.0, 3.0)) -> np.ndarray:
    th = rng.uniform(-np.pi / 24, np.pi / 24)
    ph = rng.uniform(-np.pi / 12, np.pi / 12)
    R = Rotation.from_euler("XYZ", [th, ph, 0.0]).as_matrix()
    c = np.asarray(center, dtype=np.float64)
    jitter = rng.standard_normal(3) / 3
    jitter[2] = 0
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = c - R @ c + jitter
    return T.astype(np.float32)


def synth_intrinsic(rng: np.random.Generator, S: int) -> np.ndarray:
    cands = G.candidate_intrinsics()
    p = np.asarray(G.K_WEIGHTS, dtype=np.float64)
    k = cands[rng.choice(len(cands), p=p / p.sum())]
    return G.intrinsic_transform(k, resize=S, centercrop=S).astype(np.float32)


def synth_scene(base_seed: int, scene_index: int, S: int):
    """-> (depth (S,S) f32, K (3,3) f32, pose (4,4) f32) for one scene."""
    rng = scene_rng(base_seed, scene_index)
    depth = synth_depth(rng, S)
    K = synth_intrinsic(rng, S)
    pose = synth_pose(rng)
    return depth, K, pose


def synth_batch(base_seed: int, scene_indices, S: int):
    """-> depth (B,1,S,S), K (B,3,3), pose (B,4,4), all float32 numpy."""
    ds, ks, ps = zip(*(synth_scene(base_seed, i, S) for i in scene_indices))
    return np.stack(ds)[:, None], np.stack(ks), np.stack(ps)


def noise_seed(base_seed: int, scene_index: int, sample_index: int = 0) -> int:
    """64-bit Philox key for the diffusion noise of (scene, sample): shard-invariant (depends only on the run's seed and
    the two indices; no aliasing between scenes and samples)."""
    words = [int(base_seed) & 0xFFFFFFFFFFFFFFFF, int(scene_index), 0x6E6F6973]
    if sample_index:
        words.append(int(sample_index))
    return int(np.random.SeedSequence(words).generate_state(1, np.uint64)[0])
