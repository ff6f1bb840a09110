"""`Tester.sample` / `Tester.generate` (sd:1960-2247) on the HIP hot path: successive multi-view generation.

Both start from an UNCONDITIONAL sample per scene and then generate view after view, each conditioned (DDNM) on what the
previous views already show:

  sample()    the camera moves 0.5 m forward per view; the previous view's depth is reprojected into the new camera
              (fused unproject -> SE(3) -> z-buffer kernel), passed through `occlusion_filter` (sd:446-463) and becomes
              the condition (sd:2026-2046);
  generate()  the camera turns by `random_sample_transform` (sd:377-415); the scene's accumulated cloud (voxel grid
              0.005) is z-buffered into the new view (one ragged launch for the batch), the new view is unprojected,
              merged into the cloud and the grid is refreshed (sd:2150-2228); at the end a 0.025 grid is written per scene.

Every kernel is the generator's; only the orchestration differs.  Files: `scene-<i>-sample-<k>.ply|png`,
`scene-<i>-camera-intrinsics.txt` (sample), `scene-<i>.ply` (generate), written by the C++ writer pool.  The reference's
matplotlib triptychs (`plt.imsave`, cmap gray / plasma) are written as 8-bit grey [last | reprojected | new] strips; its
`overview.png` grid is omitted (visual aid only).
sd = /root/reference/denoising_diffusion_pytorch/successive_ddnm_diffusion.py
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import geometry as G
from . import postprocess as PP
from . import synthetic
from .sharding import num_to_groups


class Tester:
    def __init__(self, diffusion_model, *, batch_size=16, samples_folder="./samples", device="cuda", seed: int = 0,
                 **_ignored):
        self.model = diffusion_model                      # pointreggpt_amd.diffusion.GaussianDiffusion
        self.batch_size = int(batch_size)
        self.image_size = diffusion_model.image_size
        self.samples_folder = Path(samples_folder)
        self.samples_folder.mkdir(parents=True, exist_ok=True)
        self.device = torch.device(device)
        self.seed = int(seed)

    # -- helpers ---------------------------------------------------------------------------------------------------
    def _intrinsics(self, batch: int) -> np.ndarray:
        """sd:1967-1971: the six 3DMatch intrinsics drawn from numpy's legacy stream, through Resize + CenterCrop."""
        return G.intrinsic_transform(G.random_sample_intrinsic(batch), resize=self.image_size,
                                     centercrop=self.image_size).astype(np.float32)

    def _sample(self, param_cond, cond, scene_ids: Sequence[int], stage: int, noise, has_refine_step=False):
        if noise is not None:
            return self.model.sample(param_cond=param_cond, img_cond=cond, noise=noise[stage].to(self.device),
                                     has_refine_step=has_refine_step)
        seeds = [synthetic.noise_seed(self.seed, i, stage) for i in scene_ids]
        return self.model.sample(param_cond=param_cond, img_cond=cond, seeds=seeds, has_refine_step=has_refine_step)

    @staticmethod
    def _strip(last, rpj, new) -> np.ndarray:
        return np.concatenate([last, rpj, new], axis=-1)

    # -- Tester.sample (sd:1960-2093) --------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, num_scenes: int, num_samples: int, noise: Optional[List[torch.Tensor]] = None,
               has_refine_step: bool = False, step=(0.0, 0.0, 0.5)) -> List[torch.Tensor]:
        """Returns, per batch, the (B, 1, S, S*num_samples) strip of all views (what the reference concatenates for its
        overview).  `noise[k]` = stored draws for view k (parity tests); default on-device Philox keyed per scene/view."""
        dev, S = self.device, self.image_size
        with PP.WriterPool() as pool:
            strips = []
            first = 0
            for batch in num_to_groups(num_scenes, self.batch_size):
                ids = list(range(first, first + batch))
                first += batch
                K = self._intrinsics(batch)
                K_dev = torch.from_numpy(K).to(dev)
                absolute = np.stack([np.eye(4) for _ in range(batch)]).astype(np.float32)
                param_cond = G.param_vector(K_dev)
                images = self._sample(param_cond, None, ids, 0, noise)                       # unconditional (sd:1978)
                views = [images]
                clouds = G.point_clouds(images, K_dev, None, clip=(0.5, 3.5))
                img_host = images.cpu().numpy()
                zero = np.zeros((S, S), dtype=np.float32)
                for j, i in enumerate(ids):
                    pool.image01(str(self.samples_folder / f"scene-{i}-sample-0.png"), self._strip(zero, zero, img_host[j, 0]))
                    pool.cloud(str(self.samples_folder / f"scene-{i}-sample-0.ply"), clouds[j], None, crop=False, voxel=0.0)
                    pool.text(str(self.samples_folder / f"scene-{i}-camera-intrinsics.txt"), K[j])
                for k in range(1, num_samples):
                    relative = np.stack([np.eye(4) for _ in range(batch)])
                    relative[..., :3, 3] = np.asarray(step, dtype=np.float64)
                    relative = relative.astype(np.float32)
                    absolute = relative @ absolute
                    rel_dev = torch.from_numpy(relative).to(dev)
                    rpj, hit = G.reproject_tensor(images, K_dev, rel_dev, clip=(0, 10), depth_unit=10.0, out_scale=1.0)
                    if np.sum(absolute[..., :3, 3] ** 2) != 0:
                        rpj, hit = G.occlusion_filter(rpj, hit)
                    rpj = rpj * 0.1
                    cond = torch.cat([rpj, hit.to(rpj.dtype)], dim=1) * 2 - 1
                    last = images
                    images = self._sample(param_cond, cond, ids, k, noise, has_refine_step)
                    views.append(images)
                    clouds = G.point_clouds(images, K_dev, torch.from_numpy(absolute).to(dev), clip=(0.5, 3.5))
                    l_h, r_h, n_h = last.cpu().numpy(), rpj.cpu().numpy(), images.cpu().numpy()
                    for j, i in enumerate(ids):
                        pool.image01(str(self.samples_folder / f"scene-{i}-sample-{k}.png"),
                                     self._strip(l_h[j, 0], r_h[j, 0], n_h[j, 0]))
                        pool.cloud(str(self.samples_folder / f"scene-{i}-sample-{k}.ply"), clouds[j], None, crop=False, voxel=0.0)
                strips.append(torch.cat(views, dim=-1))
            pool.wait()
        return strips

    # -- Tester.generate (sd:2095-2247) ------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, num_scenes: int, num_samples: int, voxel_size: float = 0.005,
                 noise: Optional[List[torch.Tensor]] = None, has_refine_step: bool = False) -> List[List[np.ndarray]]:
        """Returns, per batch, the scenes' accumulated float32 clouds (before the final 0.025 grid written to disk).
        (The reference hands pc2depth_tensor the whole batch's intrinsics next to ONE scene's cloud, sd:2177-2184; the
        evident intent — each scene with its own intrinsics — is what runs here.)"""
        dev, S = self.device, self.image_size
        with PP.WriterPool() as pool:
            out = []
            first = 0
            for batch in num_to_groups(num_scenes, self.batch_size):
                ids = list(range(first, first + batch))
                first += batch
                K = self._intrinsics(batch)
                K_dev = torch.from_numpy(K).to(dev)
                absolute = np.stack([np.eye(4) for _ in range(batch)]).astype(np.float32)
                param_cond = G.param_vector(K_dev)
                images = self._sample(param_cond, None, ids, 0, noise)
                scene = [PP.native_voxel_down_sample(c, voxel_size).astype(np.float32)
                         for c in G.point_clouds(images, K_dev, None, clip=(0.5, 3.5))]
                img_host = images.cpu().numpy()
                zero = np.zeros((S, S), dtype=np.float32)
                for j, i in enumerate(ids):
                    pool.image01(str(self.samples_folder / f"scene-{i}-sample-0.png"), self._strip(zero, zero, img_host[j, 0]))
                for k in range(1, num_samples):
                    relative = G.random_sample_transform(K, image_size=S)
                    absolute = relative @ absolute
                    # the accumulated clouds moved into the new cameras and z-buffered: one ragged launch (sd:2168-2190)
                    rpj, hit = G.project_clouds(scene, absolute, K, S, dev, depth_scale=0.1)
                    cond = torch.cat([rpj, hit.to(rpj.dtype)], dim=1) * 2 - 1
                    last = images
                    images = self._sample(param_cond, cond, ids, k, noise, has_refine_step)
                    new = G.point_clouds(images, K_dev, torch.from_numpy(absolute).to(dev), clip=(0.5, 3.5))
                    l_h, r_h, n_h = last.cpu().numpy(), rpj.cpu().numpy(), images.cpu().numpy()
                    for j, i in enumerate(ids):
                        pool.image01(str(self.samples_folder / f"scene-{i}-sample-{k}.png"),
                                     self._strip(l_h[j, 0], r_h[j, 0], n_h[j, 0]))
                        merged = np.concatenate([scene[j].astype(np.float64), new[j]], axis=0)
                        scene[j] = PP.native_voxel_down_sample(merged, voxel_size).astype(np.float32)
                for j, i in enumerate(ids):
                    pool.cloud(str(self.samples_folder / f"scene-{i}.ply"), scene[j], None, crop=False, voxel=0.025)
                out.append(scene)
            pool.wait()
        return out
