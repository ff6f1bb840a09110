#!/usr/bin/env python
"""Screen (weight seed, scene seed) candidates for a long-chain fixture ON THE GPU (fp32 mode of the library, seconds per
chain) before spending an hour of CPU on the real reference: fraction of in-painted pixels that end on the [0, 1] clamp.
  python tools/screen_chain_seeds.py 256 250 21 22 23 ...        (size, DDIM steps or 0 = ancestral, seeds)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointreggpt_amd import geometry as G, synthetic, weights as W  # noqa: E402
from pointreggpt_amd.diffusion import GaussianDiffusion  # noqa: E402
from pointreggpt_amd.unet import Unet  # noqa: E402

S, steps = int(sys.argv[1]), int(sys.argv[2]) or None
for seed in map(int, sys.argv[3:]):
    depth, K, pose = synthetic.synth_batch(seed, range(1), S)
    d, Kt, Pt = (torch.from_numpy(a).cuda() for a in (depth, K, pose))
    rpj, hit = G.reproject_tensor(d, Kt, Pt, clip=(0, 10), depth_unit=10.0, out_scale=0.1)
    cond = torch.cat([rpj, hit.to(rpj.dtype)], dim=1) * 2 - 1
    net = Unet(64, dtype="fp32").load_state_dict(W.synth_state_dict(W.unet_config(64), seed, calibrated=True))
    diff = GaussianDiffusion(net, image_size=S, timesteps=1000, sampling_timesteps=steps)
    n = len(diff.step_table()) + 1
    torch.manual_seed(seed * 100)
    nz = torch.stack([torch.randn((1, 1, S, S)) for _ in range(n)]).reshape(-1, 1, 1, S, S).cuda()
    img = diff.sample(param_cond=G.param_vector(Kt), img_cond=cond, noise=nz)
    free = ~hit.bool()
    sat = float(((img <= 0) | (img >= 1))[free].float().mean())
    near = float(((img <= 0.01) | (img >= 0.99))[free].float().mean())
    print(f"seed {seed}: inpainted {float(free.float().mean()):.3f}  saturated {sat:.4f}  within 0.01 of the clamp {near:.4f}  "
          f"depth mean {float(img[free].mean()) * 10:.2f} m std {float(img[free].std()) * 10:.2f} m", flush=True)
    diff.close(); net.close()
