"""Per-transition error budget of the G12 chain (64x64, 50-step DDIM, dim 64): at every transition k the HIP fp32 sampler
is started from the ORACLE's state x_k and its x_{k+1} is compared with the oracle's fp32 and fp64 x_{k+1} — the local
error of one transition, free of chain amplification — next to the accumulated error of the free-running HIP chain.
Run on the GPU box: python tools/chain_budget.py   (uses oracle/: a diagnostic tool, not product code)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import diffusion as OD, geometry as OG, unet as OU  # noqa: E402
from pointreggpt_amd import weights as W  # noqa: E402
from pointreggpt_amd.diffusion import GaussianDiffusion  # noqa: E402
from pointreggpt_amd.unet import Unet  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "G12_end_to_end_64.npz"))
sd32 = W.synth_state_dict(W.unet_config(64), 12)
sd64 = {k: v.double() for k, v in sd32.items()}
sch = OD.schedule(1000)
pc = OG.param_vector(torch.tensor(g["K"]))
cond, noise = torch.tensor(g["img_cond"]), torch.tensor(g["noise"])
pairs = OD.ddim_time_pairs(1000, 50)
ac = sch["alphas_cumprod"]


def step(den, img, k, dt):
    t, tn = pairs[k]
    eps, x0 = OD.model_predictions(sch, den, img, t, pc.to(dt), cond.to(dt), clip_x_start=True)
    if tn < 0:
        return x0, x0
    a, an = ac[t], ac[tn]
    sigma = ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
    c = (1 - an - sigma ** 2).sqrt()
    return x0 * an.sqrt() + c * eps + sigma * noise[k + 1].to(dt), x0


den32 = lambda x, t, c: OU.unet_forward(sd32, x, t, c)
den64 = lambda x, t, c: OU.unet_forward(sd64, x.double(), t.double(), c.double())
net = Unet(64, dtype="fp32").load_state_dict(sd32)
known = OD.cond_mask(cond)
x32 = noise[0].clone()
x64 = noise[0].double()
xh = noise[0].clone()        # free-running HIP chain
table = GaussianDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=50).step_table()
print("k   t | local: hip-o32  hip-o64  o32-o64 | chain: hip-o32  hip-o64  o32-o64 | |u|max free-px")
for k in range(50):
    n32, _ = step(den32, x32, k, torch.float32)
    n64, _ = step(den64, x64, k, torch.float64)
    # local: one HIP transition from the oracle's fp32 state, compared with fp32 / fp64 transitions from the SAME state
    l64, _ = step(den64, x32.double(), k, torch.float64)
    d = GaussianDiffusion(net, image_size=64, timesteps=1000, sampling_timesteps=50)
    rows = [dict(table[k])]
    d.step_table = lambda rows=rows: rows
    nz = torch.stack([x32, noise[k + 1] if k + 1 < len(noise) else torch.zeros_like(x32)])
    lh = d.sample(param_cond=pc.cuda(), img_cond=cond.cuda(), noise=nz.cuda()).cpu() * 2 - 1
    nzf = torch.stack([xh, noise[k + 1] if k + 1 < len(noise) else torch.zeros_like(x32)])
    xh = d.sample(param_cond=pc.cuda(), img_cond=cond.cuda(), noise=nzf.cuda()).cpu() * 2 - 1
    d.close()
    e = lambda a, b: float((a.double() - b.double()).abs().max())
    print(f"{k:2d} {pairs[k][0]:4d} | {e(lh, n32):.2e} {e(lh, l64):.2e} {e(n32, l64):.2e} | {e(xh, n32):.2e} {e(xh, n64):.2e} {e(n32, n64):.2e} | {float(n32[~known].abs().max()):.2f}")
    x32, x64 = n32, n64

# ---- the parity metric (point-XYZ L-infinity, metres) between every pair of: golden (reference on the build container's
# CPU), oracle fp32 on THIS host, oracle fp64, HIP fp32 ----
mask_sd = W.synth_state_dict(W.maskunet_config(64), 13, final_bias=6.0)
thr2 = float(g["thr2"])


def cloud_of(img01):
    prob2 = OU.maskunet_forward(mask_sd, img01.float())
    o = torch.where(prob2 > thr2, img01.float(), torch.zeros_like(img01.float()))
    c = OG.point_cloud(o[0, 0].numpy() * 10, g["K"][0], (0.5, 10.0))
    return OG.inverse_pose_apply(c, g["pose"][0]), (o > 0)


runs = {"golden(container CPU)": torch.tensor(g["sampled"]), "oracle32(this host)": (x32 + 1) * 0.5,
        "oracle64": ((x64 + 1) * 0.5), "hip32": (xh + 1) * 0.5}
clouds = {k: cloud_of(v) for k, v in runs.items()}
names = list(runs)
print("\npairwise: |depth| max (normalised)  /  point-XYZ L-infinity (m)")
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        a, b = names[i], names[j]
        dd = float((runs[a].double() - runs[b].double()).abs().max())
        same = torch.equal(clouds[a][1], clouds[b][1])
        xyz = float(np.abs(clouds[a][0] - clouds[b][0]).max()) if same else float("nan")
        print(f"  {a:24s} vs {b:24s}: {dd:.3e}  /  {xyz:.3e}")
print("torch threads:", torch.get_num_threads(), " cpu count:", os.cpu_count())
