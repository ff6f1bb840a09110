"""Dump the float64 oracle chain of the G12 pair (every transition: DDIM scalars sc<k>, network output u<k>, state x<k>) and,
given the dump of another host, print where the two diverge.  Usage: python tools/chain_host_compare.py <repo root> <out.npz> [other.npz]
Finding (round 2): the FIRST difference between the build container and the GPU box host is sc0 — the float32 DDIM
coefficients sigma / c of the first transition (8.7e-6 relative) — not the network."""
import sys, os, numpy as np, torch
ROOT = sys.argv[1]; sys.path.insert(0, ROOT)
from oracle import diffusion as OD, unet as OU, geometry as OG
from pointreggpt_amd import weights as W
g = np.load(os.path.join(ROOT, "tests", "golden", "G12_end_to_end_64.npz"))
sd64 = {k: v.double() for k, v in W.synth_state_dict(W.unet_config(64), 12).items()}
sch = OD.schedule(1000)
pc = OG.param_vector(torch.tensor(g["K"])).double()
cond, noise = torch.tensor(g["img_cond"]).double(), torch.tensor(g["noise"])
den64 = lambda x, t, c: OU.unet_forward(sd64, x.double(), t.double(), c.double())
pairs = OD.ddim_time_pairs(1000, 50); ac = sch["alphas_cumprod"]
img = noise[0].double(); out = {}
for k, (t, tn) in enumerate(pairs):
    eps, x0 = OD.model_predictions(sch, den64, img, t, pc, cond, clip_x_start=True)
    out[f"u{k}"] = x0
    if tn < 0:
        img = x0
    else:
        a, an = ac[t], ac[tn]
        sigma = ((1 - a / an) * (1 - an) / (1 - a)).sqrt(); c = (1 - an - sigma ** 2).sqrt()
        out[f"sc{k}"] = torch.stack([sigma, c, an.sqrt()]).double()
        img = x0 * an.sqrt() + c * eps + sigma * noise[k + 1].double()
    out[f"x{k}"] = img
np.savez(sys.argv[2], **{k: v.numpy() for k, v in out.items()})
if len(sys.argv) > 3:
    ref = np.load(sys.argv[3])
    for k in range(50):
        for nm in ("sc", "u", "x"):
            key = f"{nm}{k}"
            if key in ref.files:
                d = np.abs(ref[key] - out[key].numpy()).max()
                if d > 0 or k < 2: print(key, f"{d:.3e}")
