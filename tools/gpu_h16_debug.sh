#!/bin/bash
# h16 bisect: the numerics probe of the packed-f16 SiLU, then the dim-64 U-Net against G8 under every PRG_H16 mask, taps included
cd $GRAFT_REPO_ROOT
tools/micro/h16_silu_probe
for M in 0 1 2 3 5 7; do
  echo "== PRG_H16=$M"
  PRG_H16=$M python - <<'PY'
import numpy as np, torch, os, sys
sys.path.insert(0, os.getcwd())
from pointreggpt_amd import weights as W
from pointreggpt_amd.unet import Unet
g = np.load("tests/golden/G8_unet_dim64.npz")
sd = W.synth_state_dict(W.unet_config(64), 8)
net = Unet(64, dtype="bf16").load_state_dict(sd)
net.set_taps(True)
D = lambda a: torch.from_numpy(np.asarray(a)).cuda()
y = net(D(g["x"]), D(g["t"]), D(g["pc"]))
d = (y.cpu().double().numpy() - g["y"])
print("y: nan", int(np.isnan(d).sum()), "max", float(np.nanmax(np.abs(d))), "mean", float(np.nanmean(np.abs(d))))
for k in ("init_conv", "down0_block0", "down0_attn", "down0_out", "mid_attn", "up0_out", "final_res"):
    t = net.get_tap(k, 2).cpu().double().numpy()
    print("  tap", k, "nan", int(np.isnan(t).sum()), "absmax", float(np.nanmax(np.abs(t))))
PY
done
