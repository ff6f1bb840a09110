import os, sys, subprocess, numpy as np
root = sys.argv[1]
script = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from pointreggpt_amd import weights as W
from pointreggpt_amd.unet import Unet, MaskUnet
out = sys.argv[1]
res = {}
for (B, S) in [(2, 40), (3, 72), (1, 64), (3, 128), (5, 96), (1, 256), (2, 192)]:
    sd = W.synth_state_dict(W.unet_config(64), 8)
    net = Unet(64, dtype="bf16").load_state_dict(sd)
    g = torch.Generator().manual_seed(B * 1000 + S)
    x = torch.randn((B, 1, S, S), generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    pc = (torch.tensor([[56.8, 57.0, 24.4, 24.0]]) + torch.randn((B, 4), generator=g)).cuda()
    res["u_%%d_%%d" %% (B, S)] = net(x, t, pc).float().cpu().numpy()
    del net
np.savez(out, **res)
''' % root
outs = {}
for name, env in {"fast": {}, "generic": {"PRG_CONV_WS": "0", "PRG_FUSED_ATTN": "0"}}.items():
    o = "/tmp/sweep_%s.npz" % name
    r = subprocess.run([sys.executable, "-c", script, o], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    if r.returncode: print(name, "FAILED", r.stderr[-1500:]); sys.exit(1)
    outs[name] = np.load(o)
for k in outs["fast"].files:
    d = np.abs(outs["fast"][k].astype(np.float64) - outs["generic"][k])
    print(k, "finite", np.isfinite(outs["fast"][k]).all(), "max %.4f mean %.5f absmax %.2f" % (d.max(), d.mean(), np.abs(outs["generic"][k]).max()))
