#!/usr/bin/env python
"""Error budget of the reduced-precision modes (round-3 VERDICT item 1b): which rounding costs what, in metres.

The CPU oracle (bit-exact to the reference in fp32) is re-run with ONE kind of rounding injected at a time:

  operands   the two operands of a contraction are rounded before it (products and sums stay float32) — what an MFMA input
             format does.  `conv3x3` = only the 3x3 / 4x4 / 7x7 convolutions, `all` = also 1x1 convs and the attention einsums;
  storage    every tensor a layer writes to HBM (conv outputs, ResnetBlock / attention block outputs) is rounded to bf16.

Formats: bf16 (8 significant bits), f16 (11), bf16x2 = hi + lo bf16 split (16: "bf16x3", three MFMAs), f16x2 = hi + lo f16 split
(22: the `f16x3` mode of csrc/conv_split.hip, three MFMAs).  Measured against the float64 twin / the reference on
  G13  one U-Net evaluation @128x128 (output, O(5)),  G14  8-step ancestral chain @128x128,
  G19  the 1000-step ancestral chain @64x64 (normalised depth x 10 = metres; the north star is 1e-4 m point-XYZ).

  python tools/precision_budget.py [g13] [g14] [g19] > profiles/r04_precision_budget.txt      (CPU, ~25 min with g19)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import diffusion as OD  # noqa: E402
from oracle import unet as OU  # noqa: E402
from pointreggpt_amd import weights as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
F_REAL, TORCH_REAL = OU.F, OU.torch


def q_bf16(t):
    return t.bfloat16().float()


def q_f16(t):
    return t.half().float()


def q_bf16x2(t):
    hi = t.bfloat16().float()
    return hi + (t - hi).bfloat16().float()


def q_f16x2(t):
    hi = t.half().float()
    return hi + (t - hi).half().float()


ident = lambda t: t


class Cfg:
    op3 = ident       # operands of the spatial convolutions
    op1 = ident       # operands of 1x1 convolutions and attention einsums
    st = ident        # storage of layer outputs


class FShim:
    def __getattr__(self, k):
        return getattr(F_REAL, k)

    @staticmethod
    def conv2d(x, w, b=None, **kw):
        q = Cfg.op1 if w.shape[-1] == 1 else Cfg.op3
        return Cfg.st(F_REAL.conv2d(q(x), q(w), b, **kw))


class TorchShim:
    def __getattr__(self, k):
        return getattr(TORCH_REAL, k)

    @staticmethod
    def einsum(eq, a, b):
        return TORCH_REAL.einsum(eq, Cfg.op1(a), Cfg.op1(b))


_rb, _pr = OU.resnet_block, OU.prenorm_residual


def install():
    OU.F, OU.torch = FShim(), TorchShim()
    OU.resnet_block = lambda *a, **k: Cfg.st(_rb(*a, **k))
    OU.prenorm_residual = lambda *a, **k: Cfg.st(_pr(*a, **k))


VARIANTS = [
    ("fp32 (the reference's own arithmetic)", ident, ident, ident),
    ("operands bf16, conv3x3 only", q_bf16, ident, ident),
    ("operands bf16, all contractions", q_bf16, q_bf16, ident),
    ("storage bf16 only", ident, ident, q_bf16),
    ("operands + storage bf16 (~ the bf16 mode)", q_bf16, q_bf16, q_bf16),
    ("operands f16 (11 bits), all contractions", q_f16, q_f16, ident),
    ("operands bf16x2 split (16 bits, 3 MFMAs)", q_bf16x2, q_bf16x2, ident),
    ("operands f16x2 split (22 bits, 3 MFMAs) = f16x3", q_f16x2, q_f16x2, ident),
    ("operands f16x2 split, conv only; attention fp32", q_f16x2, ident, ident),
]


def run(name, fn):
    print(f"\n== {name}")
    print(f"{'variant':52s} {'max':>11s} {'mean':>11s}   unit")
    for label, o3, o1, st in VARIANTS:
        Cfg.op3, Cfg.op1, Cfg.st = o3, o1, st
        t0 = time.time()
        mx, mean, unit = fn()
        print(f"{label:52s} {mx:11.3e} {mean:11.3e}   {unit}   ({time.time() - t0:.0f} s)", flush=True)


def g13():
    g = dict(np.load(os.path.join(GOLD, "G13_unet_dim64_128.npz")))
    sd = W.synth_state_dict(W.unet_config(64), 13)
    x, t, pc = (torch.from_numpy(g[k]) for k in ("x", "t", "pc"))

    def fn():
        y = OU.unet_forward(sd, x, t, pc).double().numpy()
        e = np.abs(y - g["y64"])
        return float(e.max()), float(e.mean()), "U-Net output, O(5), vs float64 twin"
    run("G13: one evaluation @128x128 (reference itself: max 4.9e-6)", fn)


def g14():
    g = dict(np.load(os.path.join(GOLD, "G14_chain8_dim64_128.npz")))
    sd = W.synth_state_dict(W.unet_config(64), 14)
    sch = OD.schedule(8)
    pc, cond, nz = (torch.from_numpy(g[k]) for k in ("pc", "cond", "noise"))
    known = (g["cond"][:, 1:2] + 1) * 0.5 > 0.5

    def fn():
        den = lambda x, t, c: OU.unet_forward(sd, x, t, c)
        out = OD.sample(sch, den, pc, cond, 128, OD.stored_noise(nz)).double().numpy()
        e = np.abs(out - g["out64"])[~known] * 10
        return float(e.max()), float(e.mean()), "m, in-painted depth vs float64 twin"
    run("G14: 8-step ancestral chain @128x128 (reference itself: 2.6e-5 m)", fn)


def g19():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import regenerate_chain_noise
    g = dict(np.load(os.path.join(GOLD, "G19_chain1000_ancestral_64.npz")))
    sd = W.synth_state_dict(W.unet_config(64), int(g["wseed"]), calibrated=True)
    sch = OD.schedule(1000)
    pc, cond = torch.from_numpy(g["pc"]), torch.from_numpy(g["img_cond"])
    nz = regenerate_chain_noise(g)
    known = (g["img_cond"][:, 1:2] + 1) * 0.5 > 0.5

    def fn():
        den = lambda x, t, c: OU.unet_forward(sd, x, t, c)
        out = OD.sample(sch, den, pc, cond, 64, OD.stored_noise(nz)).double().numpy()
        e = np.abs(out - g["sampled"].astype(np.float64))[~known] * 10
        return float(e.max()), float(e.mean()), "m, in-painted depth vs the reference"
    run("G19: 1000-step ancestral DDNM chain @64x64 (north star 1e-4 m; reference 1-vs-8 threads 4.9e-6 m)", fn)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("BUDGET_THREADS", "8")))
    install()
    which = set(sys.argv[1:]) or {"g13", "g14", "g19"}
    print("precision budget: CPU oracle with one rounding injected at a time (tools/precision_budget.py)")
    for k, f in (("g13", g13), ("g14", g14), ("g19", g19)):
        if k in which:
            f()
