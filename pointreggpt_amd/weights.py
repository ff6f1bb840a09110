"""Parameter layout, deterministic initialiser and checkpoint reading for the two U-Nets.

The HIP library never sees torch modules: it is handed flat, packed weight arenas.  This file owns the
*names and shapes* of the parameters (they must equal the reference's ``state_dict`` keys so released
checkpoints load unchanged) and a counter-based deterministic initialiser used for synthetic weights
(tests, goldens, bench: there is no network for real checkpoints).

Reference layout being mirrored (names only, no code):
  Unet      /root/reference/denoising_diffusion_pytorch/successive_ddnm_diffusion.py:802-918
  MaskUnet  /root/reference/depth_correction_pytorch/depth_correction.py:807-869
  checkpoint dicts  successive_ddnm_diffusion.py:1681-1699, 2307-2318 ; depth_correction.py:1189-1207
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

HEADS = 4
DIM_HEAD = 32
HIDDEN = HEADS * DIM_HEAD  # 128


@dataclass(frozen=True)
class UnetConfig:
    """Static architecture of one U-Net (conditional denoiser or depth-correction mask net)."""
    dim: int = 64
    dim_mults: Tuple[int, ...] = (1, 2, 4, 8)
    in_channels: int = 1          # Unet: 1 ; MaskUnet: 3 (DepthAugment output)
    out_channels: int = 1
    conditional: bool = True      # Unet: time + camera-intrinsic conditioning ; MaskUnet: none
    param_cond_dim: int = 4
    groups: int = 8
    sigmoid_out: bool = False     # MaskUnet ends in Sigmoid

    @property
    def emb_dim(self) -> int:
        return self.dim * 4

    @property
    def dims(self):
        return [self.dim] + [self.dim * m for m in self.dim_mults]

    @property
    def in_out(self):
        d = self.dims
        return list(zip(d[:-1], d[1:]))


def unet_config(dim=64, dim_mults=(1, 2, 4, 8)) -> UnetConfig:
    return UnetConfig(dim=dim, dim_mults=tuple(dim_mults), in_channels=1, conditional=True)


def maskunet_config(dim=64, dim_mults=(1, 2, 4, 8)) -> UnetConfig:
    return UnetConfig(dim=dim, dim_mults=tuple(dim_mults), in_channels=3, conditional=False,
                      sigmoid_out=True)


# --------------------------------------------------------------------------------------------
# parameter spec
# --------------------------------------------------------------------------------------------

def _resblock(spec, prefix, cin, cout, cfg: UnetConfig):
    if cfg.conditional:
        spec[f"{prefix}.mlp.1.weight"] = (2 * cout, 2 * cfg.emb_dim)
        spec[f"{prefix}.mlp.1.bias"] = (2 * cout,)
    for blk, ci in (("block1", cin), ("block2", cout)):
        spec[f"{prefix}.{blk}.proj.weight"] = (cout, ci, 3, 3)
        spec[f"{prefix}.{blk}.proj.bias"] = (cout,)
        spec[f"{prefix}.{blk}.norm.weight"] = (cout,)
        spec[f"{prefix}.{blk}.norm.bias"] = (cout,)
    if cin != cout:
        spec[f"{prefix}.res_conv.weight"] = (cout, cin, 1, 1)
        spec[f"{prefix}.res_conv.bias"] = (cout,)


def _attn(spec, prefix, c, linear: bool):
    spec[f"{prefix}.fn.fn.to_qkv.weight"] = (3 * HIDDEN, c, 1, 1)
    if linear:
        spec[f"{prefix}.fn.fn.to_out.0.weight"] = (c, HIDDEN, 1, 1)
        spec[f"{prefix}.fn.fn.to_out.0.bias"] = (c,)
        spec[f"{prefix}.fn.fn.to_out.1.g"] = (1, c, 1, 1)
    else:
        spec[f"{prefix}.fn.fn.to_out.weight"] = (c, HIDDEN, 1, 1)
        spec[f"{prefix}.fn.fn.to_out.bias"] = (c,)
    spec[f"{prefix}.fn.norm.g"] = (1, c, 1, 1)


def param_spec(cfg: UnetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered name -> shape map, identical to the reference module's ``state_dict()`` keys."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    d0 = cfg.dim
    s["init_conv.weight"] = (d0, cfg.in_channels, 7, 7)
    s["init_conv.bias"] = (d0,)
    if cfg.conditional:
        e = cfg.emb_dim
        s["time_mlp.1.weight"] = (e, d0)
        s["time_mlp.1.bias"] = (e,)
        s["time_mlp.3.weight"] = (e, e)
        s["time_mlp.3.bias"] = (e,)
        s["param_mlp.0.weight"] = (e, cfg.param_cond_dim)
        s["param_mlp.0.bias"] = (e,)
        s["param_mlp.2.weight"] = (e, e)
        s["param_mlp.2.bias"] = (e,)
    n = len(cfg.in_out)
    for i, (ci, co) in enumerate(cfg.in_out):
        _resblock(s, f"downs.{i}.0", ci, ci, cfg)
        _resblock(s, f"downs.{i}.1", ci, ci, cfg)
        _attn(s, f"downs.{i}.2", ci, linear=True)
        k = 3 if i == n - 1 else 4
        s[f"downs.{i}.3.weight"] = (co, ci, k, k)
        s[f"downs.{i}.3.bias"] = (co,)
    for i, (ci, co) in enumerate(reversed(cfg.in_out)):
        _resblock(s, f"ups.{i}.0", co + ci, co, cfg)
        _resblock(s, f"ups.{i}.1", co + ci, co, cfg)
        _attn(s, f"ups.{i}.2", co, linear=True)
        last = i == n - 1
        name = f"ups.{i}.3" if last else f"ups.{i}.3.1"   # Upsample = Sequential(nn.Upsample, Conv2d)
        s[f"{name}.weight"] = (ci, co, 3, 3)
        s[f"{name}.bias"] = (ci,)
    # (the reference registers both ModuleLists before the middle blocks, hence this key order)
    mid = cfg.dims[-1]
    _resblock(s, "mid_block1", mid, mid, cfg)
    _attn(s, "mid_attn", mid, linear=False)
    _resblock(s, "mid_block2", mid, mid, cfg)
    _resblock(s, "final_res_block", 2 * d0, d0, cfg)
    fc = "final_conv.0" if cfg.sigmoid_out else "final_conv"   # MaskUnet: Sequential(Conv2d, Sigmoid)
    s[f"{fc}.weight"] = (cfg.out_channels, d0, 1, 1)
    s[f"{fc}.bias"] = (cfg.out_channels,)
    return s


def num_params(cfg: UnetConfig) -> int:
    return int(sum(int(np.prod(v)) for v in param_spec(cfg).values()))


# --------------------------------------------------------------------------------------------
# deterministic initialiser
# --------------------------------------------------------------------------------------------

# Output calibration of the synthetic denoiser (`calibrated=True`).  The unit-gain initialiser ends in an O(5) network
# output: every x0 prediction lands on the sampler's [-1,1] clamp and 98 % of the in-painted pixels of a chain finish
# saturated — a constant image that hides precision drift and starves the kernels after the sampler.  Shrinking the
# 1x1 head by CALIBRATED_FINAL_GAIN and centring it with CALIBRATED_FINAL_BIAS keeps the prediction inside the open
# interval (measured with the reference on CPU: 0.1 % of the in-painted pixels saturate after a 50-step DDIM chain at
# 64x64, depth 2-4 m with 0.7-1.5 m spread): the workload every round-3 fixture, test and bench.py use.
CALIBRATED_FINAL_GAIN = 0.2
CALIBRATED_FINAL_BIAS = -0.25
# The same for the synthetic depth-correction net (MaskUnet, thresholded at 0.99 = logit 4.595 by its caller, sd:2564-2581):
# an untrained head never crosses the threshold, a plain bias shift makes it keep everything or (round 2's final_bias=6:
# 0.7 % kept) nothing.  Gain 0.2 / bias 5.4 puts the threshold inside the logit distribution of the seed-2 net bench.py uses
# (raw logits -3.3 +- 0.8 on reprojected synthetic depth): ~85 % of the pixels are kept, in a spatially varying pattern.
CALIBRATED_MASK_GAIN = 0.2
CALIBRATED_MASK_BIAS = 5.4


def synth_state_dict(cfg: UnetConfig, seed: int = 0, *, final_bias: float | None = None,
                     final_gain: float | None = None, calibrated: bool = False,
                     dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Counter-based synthetic weights: parameter i of the spec is drawn from Philox(key=seed, counter=i).

    Distribution: conv / linear weights ~ U(-a, a) with a = sqrt(3 / fan_in) (unit-gain variance
    scaling, so activations neither vanish nor explode through ~100 layers); biases ~ U(-0.1, 0.1);
    norm gains ~ 1 + U(-0.2, 0.2) and norm biases ~ U(-0.1, 0.1) so that every affine term is exercised
    by the parity tests.  ``final_bias`` overrides the last conv's bias (MaskUnet thresholds at 0.99:
    an untrained net never crosses it unless the logit is shifted); ``final_gain`` multiplies the last conv's weight;
    ``calibrated=True`` = the two CALIBRATED_* constants above (a denoiser whose x0 prediction stays inside (-1, 1)).
    """
    if calibrated:
        g0, b0 = (CALIBRATED_MASK_GAIN, CALIBRATED_MASK_BIAS) if cfg.sigmoid_out else (CALIBRATED_FINAL_GAIN, CALIBRATED_FINAL_BIAS)
        final_gain = g0 if final_gain is None else final_gain
        final_bias = b0 if final_bias is None else final_bias
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for i, (name, shape) in enumerate(param_spec(cfg).items()):
        g = np.random.Generator(np.random.Philox(key=int(seed) & 0xFFFFFFFFFFFFFFFF, counter=[0, 0, 0, i]))
        u = g.random(size=shape, dtype=np.float64) * 2.0 - 1.0
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "g" or name.endswith("norm.weight"):
            v = 1.0 + 0.2 * u
        elif leaf == "bias":
            v = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            v = np.sqrt(3.0 / fan_in) * u
        out[name] = torch.from_numpy(v.astype(np.float32)).to(dtype)
    fc = "final_conv.0" if cfg.sigmoid_out else "final_conv"
    if final_gain is not None:
        out[fc + ".weight"] = (out[fc + ".weight"].to(torch.float32) * np.float32(final_gain)).to(dtype)
    if final_bias is not None:
        out[fc + ".bias"] = torch.full_like(out[fc + ".bias"], float(final_bias))
    return out


def round_to_bf16(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """Weights rounded to bf16 and widened back to fp32 (what the bf16 arena actually holds)."""
    return OrderedDict((k, v.to(torch.bfloat16).to(torch.float32)) for k, v in sd.items())


# --------------------------------------------------------------------------------------------
# checkpoint reading (released weights; SURVEY §5 'checkpoint / resume')
# --------------------------------------------------------------------------------------------

def _strip(sd: Dict[str, torch.Tensor], prefixes: Iterable[str]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
        out[k] = v
    return out


def unet_state_from_checkpoint(data: dict, cfg: UnetConfig) -> "OrderedDict[str, torch.Tensor]":
    """Extract the sampling U-Net's weights from a diffusion checkpoint dict.

    The reference samples from the EMA copy (successive_ddnm_diffusion.py:2572) whose keys are stored
    under ``data['ema']`` with an ``ema_model.`` prefix and a further ``model.`` prefix for the U-Net
    inside GaussianDiffusion (ema_pytorch 0.2.2 layout: not verifiable here, so both that layout and
    the plain ``data['model']`` fallback are accepted).
    """
    spec = param_spec(cfg)
    candidates = []
    if isinstance(data.get("ema"), dict):
        candidates.append(_strip(data["ema"], ("ema_model.", "model.")))
    if isinstance(data.get("model"), dict):
        candidates.append(_strip(data["model"], ("model.",)))
    candidates.append(_strip(data, ("ema_model.", "model.")))
    for cand in candidates:
        if all(k in cand for k in spec):
            out = OrderedDict()
            for k, shape in spec.items():
                t = cand[k].detach().to(torch.float32).cpu()
                if tuple(t.shape) != tuple(shape):
                    raise ValueError(f"checkpoint tensor {k} has shape {tuple(t.shape)}, expected {shape}")
                out[k] = t
            return out
    raise KeyError("checkpoint does not contain the U-Net parameters under 'ema' or 'model'")


def maskunet_state_from_checkpoint(data: dict, cfg: UnetConfig) -> "OrderedDict[str, torch.Tensor]":
    """depth_correction_results/model-best.pt holds the net under ``ckpt['model']``."""
    return unet_state_from_checkpoint({"model": data.get("model", data)}, cfg)
