"""Oracle: the conditional denoiser U-Net and the depth-correction mask U-Net as pure functions of a
state dict (TEST INFRASTRUCTURE — see oracle/__init__.py).  torch-CPU fp32, same operator order as the
reference so that oneDNN/ATen arithmetic is shared with it.

sd = /root/reference/denoising_diffusion_pytorch/successive_ddnm_diffusion.py
dc = /root/reference/depth_correction_pytorch/depth_correction.py
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

HEADS, DIM_HEAD = 4, 32


def ws_weight(w: torch.Tensor) -> torch.Tensor:
    """Weight standardisation per output channel, biased variance, eps 1e-5 in fp32 (sd:601-616)."""
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return (w - mean) * (var + 1e-5).rsqrt()


def channel_layernorm(x, g):
    """Per-pixel normalisation over channels, gain only (sd:619-628)."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def sinusoidal(t: torch.Tensor, dim: int) -> torch.Tensor:
    """[sin(t f_i), cos(t f_i)], f_i = exp(-i ln(1e4)/(dim/2-1)) (sd:645-657)."""
    half = dim // 2
    f = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    a = t[:, None] * f[None, :]
    return torch.cat([a.sin(), a.cos()], dim=-1)


def block(p, pre, x, groups, scale_shift=None):
    """WS-conv3x3 -> GroupNorm -> optional (scale+1, shift) -> SiLU (sd:681-697)."""
    x = F.conv2d(x, ws_weight(p[pre + ".proj.weight"]), p[pre + ".proj.bias"], padding=1)
    x = F.group_norm(x, groups, p[pre + ".norm.weight"], p[pre + ".norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def resnet_block(p, pre, x, groups, cond=None):
    """Two blocks plus a (1x1 when widths differ) skip; conditioning enters block 1 only (sd:700-734)."""
    ss = None
    if cond is not None and (pre + ".mlp.1.weight") in p:
        e = F.linear(F.silu(cond), p[pre + ".mlp.1.weight"], p[pre + ".mlp.1.bias"])
        ss = e[:, :, None, None].chunk(2, dim=1)
    h = block(p, pre + ".block1", x, groups, ss)
    h = block(p, pre + ".block2", h, groups)
    if (pre + ".res_conv.weight") in p:
        x = F.conv2d(x, p[pre + ".res_conv.weight"], p[pre + ".res_conv.bias"])
    return h + x


def linear_attention(p, pre, x):
    """O(N) attention: softmax(q) over head-dim, softmax(k) over pixels, 32x32 context per head (sd:737-769)."""
    B, C, H, W = x.shape
    q, k, v = F.conv2d(x, p[pre + ".to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(B, HEADS, DIM_HEAD, H * W) for t in (q, k, v))
    q = q.softmax(dim=-2) * DIM_HEAD ** -0.5
    k = k.softmax(dim=-1)
    v = v / (H * W)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B, HEADS * DIM_HEAD, H, W)
    out = F.conv2d(out, p[pre + ".to_out.0.weight"], p[pre + ".to_out.0.bias"])
    return channel_layernorm(out, p[pre + ".to_out.1.g"])


def full_attention(p, pre, x):
    """softmax(q k^T / sqrt(32)) v over all pixels of the bottleneck (sd:772-796)."""
    B, C, H, W = x.shape
    q, k, v = F.conv2d(x, p[pre + ".to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(B, HEADS, DIM_HEAD, H * W) for t in (q, k, v))
    sim = torch.einsum("bhdi,bhdj->bhij", q * DIM_HEAD ** -0.5, k)
    out = torch.einsum("bhij,bhdj->bhid", sim.softmax(dim=-1), v)
    out = out.permute(0, 1, 3, 2).reshape(B, HEADS * DIM_HEAD, H, W)
    return F.conv2d(out, p[pre + ".to_out.weight"], p[pre + ".to_out.bias"])


def prenorm_residual(p, pre, x, fn):
    """x + fn(LayerNorm(x))  (sd:583-589, sd:631-639)."""
    return fn(p, pre + ".fn.fn", channel_layernorm(x, p[pre + ".fn.norm.g"])) + x


def depth_augment(depth: torch.Tensor) -> torch.Tensor:
    """(B,1,H,W) -> (B,3,H,W): depth, 3x3 min over non-zero neighbours (raw 3x3 min when the window has
    none), and their difference (dc:577-604)."""
    holes_inf = torch.where(depth == 0, torch.full_like(depth, float("inf")), depth)
    m_valid = -F.max_pool2d(-holes_inf, 3, 1, 1)
    m_raw = -F.max_pool2d(-depth, 3, 1, 1)
    m = torch.where(m_valid.isinf(), m_raw, m_valid)
    return torch.cat([depth, m, m - depth], dim=1)


def _trunk(p: Dict[str, torch.Tensor], x, cond, groups, taps):
    n_levels = sum(1 for k in p if k.startswith("downs.") and k.endswith(".3.weight"))
    r = x.clone()
    skips = []
    for i in range(n_levels):
        x = resnet_block(p, f"downs.{i}.0", x, groups, cond)
        skips.append(x)
        if taps is not None and i == 0:
            taps["down0_block0"] = x
        x = resnet_block(p, f"downs.{i}.1", x, groups, cond)
        x = prenorm_residual(p, f"downs.{i}.2", x, linear_attention)
        if taps is not None and i == 0:
            taps["down0_attn"] = x
        skips.append(x)
        w = p[f"downs.{i}.3.weight"]
        x = F.conv2d(x, w, p[f"downs.{i}.3.bias"], stride=2 if w.shape[-1] == 4 else 1, padding=1)
        if taps is not None and i == 0:
            taps["down0_out"] = x
    x = resnet_block(p, "mid_block1", x, groups, cond)
    x = prenorm_residual(p, "mid_attn", x, full_attention)
    if taps is not None:
        taps["mid_attn"] = x
    x = resnet_block(p, "mid_block2", x, groups, cond)
    for i in range(n_levels):
        x = resnet_block(p, f"ups.{i}.0", torch.cat([x, skips.pop()], 1), groups, cond)
        x = resnet_block(p, f"ups.{i}.1", torch.cat([x, skips.pop()], 1), groups, cond)
        x = prenorm_residual(p, f"ups.{i}.2", x, linear_attention)
        if f"ups.{i}.3.1.weight" in p:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, p[f"ups.{i}.3.1.weight"], p[f"ups.{i}.3.1.bias"], padding=1)
            if taps is not None and i == 0:
                taps["up0_out"] = x
        else:
            x = F.conv2d(x, p[f"ups.{i}.3.weight"], p[f"ups.{i}.3.bias"], padding=1)
    x = resnet_block(p, "final_res_block", torch.cat([x, r], 1), groups, cond)
    if taps is not None:
        taps["final_res"] = x
    return x


@torch.no_grad()
def unet_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, time: torch.Tensor, param_cond: torch.Tensor,
                 groups: int = 8, taps: Optional[dict] = None) -> torch.Tensor:
    """Denoiser: (B,1,S,S), (B,) integer timesteps, (B,4) [fx,fy,cx,cy] -> (B,1,S,S)  (sd:920-964).

    Conditioning vector = cat[time_mlp(t), param_mlp(K)], both Linear-GELU(erf)-Linear (sd:845-856)."""
    dim = p["init_conv.weight"].shape[0]
    pe = F.linear(F.gelu(F.linear(param_cond, p["param_mlp.0.weight"], p["param_mlp.0.bias"])),
                  p["param_mlp.2.weight"], p["param_mlp.2.bias"])
    x = F.conv2d(x, p["init_conv.weight"], p["init_conv.bias"], padding=3)
    if taps is not None:
        taps["init_conv"] = x
    te = sinusoidal(time.to(torch.float32) if not time.is_floating_point() else time, dim)
    te = F.linear(F.gelu(F.linear(te, p["time_mlp.1.weight"], p["time_mlp.1.bias"])),
                  p["time_mlp.3.weight"], p["time_mlp.3.bias"])
    x = _trunk(p, x, torch.cat([te, pe], dim=-1), groups, taps)
    return F.conv2d(x, p["final_conv.weight"], p["final_conv.bias"])


@torch.no_grad()
def maskunet_forward(p: Dict[str, torch.Tensor], depth: torch.Tensor, groups: int = 8,
                     taps: Optional[dict] = None, return_logits: bool = False) -> torch.Tensor:
    """Depth correction: (B,1,S,S) normalised depth -> (B,1,S,S) keep-probability (dc:871-906)."""
    x = depth_augment(depth)
    if taps is not None:
        taps["augment"] = x
    x = F.conv2d(x, p["init_conv.weight"], p["init_conv.bias"], padding=3)
    x = _trunk(p, x, None, groups, taps)
    logits = F.conv2d(x, p["final_conv.0.weight"], p["final_conv.0.bias"])
    return logits if return_logits else torch.sigmoid(logits)
