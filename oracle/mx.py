"""Oracle for the MX-fp8 operand path (TEST INFRASTRUCTURE — see oracle/__init__.py): OCP microscaling restated in numpy.

Not part of the reference (PointRegGPT computes in fp32); it pins the arithmetic BASELINE.json's configs[4] asks of the
build — "fp8 UNet weights on CDNA4 MFMA" — independently of the HIP kernel: OCP MX v1.0 block format with k = 32, element
type FP8 E4M3 (e4m3fn: bias 7, max 448, no infinities), scale type E8M0 (bias 127):

    X = 2^(floor(log2 max|v|) - emax_elem),  emax_elem = 8;   P_i = e4m3(v_i / X), round-to-nearest-even, saturating.
"""
import numpy as np
import torch


def e4m3_round(a: np.ndarray) -> np.ndarray:
    """Nearest e4m3fn value (ties to even), saturating at +-448; float64 in/out."""
    a = np.asarray(a, dtype=np.float64)
    s, m = np.sign(a), np.abs(a)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(m > 0, m, 1.0)))
    e = np.clip(e, -6, 8)                         # subnormals share the exponent of the smallest normal (step 2^-9)
    step = np.exp2(e - 3)
    q = np.rint(m / step) * step                  # np.rint rounds half to even
    return s * np.minimum(q, 448.0)


def mx_quantize_dequantize(v: np.ndarray, axis: int) -> np.ndarray:
    """Values an MX-fp8 tensor holds after quantising `v` in blocks of 32 along `axis` (length a multiple of 32)."""
    v = np.moveaxis(np.asarray(v, dtype=np.float32), axis, -1)
    shp = v.shape
    blk = v.reshape(shp[:-1] + (shp[-1] // 32, 32)).astype(np.float64)
    am = np.abs(blk).max(axis=-1, keepdims=True).astype(np.float32)
    E = ((am.view(np.uint32) >> 23) & 0xFF).astype(np.int64)           # biased exponent = floor(log2 am) + 127
    S = np.maximum(E - 8, 0)                                           # E8M0 byte; am == 0 -> 0
    scale = np.exp2((S - 127).astype(np.float64))
    p = e4m3_round(np.clip(blk / scale, -448.0, 448.0))
    return np.moveaxis((p * scale).reshape(shp), -1, axis)


def conv3x3_mx_reference(x: torch.Tensor, w: torch.Tensor, bias=None) -> torch.Tensor:
    """3x3 / pad 1 convolution of MX-fp8 operands in float64: x (B,Cin,H,W) is first rounded to bf16 (the activation
    storage of the build), then quantised per pixel in blocks of 32 channels; w (Cout,Cin,3,3) per (tap, output channel)
    in blocks of 32 input channels."""
    xb = x.to(torch.bfloat16).to(torch.float32).numpy()
    xq = mx_quantize_dequantize(xb, axis=1)
    wq = mx_quantize_dequantize(w.numpy(), axis=1)
    y = torch.nn.functional.conv2d(torch.from_numpy(xq), torch.from_numpy(wq), None, padding=1)
    if bias is not None:
        y = y + bias.double().view(1, -1, 1, 1)
    return y
