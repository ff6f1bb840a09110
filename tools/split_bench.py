#!/usr/bin/env python
"""Per-launch timing of single convolutions through prg_debug_conv (round 4: the f16x3 kernels).

  rocprofv3 --kernel-trace --output-format csv -d OUT -o r -- python tools/split_bench.py [dtype]
  python tools/split_bench.py --summarise OUT/r_kernel_trace.csv

Shapes = the denoiser's convolutions at B = 64, 128x128."""
import ctypes as C
import csv
import sys

SHAPES = [  # (label, B, Cin, Cout, H, W, K, stride)
    ("L0 64->64 @128", 64, 64, 64, 128, 128, 3, 1),
    ("L0 128->64 @128 (K=1152)", 64, 128, 64, 128, 128, 3, 1),
    ("L1 64->64 @64", 64, 64, 64, 64, 64, 3, 1),
    ("L2 128->128 @32", 64, 128, 128, 32, 32, 3, 1),
    ("L3 256->256 @16", 64, 256, 256, 16, 16, 3, 1),
    ("mid 512->512 @16", 64, 512, 512, 16, 16, 3, 1),
    ("up1 256->256 @32", 64, 256, 256, 32, 32, 3, 1),
    ("qkv 64->384 @128 1x1", 64, 64, 384, 128, 128, 1, 1),
    ("res 512->512 @16 1x1", 64, 512, 512, 16, 16, 1, 1),
    ("down 64->64 @128 4x4s2", 64, 64, 64, 128, 128, 4, 2),
]


def run(dtype_name):
    import numpy as np
    import torch
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pointreggpt_amd import _lib
    lib = _lib.load()
    dt = {"f16x3": _lib.PRG_F16X3, "fp32": _lib.PRG_F32, "bf16": _lib.PRG_BF16}[dtype_name]
    g = torch.Generator().manual_seed(0)
    for label, B, Cin, Cout, H, W, K, st in SHAPES:
        if dt == _lib.PRG_BF16 and K == 1:
            continue
        x = torch.nn.functional.silu(torch.randn((B, Cin, H, W), generator=g)).cuda()
        w = np.ascontiguousarray(torch.randn((Cout, Cin, K, K), generator=g).numpy())
        bias = np.zeros(Cout, dtype=np.float32)
        pad = 0 if K == 1 else 1
        Ho, Wo = (H + 2 * pad - K) // st + 1, (W + 2 * pad - K) // st + 1
        out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device="cuda")
        for _ in range(3):
            _lib.check(lib.prg_debug_conv(_lib.ptr(x), w.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p), _lib.ptr(out),
                                          B, Cin, Cout, H, W, dt, K, st, _lib.stream_ptr()), label)
        torch.cuda.synchronize()


def summarise(path):
    rows = [r for r in csv.DictReader(open(path)) if "conv" in r["Kernel_Name"] and "nchw" not in r["Kernel_Name"].lower()]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i = 0
    for label, B, Cin, Cout, H, W, K, st in SHAPES:
        grp = rows[i:i + 3]
        i += 3
        if len(grp) < 3:
            break
        us = min((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp)
        pad = 0 if K == 1 else 1
        Ho, Wo = (H + 2 * pad - K) // st + 1, (W + 2 * pad - K) // st + 1
        gf = 2.0 * B * Ho * Wo * Cout * Cin * K * K / 1e9
        print(f"{label:26s} {us:9.1f} us  {gf / us * 1e3:7.1f} TFLOP/s (algorithmic)   {grp[0]['Kernel_Name'][:60]}  grid {grp[0].get('Grid_Size_X', '?')}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run(sys.argv[1] if len(sys.argv) > 1 else "f16x3")
