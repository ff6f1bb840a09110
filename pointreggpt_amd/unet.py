"""Host-side mirror of the reference's two networks, backed by the HIP library.

    Unet(dim=64, param_cond_dim=4, dim_mults=(1,2,4,8), channels=1)   sd:802-964
    MaskUnet(dim=64, dim_mults=(1,2,4,8))                             dc:807-906

Both are thin handles: ``load_state_dict`` flattens a reference-layout state dict into the float32 arena the
C-ABI expects (``prg_unet_create`` standardises / packs / uploads), ``forward`` / ``__call__`` launch the HIP
kernels.  dtype 'bf16' (MFMA bf16, fp32 accumulate; the throughput mode), 'fp32' (exact-f32 MFMA; the parity mode)
or 'mxfp8' (bf16 storage, 3x3 convs on the block-scaled fp8 MFMA: BASELINE configs[4]).  sd / dc = the reference's successive_ddnm_diffusion.py / depth_correction.py.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .weights import UnetConfig, maskunet_config, param_spec, synth_state_dict, unet_config

_DTYPES = {"fp32": _lib.PRG_F32, "f32": _lib.PRG_F32, "float32": _lib.PRG_F32, "bf16": _lib.PRG_BF16,
           "bfloat16": _lib.PRG_BF16, "mxfp8": _lib.PRG_MXFP8, "f16x3": _lib.PRG_F16X3}


def _cfg_c(cfg: UnetConfig) -> _lib.UnetConfigC:
    c = _lib.UnetConfigC()
    c.dim, c.n_levels = cfg.dim, len(cfg.dim_mults)
    for i, m in enumerate(cfg.dim_mults):
        c.dim_mults[i] = m
    c.in_channels, c.conditional = cfg.in_channels, int(cfg.conditional)
    c.param_cond_dim, c.groups, c.sigmoid_out = cfg.param_cond_dim, cfg.groups, int(cfg.sigmoid_out)
    return c


def flatten_state_dict(cfg: UnetConfig, sd: Dict[str, torch.Tensor]) -> np.ndarray:
    """state dict -> one float32 vector in the reference's state_dict order (what prg_unet_create takes)."""
    parts = []
    for name, shape in param_spec(cfg).items():
        if name not in sd:
            raise KeyError(f"missing parameter {name}")
        t = sd[name].detach().to(torch.float32).cpu().contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"parameter {name}: shape {tuple(t.shape)} != {tuple(shape)}")
        parts.append(t.reshape(-1).numpy())
    return np.ascontiguousarray(np.concatenate(parts))


class _HipNet:
    def __init__(self, cfg: UnetConfig, dtype: str = "bf16"):
        if dtype not in _DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_DTYPES)}")
        self.cfg, self.dtype = cfg, dtype
        self._h: Optional[C.c_void_p] = None
        # objects holding library handles that point INTO this network's handle (samplers: captured graphs bake in its
        # weight and workspace pointers); they are closed before the handle is destroyed
        self._dependents: "weakref.WeakSet" = weakref.WeakSet()
        self._freqs = None
        self.channels = cfg.in_channels if cfg.conditional else 1
        self.out_dim = cfg.out_channels
        self.random_or_learned_sinusoidal_cond = False   # asserted off by GaussianDiffusion (sd:1034)

    # -- weights ---------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        lib = _lib.load()
        _lib.require_gpu()
        flat = flatten_state_dict(self.cfg, sd)
        cc = _cfg_c(self.cfg)
        n = lib.prg_unet_param_count(C.byref(cc))
        if n != flat.size:
            raise _lib.PrgError(f"parameter count mismatch: library expects {n}, state dict has {flat.size}")
        self.close()
        h = C.c_void_p()
        _lib.check(lib.prg_unet_create(C.byref(cc), flat.ctypes.data_as(C.c_void_p), flat.size, _DTYPES[self.dtype],
                                       C.byref(h)), "prg_unet_create")
        self._h = h
        if self.cfg.conditional:
            self.set_time_freqs(self._freqs)
        return self

    def set_time_freqs(self, freqs=None):
        """SinusoidalPosEmb's frequency table (sd:645-657), a float32 `exp` whose last bit matters: one ulp moves a 50-step
        chain by 4e-5.  The reference evaluates it with torch on the device its model lives on
        (`torch.arange(half_dim, device=x.device)`), so:
          * None (default) — torch on THIS HOST'S CPU: what the reference's CPU path, the parity target named by
            BASELINE.json's north star, computes on this machine;
          * "device" — torch on the HIP device: what the reference would compute when it runs on an accelerator;
          * an array — the table of another host (the golden fixtures carry the one of the host that made them)."""
        import math
        half = self.cfg.dim // 2
        if freqs is None or (isinstance(freqs, str) and freqs == "cpu"):
            freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
        elif isinstance(freqs, str):
            if freqs != "device":
                raise ValueError("freqs must be None, 'cpu', 'device' or an array of dim/2 values")
            _lib.require_gpu()
            freqs = torch.exp(torch.arange(half, device="cuda") * -(math.log(10000) / (half - 1))).cpu()
        f = np.ascontiguousarray(np.asarray(freqs, dtype=np.float32).reshape(-1))
        if f.size != half:
            raise ValueError(f"need {half} frequencies")
        self._freqs = f
        if self._h is not None:
            for dep in list(self._dependents):
                dep.close()          # samplers bake the time table built from the old frequencies
            _lib.check(_lib.load().prg_unet_set_time_freqs(self._h, f.ctypes.data_as(C.c_void_p), int(f.size)),
                       "prg_unet_set_time_freqs")
        return self

    def init_synthetic(self, seed: int = 0, **kw):
        return self.load_state_dict(synth_state_dict(self.cfg, seed, **kw))

    def close(self):
        for dep in list(self._dependents):
            dep.close()
        if self._h is not None:
            _lib.load().prg_unet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if self._h is None:
            raise _lib.PrgError("network has no weights: call load_state_dict() / init_synthetic() first")
        return self._h

    def reserve(self, batch: int, size: int):
        _lib.check(_lib.load().prg_unet_reserve(self.handle, int(batch), int(size)), "prg_unet_reserve")

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    # -- debug taps ------------------------------------------------------------------------------
    def set_taps(self, enable: bool):
        _lib.check(_lib.load().prg_unet_set_taps(self.handle, int(enable)))

    def get_tap(self, name: str, batch: int, max_floats: int = 1 << 26) -> torch.Tensor:
        lib = _lib.load()
        buf = torch.empty(max_floats, dtype=torch.float32, device="cuda")
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.prg_unet_get_tap(self.handle, name.encode(), _lib.ptr(buf), buf.numel(), C.byref(c), C.byref(h),
                                        C.byref(w), _lib.stream_ptr()), f"prg_unet_get_tap({name})")
        n = batch * c.value * h.value * w.value
        return buf[:n].reshape(batch, c.value, h.value, w.value).clone()


class Unet(_HipNet):
    """Conditional denoiser.  forward(x (B,1,S,S), time (B,) int64, param_cond (B,4)) -> (B,1,S,S)  (sd:920)."""

    def __init__(self, dim, param_cond_dim=4, dim_mults=(1, 2, 4, 8), channels=1, dtype: str = "bf16"):
        if channels != 1 or param_cond_dim != 4:
            raise ValueError("this path supports channels=1, param_cond_dim=4 (the generator's configuration)")
        super().__init__(unet_config(dim, dim_mults), dtype)
        self.channels = 1
        self.param_cond_dim = param_cond_dim

    def forward(self, x: torch.Tensor, time: torch.Tensor, param_cond: torch.Tensor, img_cond=None) -> torch.Tensor:
        lib = _lib.load()
        x = x.contiguous().to(torch.float32)
        B, Cc, S, S2 = x.shape
        assert Cc == 1 and S == S2
        time = time.to(device=x.device, dtype=torch.int64).contiguous()
        pc = param_cond.to(device=x.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        _lib.check(lib.prg_unet_forward(self.handle, _lib.ptr(x), _lib.ptr(time), _lib.ptr(pc), _lib.ptr(out), B, S,
                                        _lib.stream_ptr()), "prg_unet_forward")
        return out

    __call__ = forward


class MaskUnet(_HipNet):
    """Depth correction.  forward(depth (B,1,S,S) in [0,1]) -> keep-probability (B,1,S,S)  (dc:871)."""

    def __init__(self, dim, dim_mults=(1, 2, 4, 8), dtype: str = "bf16"):
        super().__init__(maskunet_config(dim, dim_mults), dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        x = x.contiguous().to(torch.float32)
        B, Cc, S, S2 = x.shape
        assert Cc == 1 and S == S2
        out = torch.empty_like(x)
        _lib.check(lib.prg_maskunet_forward(self.handle, _lib.ptr(x), _lib.ptr(out), B, S, _lib.stream_ptr()),
                   "prg_maskunet_forward")
        return out

    __call__ = forward
