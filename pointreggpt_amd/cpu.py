"""ctypes front-end of libprg_cpu.so (include/prg_cpu.h): the `prg_cpu_*` twins of the hot path with host tensors.

What it is for: BASELINE configs[0] — `generate_dataset.py -start=0 -stop=1` ON CPU (64x64, 50-step DDIM, no GPU), selected
explicitly with `--device cpu` — and a native second opinion beside the torch oracle in the CPU test-suite.  It is never a
fallback: nothing here is imported by the GPU front-ends (`unet.py`, `diffusion.py`, `geometry.py`), which keep raising when
there is no HIP device or no libprg_hip.so.

Mirrors the GPU front-ends' names so `Generator` can run on either: `Unet` / `MaskUnet` / `GaussianDiffusion` and an `ops`
namespace with the geometry calls `Generator` makes.  sd / dc = the reference's successive_ddnm_diffusion.py /
depth_correction.py.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib as _hip
from . import diffusion as _diff
from . import geometry as _G
from .unet import _cfg_c, flatten_state_dict
from .weights import maskunet_config, synth_state_dict, unet_config

LIB_PATH = Path(os.environ.get("PRG_CPU_LIB", Path(__file__).resolve().parent / "libprg_cpu.so"))
_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64

PROTOTYPES = {
    "prg_cpu_last_error": (C.c_char_p, []),
    "prg_cpu_depth2pc": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _F]),
    "prg_cpu_pc2depth": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I]),
    "prg_cpu_project_points_zbuffer": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F]),
    "prg_cpu_reproject_zbuffer": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F]),
    "prg_cpu_unproject_f64": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F]),
    "prg_cpu_depth_augment": (C.c_int, [_P, _P, _I, _I, _I]),
    "prg_cpu_apply_mask": (C.c_int, [_P, _P, _P, _F, _P, _P, _P, _I, _I, _I]),
    "prg_cpu_unet_create": (C.c_int, [C.POINTER(_hip.UnetConfigC), _P, _L, C.POINTER(_P)]),
    "prg_cpu_unet_destroy": (C.c_int, [_P]),
    "prg_cpu_unet_set_time_freqs": (C.c_int, [_P, _P, _I]),
    "prg_cpu_unet_forward": (C.c_int, [_P, _P, _P, _P, _P, _I, _I]),
    "prg_cpu_maskunet_forward": (C.c_int, [_P, _P, _P, _I, _I]),
    "prg_cpu_sampler_run": (C.c_int, [_P, C.POINTER(_hip.StepC), _I, _P, _P, _P, _L, _P, _P, _I, _I]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise _hip.PrgError(f"{LIB_PATH} not found: build it with `make -C pointreggpt_amd/csrc`")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().prg_cpu_last_error()
        raise _hip.PrgError(f"{what or 'libprg_cpu'} failed ({rc}): {msg.decode() if msg else ''}")


def _t(x, dtype=torch.float32) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    if t.is_cuda:
        raise _hip.PrgError("pointreggpt_amd.cpu takes host tensors")
    return t.to(dtype).contiguous()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------------------------------------------
# networks
# ------------------------------------------------------------------------------------------------------------------
class _CpuNet:
    def __init__(self, cfg):
        self.cfg, self.dtype, self._h = cfg, "fp32", None
        self.channels = 1
        self.out_dim = cfg.out_channels
        self.random_or_learned_sinusoidal_cond = False

    def load_state_dict(self, sd):
        lib = load()
        flat = flatten_state_dict(self.cfg, sd)
        self.close()
        h = C.c_void_p()
        cc = _cfg_c(self.cfg)
        check(lib.prg_cpu_unet_create(C.byref(cc), flat.ctypes.data_as(C.c_void_p), flat.size, C.byref(h)), "prg_cpu_unet_create")
        self._h = h
        if self.cfg.conditional:
            self.set_time_freqs(None)
        return self

    def set_time_freqs(self, freqs=None):
        """Same table policy as the GPU front-end: default = torch on this host's CPU (what the reference computes here)."""
        import math
        half = self.cfg.dim // 2
        if freqs is None:
            freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
        f = np.ascontiguousarray(np.asarray(freqs, dtype=np.float32).reshape(-1))
        check(load().prg_cpu_unet_set_time_freqs(self.handle, f.ctypes.data_as(C.c_void_p), int(f.size)))
        return self

    def init_synthetic(self, seed: int = 0, **kw):
        return self.load_state_dict(synth_state_dict(self.cfg, seed, **kw))

    @property
    def handle(self):
        if self._h is None:
            raise _hip.PrgError("network has no weights: call load_state_dict() / init_synthetic() first")
        return self._h

    def close(self):
        if self._h is not None:
            load().prg_cpu_unet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self


class Unet(_CpuNet):
    """forward(x (B,1,S,S), time (B,) int64, param_cond (B,4)) -> (B,1,S,S)  (sd:920), on the host."""

    def __init__(self, dim, param_cond_dim=4, dim_mults=(1, 2, 4, 8), channels=1, dtype: str = "fp32"):
        if dtype not in ("fp32", "f32", "float32"):
            raise ValueError("the CPU twins compute in float32 (float64 accumulation)")
        super().__init__(unet_config(dim, dim_mults))

    def forward(self, x, time, param_cond, img_cond=None):
        x, pc = _t(x), _t(param_cond)
        t = _t(time, torch.int64)
        B, _c, S, _s = x.shape
        out = torch.empty_like(x)
        check(load().prg_cpu_unet_forward(self.handle, _p(x), _p(t), _p(pc), _p(out), B, S), "prg_cpu_unet_forward")
        return out

    __call__ = forward


class MaskUnet(_CpuNet):
    """forward(depth (B,1,S,S) in [0,1]) -> keep-probability (B,1,S,S)  (dc:871), on the host."""

    def __init__(self, dim, dim_mults=(1, 2, 4, 8), dtype: str = "fp32"):
        super().__init__(maskunet_config(dim, dim_mults))

    def forward(self, x):
        x = _t(x)
        B, _c, S, _s = x.shape
        out = torch.empty_like(x)
        check(load().prg_cpu_maskunet_forward(self.handle, _p(x), _p(out), B, S), "prg_cpu_maskunet_forward")
        return out

    __call__ = forward


class GaussianDiffusion(_diff.GaussianDiffusion):
    """The GPU front-end's schedule / transition table (computed with torch on the host exactly like the reference) driving
    prg_cpu_sampler_run."""

    def _sampler(self, batch, refine=False):          # no device handle to keep
        raise _hip.PrgError("the CPU sampler has no persistent handle")

    def close(self):
        pass

    @torch.no_grad()
    def sample(self, *, param_cond, img_cond=None, disable_tqdm=True, has_refine_step=False, noise=None,
               seeds: Optional[Sequence[int]] = None, **_unused):
        pc = _t(param_cond)
        B, S = pc.shape[0], self.image_size
        cond = _t(img_cond) if (img_cond is not None and self.is_ddnm_sampling) else None
        arr, n = self._steps_c(bool(has_refine_step) and cond is not None)
        nz, seed_arr, slabs = None, None, 0
        if noise is not None:
            nz = _t(noise)
            slabs = nz.numel() // (B * S * S)
            assert slabs >= self.n_draws, f"stored noise needs {self.n_draws} draws of shape ({B},1,{S},{S})"
        else:
            seed_arr = (C.c_uint64 * B)(*[int(s) & 0xFFFFFFFFFFFFFFFF for s in (seeds if seeds is not None else range(B))])
        out = torch.empty((B, 1, S, S), dtype=torch.float32)
        check(load().prg_cpu_sampler_run(self.model.handle, arr, n, _p(pc), _p(cond), _p(nz), slabs,
                                         C.cast(seed_arr, C.c_void_p) if seed_arr is not None else None, _p(out), B, S),
              "prg_cpu_sampler_run")
        return out


# ------------------------------------------------------------------------------------------------------------------
# geometry: the calls Generator makes, same names and return types as pointreggpt_amd.geometry
# ------------------------------------------------------------------------------------------------------------------
class ops:
    intrinsic_transform = staticmethod(_G.intrinsic_transform)
    random_sample_pose = staticmethod(_G.random_sample_pose)
    param_vector = staticmethod(_G.param_vector)

    @staticmethod
    def pc2depth_tensor(pc, valid, intrinsic, *, image_size):
        pc, K = _t(pc), _t(intrinsic)
        B, N, _ = pc.shape
        H, W = image_size
        v = None if valid is None else _t(valid, torch.uint8)
        depth = torch.empty((B, 1, H, W), dtype=torch.float32)
        mask = torch.empty((B, 1, H, W), dtype=torch.uint8)
        check(load().prg_cpu_pc2depth(_p(pc), _p(v), _p(K), _p(depth), _p(mask), B, N, H, W), "prg_cpu_pc2depth")
        return depth, mask.view(torch.bool)

    @staticmethod
    def depth2pc_tensor(depth, intrinsic, *, clip=(0, 10), invalid_num=None):
        d, K = _t(depth), _t(intrinsic)
        B, _c, H, W = d.shape
        pc = torch.empty((B, H * W, 3), dtype=torch.float32)
        valid = torch.empty((B, H * W), dtype=torch.uint8)
        lo, hi = (1.0, 0.0) if clip is None else clip
        check(load().prg_cpu_depth2pc(_p(d), _p(K), _p(pc), _p(valid), B, H, W, float(lo), float(hi),
                                      float("nan") if invalid_num is None else float(invalid_num)), "prg_cpu_depth2pc")
        return pc, valid.view(torch.bool)

    @staticmethod
    def reproject_tensor(depth, intrinsic, relative_pose, *, clip=(0, 10), depth_unit=1.0, out_scale=1.0):
        d, K, P = _t(depth), _t(intrinsic), _t(relative_pose)
        B, _c, H, W = d.shape
        out = torch.empty_like(d)
        mask = torch.empty((B, 1, H, W), dtype=torch.uint8)
        check(load().prg_cpu_reproject_zbuffer(_p(d), _p(K), _p(P), _p(out), _p(mask), B, H, W, float(depth_unit), float(clip[0]),
                                               float(clip[1]), float(out_scale)), "prg_cpu_reproject_zbuffer")
        return out, mask.view(torch.bool)

    @staticmethod
    def project_clouds(clouds, poses, intrinsic, image_size, device=None, depth_scale=1.0):
        B = len(clouds)
        offs = np.zeros(B + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(c) for c in clouds])
        pts = torch.from_numpy(np.concatenate([np.asarray(c, dtype=np.float32).reshape(-1, 3) for c in clouds], 0)
                               if offs[-1] else np.zeros((1, 3), np.float32)).contiguous()
        o, P, K = torch.from_numpy(offs), _t(poses), _t(intrinsic)
        S = int(image_size)
        depth = torch.empty((B, 1, S, S), dtype=torch.float32)
        mask = torch.empty((B, 1, S, S), dtype=torch.uint8)
        check(load().prg_cpu_project_points_zbuffer(_p(pts), _p(o), _p(P), _p(K), _p(depth), _p(mask), B, S, S, float(depth_scale)),
              "prg_cpu_project_points_zbuffer")
        return depth, mask.view(torch.bool)

    @staticmethod
    def unproject_f64(depth, intrinsic, pose, *, depth_unit=_G.DEPTH_UNIT_M, clip=(0.5, 10.0)):
        d, K = _t(depth), _t(intrinsic)
        P = None if pose is None else _t(pose)
        B, _c, H, W = d.shape
        xyz = torch.empty((B, H * W, 3), dtype=torch.float64)
        valid = torch.empty((B, H * W), dtype=torch.uint8)
        check(load().prg_cpu_unproject_f64(_p(d), _p(K), _p(P), _p(xyz), _p(valid), B, H, W, float(depth_unit), float(clip[0]),
                                           float(clip[1])), "prg_cpu_unproject_f64")
        return xyz, valid.view(torch.bool)

    @staticmethod
    def point_clouds(depth, intrinsic, pose, *, depth_unit=_G.DEPTH_UNIT_M, clip=(0.5, 10.0)):
        xyz, valid = ops.unproject_f64(depth, intrinsic, pose, depth_unit=depth_unit, clip=clip)
        x, v = xyz.numpy(), valid.numpy()
        return [x[b][v[b]] for b in range(x.shape[0])]

    @staticmethod
    def depth_augment(depth):
        d = _t(depth)
        B, _c, H, W = d.shape
        out = torch.empty((B, 3, H, W), dtype=torch.float32)
        check(load().prg_cpu_depth_augment(_p(d), _p(out), B, H, W), "prg_cpu_depth_augment")
        return out

    @staticmethod
    def apply_mask(prob, depth, hit, thr, want_cond=True):
        pr, d = _t(prob), _t(depth)
        B, _c, H, W = d.shape
        h = None if hit is None else _t(hit, torch.uint8)
        d_out = torch.empty_like(d)
        h_out = torch.empty((B, 1, H, W), dtype=torch.uint8)
        cond = torch.empty((B, 2, H, W), dtype=torch.float32) if want_cond else None
        check(load().prg_cpu_apply_mask(_p(pr), _p(d), _p(h), float(thr), _p(d_out), _p(h_out), _p(cond), B, H, W), "prg_cpu_apply_mask")
        return d_out, h_out.view(torch.bool), cond
