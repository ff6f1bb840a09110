"""Host-side mirror of ``GaussianDiffusion`` (sd:1015-1409) for sampling, backed by the HIP sampler.

The float64 schedule and the per-transition coefficients are computed here with torch exactly as the
reference computes them (tiny, one-off); the library receives them as a ``prg_step`` table and runs the
whole chain on the GPU: U-Net + fused update per transition, captured once as a hipGraph and replayed.
sd = /root/reference/denoising_diffusion_pytorch/successive_ddnm_diffusion.py
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .unet import Unet


def sigmoid_beta_schedule(timesteps: int, start=-3, end=3, tau=1) -> torch.Tensor:
    """Sigmoid alpha-bar schedule, float64, betas clipped to [0, 0.999] (sd:997-1012)."""
    t = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def make_schedule(timesteps: int) -> Dict[str, torch.Tensor]:
    """The float32 buffers GaussianDiffusion registers (sd:1056-1134), from the float64 betas."""
    betas = sigmoid_beta_schedule(timesteps)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.nn.functional.pad(ac[:-1], (1, 0), value=1.0)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    buf = {
        "betas": betas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac), "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac), "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1), "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "loss_weight": ac / (1 - ac),
    }
    return {k: v.to(torch.float32) for k, v in buf.items()}


def ddim_times(total: int, steps: int) -> List[int]:
    """[T-1, ..., -1]: linspace(-1, T-1, steps+1) truncated to int, reversed (sd:1331-1334)."""
    return list(reversed(torch.linspace(-1, total - 1, steps=steps + 1).int().tolist()))



class _ProfileShape(C.Structure):     # include/prg.h: prg_profile_shape
    _fields_ = [("cin", C.c_int32), ("cout", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32), ("ups", C.c_int32),
                ("hout", C.c_int32), ("wout", C.c_int32), ("two_source", C.c_int32), ("prologue", C.c_int32),
                ("mx", C.c_int32), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("flops_executed", C.c_double)]

class GaussianDiffusion:
    """Sampling half of the reference class: same constructor keywords, ``sample(param_cond=, img_cond=)``."""

    def __init__(self, model: Unet, *, image_size, timesteps=1000, sampling_timesteps=None, loss_type="l1",
                 objective="pred_x0", beta_schedule="sigmoid", ddim_sampling_eta=1.0, is_ddnm_sampling=True,
                 ddnm_sampling_dropout=0.0):
        if objective != "pred_x0" or beta_schedule != "sigmoid":
            raise ValueError("this path implements the generator's configuration: objective='pred_x0', "
                             "beta_schedule='sigmoid' (generate_dataset.py:34-44)")
        if ddnm_sampling_dropout != 0.0:
            raise ValueError("ddnm_sampling_dropout must be 0 (the generator never enables it)")
        assert not model.random_or_learned_sinusoidal_cond and model.channels == model.out_dim
        self.model, self.channels, self.image_size = model, model.channels, int(image_size)
        self.objective, self.is_ddnm_sampling = objective, bool(is_ddnm_sampling)
        self.num_timesteps = int(timesteps)
        self.sampling_timesteps = int(sampling_timesteps) if sampling_timesteps is not None else int(timesteps)
        assert self.sampling_timesteps <= self.num_timesteps
        self.is_ddim_sampling = self.sampling_timesteps < self.num_timesteps
        self.ddim_sampling_eta = float(ddim_sampling_eta)
        for k, v in make_schedule(self.num_timesteps).items():
            setattr(self, k, v)
        self._samplers = {}
        dep = getattr(model, "_dependents", None)
        if dep is not None:
            dep.add(self)               # reloading / closing the network drops the samplers built on its old handle

    # -- transition table -------------------------------------------------------------------------
    def step_table(self) -> List[dict]:
        """One dict per transition, in execution order; field meaning documented at prg_step (include/prg.h)."""
        rows = []
        if not self.is_ddim_sampling:                                   # p_sample_loop (sd:1283-1317)
            for t in reversed(range(self.num_timesteps)):
                sig = (0.5 * self.posterior_log_variance_clipped[t]).exp() if t > 0 else torch.tensor(0.0)
                rows.append(dict(t=t, clip_pred=2, c_x0=self.posterior_mean_coef1[t], c_x=self.posterior_mean_coef2[t],
                                 c_eps=0.0, sigma=sig, sqrt_recip=self.sqrt_recip_alphas_cumprod[t],
                                 sqrt_recipm1=self.sqrt_recipm1_alphas_cumprod[t]))
        else:                                                           # ddim_sample (sd:1319-1392)
            times = ddim_times(self.num_timesteps, self.sampling_timesteps)
            for t, tn in zip(times[:-1], times[1:]):
                row = dict(t=t, clip_pred=1, sqrt_recip=self.sqrt_recip_alphas_cumprod[t],
                           sqrt_recipm1=self.sqrt_recipm1_alphas_cumprod[t])
                if tn < 0:
                    row.update(c_x0=1.0, c_x=0.0, c_eps=0.0, sigma=0.0)
                else:
                    a, an = self.alphas_cumprod[t], self.alphas_cumprod[tn]
                    sigma = self.ddim_sampling_eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                    c = (1 - an - sigma ** 2).sqrt()
                    row.update(c_x0=an.sqrt(), c_x=0.0, c_eps=c, sigma=sigma)
                rows.append(row)
        return [{k: (float(v) if k not in ("t", "clip_pred") else int(v)) for k, v in r.items()} for r in rows]

    def _steps_c(self, refine: bool = False):
        rows = self.step_table()
        if refine:
            # has_refine_step (sd:1307-1314 / sd:1374-1388): one more evaluation at t = 0 WITHOUT the DDNM replacement;
            # the known pixels take its clamped output, the in-painted ones keep their value
            rows = rows + [dict(t=0, clip_pred=4, c_x0=0.0, c_x=0.0, c_eps=0.0, sigma=0.0,
                                sqrt_recip=float(self.sqrt_recip_alphas_cumprod[0]),
                                sqrt_recipm1=float(self.sqrt_recipm1_alphas_cumprod[0]))]
        arr = (_lib.StepC * len(rows))()
        for i, r in enumerate(rows):
            arr[i] = _lib.StepC(r["t"], r["clip_pred"], r["c_x0"], r["c_x"], r["c_eps"], r["sigma"], r["sqrt_recip"],
                                r["sqrt_recipm1"])
        return arr, len(rows)

    @property
    def n_draws(self) -> int:
        """Noise slabs a stored-noise run consumes: the start image + one per transition that adds noise
        (every transition but the last, for both samplers) = the reference's number of randn draws."""
        rows = self.step_table()
        return 1 + max([k + 1 for k, r in enumerate(rows) if r["sigma"] != 0.0], default=0)

    def _sampler(self, batch: int, refine: bool = False):
        key = (batch, self.image_size, bool(refine))
        if key not in self._samplers:
            lib = _lib.load()
            arr, n = self._steps_c(refine)
            h = C.c_void_p()
            _lib.check(lib.prg_sampler_create(self.model.handle, arr, n, batch, self.image_size, C.byref(h)),
                       "prg_sampler_create")
            self._samplers[key] = h
        return self._samplers[key]

    def close(self):
        lib = _lib.load()
        for h in self._samplers.values():
            lib.prg_sampler_destroy(h)
        self._samplers.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- sampling ---------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, *, param_cond: torch.Tensor, img_cond: Optional[torch.Tensor] = None, disable_tqdm=True,
               has_refine_step=False, noise: Optional[torch.Tensor] = None, seeds: Optional[Sequence[int]] = None,
               use_graph: bool = True, profile: bool = False) -> torch.Tensor:
        """(B,4) intrinsics vector, (B,2,S,S) condition in [-1,1] -> (B,1,S,S) depth in [0,1]  (sd:1394-1409).

        ``noise``: (n_draws,B,1,S,S) stored draws in the reference's order (parity runs); otherwise on-device
        Philox keyed by ``seeds`` (one 64-bit key per scene, default 0..B-1)."""
        lib = _lib.load()
        pc = param_cond.to(device="cuda", dtype=torch.float32).contiguous()
        B, S = pc.shape[0], self.image_size
        cond = None
        if img_cond is not None and self.is_ddnm_sampling:
            cond = img_cond.to(device="cuda", dtype=torch.float32).contiguous()
            assert tuple(cond.shape) == (B, 2, S, S)
        h = self._sampler(B, bool(has_refine_step) and cond is not None)
        _lib.check(lib.prg_sampler_set_graph(h, int(use_graph)))
        _lib.check(lib.prg_sampler_set_profile(h, int(profile)))
        nz = None
        seed_arr = None
        if noise is not None:
            nz = noise.to(device="cuda", dtype=torch.float32).contiguous()
            assert nz.numel() % (B * S * S) == 0 and nz.numel() // (B * S * S) >= self.n_draws, \
                f"stored noise needs {self.n_draws} draws of shape ({B},1,{S},{S})"
        else:
            seed_arr = (C.c_uint64 * B)(*[int(s) & 0xFFFFFFFFFFFFFFFF for s in (seeds if seeds is not None else range(B))])
        out = torch.empty((B, 1, S, S), dtype=torch.float32, device="cuda")
        _lib.check(lib.prg_sampler_run(h, _lib.ptr(pc), _lib.ptr(cond), _lib.ptr(nz),
                                       0 if nz is None else nz.numel() // (B * S * S),
                                       C.cast(seed_arr, C.c_void_p) if seed_arr is not None else None,
                                       _lib.ptr(out), _lib.stream_ptr()), "prg_sampler_run")
        self._keepalive = (pc, cond, nz)   # the run is asynchronous: keep inputs alive until the next call
        return out

    def last_profile(self, batch: int, refine: bool = False) -> dict:
        lib = _lib.load()
        h = self._sampler(batch, refine)
        ms, n, fl, tot = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        _lib.check(lib.prg_sampler_get_profile(h, C.byref(ms), C.byref(n), C.byref(fl), C.byref(tot)))
        by = C.c_double()
        _lib.check(lib.prg_sampler_get_profile_bytes(h, C.byref(by)))
        sms, sn = C.c_double(), C.c_int64()
        _lib.check(lib.prg_sampler_get_profile_step(h, C.byref(sms), C.byref(sn)))
        ex = C.c_double()
        _lib.check(lib.prg_sampler_get_profile_executed(h, C.byref(ex)))
        nrows = C.c_int32()
        _lib.check(lib.prg_sampler_get_profile_shapes(h, None, 0, C.byref(nrows)))
        rows = (_ProfileShape * max(1, nrows.value))()
        _lib.check(lib.prg_sampler_get_profile_shapes(h, C.cast(rows, C.c_void_p), nrows.value, C.byref(nrows)))
        shapes = [{f: getattr(rows[i], f) for f, _ in _ProfileShape._fields_} for i in range(nrows.value)]
        return {"conv_ms": ms.value, "conv_launches": n.value, "conv_flops": fl.value, "conv_bytes": by.value, "conv_flops_executed": ex.value,
                "total_ms": tot.value, "step_ms": sms.value, "step_launches": sn.value, "conv_shapes": shapes}
