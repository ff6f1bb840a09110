"""Oracle: noise schedule, DDNM model predictions, ancestral and DDIM samplers, and the per-pair
pipeline around them (TEST INFRASTRUCTURE — see oracle/__init__.py).

sd = /root/reference/denoising_diffusion_pytorch/successive_ddnm_diffusion.py

Noise handling: the reference draws from torch's global generator (randn(shape) for the start image,
randn_like per step, none on the last step).  Every sampler here takes a ``noise_fn(step_index) ->
tensor`` so a test can either reproduce that stream (``torch_stream_noise``) or feed stored tensors that
the HIP path consumes too.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import geometry as geo
from . import unet as U


# ----------------------------------------------------------------------------------------------
# schedule (sd:997-1012, sd:1047-1134)
# ----------------------------------------------------------------------------------------------

def sigmoid_betas(T: int, start=-3.0, end=3.0, tau=1.0) -> torch.Tensor:
    t = torch.linspace(0, T, T + 1, dtype=torch.float64) / T
    v0 = torch.tensor(start / tau).sigmoid()
    v1 = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v1) / (v1 - v0)
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


def schedule(T: int = 1000) -> Dict[str, torch.Tensor]:
    """All buffers the sampler reads, computed in float64 and stored as float32 like the reference."""
    b = sigmoid_betas(T)
    a = 1.0 - b
    ac = torch.cumprod(a, 0)
    acp = torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]])
    pv = b * (1.0 - acp) / (1.0 - ac)
    s64 = {
        "betas": b,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": acp,
        "sqrt_alphas_cumprod": ac.sqrt(),
        "sqrt_one_minus_alphas_cumprod": (1.0 - ac).sqrt(),
        "log_one_minus_alphas_cumprod": (1.0 - ac).log(),
        "sqrt_recip_alphas_cumprod": (1.0 / ac).sqrt(),
        "sqrt_recipm1_alphas_cumprod": (1.0 / ac - 1).sqrt(),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": pv.clamp(min=1e-20).log(),
        "posterior_mean_coef1": b * acp.sqrt() / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - acp) * a.sqrt() / (1.0 - ac),
        "loss_weight": ac / (1 - ac),          # objective 'pred_x0', no min-SNR clipping (sd:1138-1151)
    }
    return {k: v.to(torch.float32) for k, v in s64.items()}


def ddim_time_pairs(T: int, steps: int) -> List[tuple]:
    """linspace(-1, T-1, steps+1) truncated to int, reversed, paired (sd:1331-1337)."""
    times = list(reversed(torch.linspace(-1, T - 1, steps=steps + 1).int().tolist()))
    return list(zip(times[:-1], times[1:]))


# ----------------------------------------------------------------------------------------------
# one model evaluation + DDNM replacement (sd:1182-1232, objective 'pred_x0')
# ----------------------------------------------------------------------------------------------

Denoiser = Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor]


def cond_mask(img_cond: torch.Tensor) -> torch.Tensor:
    """Second condition channel back to a boolean (sd:507-508)."""
    return (img_cond[:, 1:2] + 1) * 0.5 > 0.5


def model_predictions(sch, denoise: Denoiser, x, t: int, param_cond, img_cond, clip_x_start: bool):
    """Returns (pred_noise, x_start).  pred_noise is derived from the (optionally clamped) network
    output BEFORE the known pixels are written into x_start — the two are deliberately inconsistent in
    the reference and that is reproduced."""
    tt = torch.full((x.shape[0],), t, dtype=torch.long)
    x0 = denoise(x, tt, param_cond)
    if clip_x_start:
        x0 = x0.clamp(-1.0, 1.0)
    eps = (sch["sqrt_recip_alphas_cumprod"][t] * x - x0) / sch["sqrt_recipm1_alphas_cumprod"][t]
    if img_cond is not None:
        x0 = torch.where(cond_mask(img_cond), img_cond[:, 0:1], x0)
    return eps, x0


def p_sample(sch, denoise, x, t: int, param_cond, img_cond, noise: Optional[torch.Tensor]):
    """One ancestral step (sd:1234-1281): clamp x_start, posterior mean, add exp(0.5 logvar) * noise for t>0."""
    _, x0 = model_predictions(sch, denoise, x, t, param_cond, img_cond, clip_x_start=False)
    x0 = x0.clamp(-1.0, 1.0)
    mean = sch["posterior_mean_coef1"][t] * x0 + sch["posterior_mean_coef2"][t] * x
    if t > 0:
        return mean + (0.5 * sch["posterior_log_variance_clipped"][t]).exp() * noise, x0
    return mean + 0.0, x0


def refine(sch, denoise, img, param_cond, img_cond):
    """has_refine_step (sd:1307-1314 ancestral, sd:1374-1388 DDIM; same arithmetic): one more evaluation at t = 0 with
    the DDNM replacement banned; the KNOWN pixels take its clamped output (posterior mean at t = 0 is x0 itself:
    coef1[0] = 1, coef2[0] = 0), the in-painted ones keep their value."""
    tt = torch.zeros((img.shape[0],), dtype=torch.long)
    x0 = denoise(img, tt, param_cond).clamp(-1.0, 1.0)
    mean = sch["posterior_mean_coef1"][0] * x0 + sch["posterior_mean_coef2"][0] * img
    return torch.where(cond_mask(img_cond), mean, img)


def p_sample_loop(sch, denoise, param_cond, img_cond, shape, noise_fn, T: Optional[int] = None,
                  has_refine_step: bool = False, stop_after: Optional[int] = None):
    """T-step DDNM ancestral chain (sd:1283-1317).  noise_fn(0) is the start image; noise_fn(k) for k = 1..T-1 feeds
    the step at t = T-k; t = 0 draws nothing.  ``stop_after=k`` returns the raw state after k transitions (the input of
    the network's call k; a test hook for checking a prefix of a long chain)."""
    T = T or sch["betas"].shape[0]
    img = noise_fn(0)
    assert tuple(img.shape) == tuple(shape)
    for k, t in enumerate(range(T - 1, -1, -1)):
        if stop_after is not None and k == stop_after:
            return img
        img, _ = p_sample(sch, denoise, img, t, param_cond, img_cond, noise_fn(k + 1) if t > 0 else None)
    if has_refine_step:
        img = refine(sch, denoise, img, param_cond, img_cond)
    return (img + 1) * 0.5


def ddim_sample(sch, denoise, param_cond, img_cond, shape, noise_fn, steps: int, eta: float = 1.0,
                T: Optional[int] = None, has_refine_step: bool = False, stop_after: Optional[int] = None):
    """DDIM with eta (sd:1319-1392): clamped x_start, no draw on the last pair.  ``stop_after``: see p_sample_loop."""
    T = T or sch["betas"].shape[0]
    ac = sch["alphas_cumprod"]
    img = noise_fn(0)
    assert tuple(img.shape) == tuple(shape)
    for k, (t, t_next) in enumerate(ddim_time_pairs(T, steps)):
        if stop_after is not None and k == stop_after:
            return img
        eps, x0 = model_predictions(sch, denoise, img, t, param_cond, img_cond, clip_x_start=True)
        if t_next < 0:
            img = x0
            continue
        a, an = ac[t], ac[t_next]
        sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
        c = (1 - an - sigma ** 2).sqrt()
        img = x0 * an.sqrt() + c * eps + sigma * noise_fn(k + 1)
    if has_refine_step:
        img = refine(sch, denoise, img, param_cond, img_cond)
    return (img + 1) * 0.5


def sample(sch, denoise, param_cond, img_cond, image_size: int, noise_fn, sampling_steps: Optional[int] = None,
           eta: float = 1.0, has_refine_step: bool = False, stop_after: Optional[int] = None):
    """Dispatch exactly like sd:1394-1409: ancestral iff sampling_steps == T."""
    T = sch["betas"].shape[0]
    steps = sampling_steps or T
    shape = (param_cond.shape[0], 1, image_size, image_size)
    if steps < T:
        return ddim_sample(sch, denoise, param_cond, img_cond, shape, noise_fn, steps, eta, has_refine_step=has_refine_step,
                           stop_after=stop_after)
    return p_sample_loop(sch, denoise, param_cond, img_cond, shape, noise_fn, has_refine_step=has_refine_step,
                         stop_after=stop_after)


def torch_stream_noise(shape, generator: Optional[torch.Generator] = None):
    """noise_fn reproducing the reference's draw order from a torch CPU generator."""
    def fn(_k):
        return torch.randn(shape, generator=generator)
    return fn


def stored_noise(noise: torch.Tensor):
    """noise_fn over a stacked (n_draws, B, 1, S, S) tensor (what the HIP sampler is fed in parity mode)."""
    def fn(k):
        return noise[k]
    return fn


# ----------------------------------------------------------------------------------------------
# the per-pair pipeline around the sampler (sd:2525-2628, num_samples = 1, synthetic depth input)
# ----------------------------------------------------------------------------------------------

MASK_THRESHOLD = 0.99   # sd:2565, sd:2580


def correct_and_condition(mask_prob: torch.Tensor, depth_rpj: torch.Tensor, hit: torch.Tensor):
    """Threshold the keep-probability, zero rejected depth, AND the masks, build the [-1,1] condition
    (sd:2564-2570).  Returns (img_cond (B,2,S,S), corrected depth, combined mask)."""
    keep = mask_prob > MASK_THRESHOLD
    d = torch.where(keep, depth_rpj, torch.zeros_like(depth_rpj))
    m = hit & keep
    cond = torch.cat([d, m.to(d.dtype)], dim=1) * 2 - 1
    return cond, d, m


def generate_pairs(sch, unet_p, mask_p, depth0: torch.Tensor, K: np.ndarray, pose: np.ndarray, noise_fn,
                   sampling_steps: Optional[int] = None, clip=(0.5, 10.0)) -> dict:
    """One batch of synthetic scene pairs, depth-map form of the pipeline:

      source depth (B,1,S,S) [1.0 == 10 m] --reproject by pose--> z-buffer depth + hit mask (sd:268-286)
      -> x0.1 -> MaskUnet > 0.99 -> condition -> sampler -> MaskUnet > 0.99 -> zero rejected
      -> numpy unprojection (clip [0.5,10] m, float64) -> inverse pose  (sd:2552-2628)

    Returns the intermediate maps and, per scene, the (n,3) float64 cloud in the common frame whose
    L-infinity difference is the parity metric."""
    B, _, S, _ = depth0.shape
    Kt, Pt = torch.tensor(K), torch.tensor(pose)
    d_rpj, hit = geo.reproject_tensor(depth0 * geo.DEPTH_UNIT_M, Kt, Pt)
    d_rpj = d_rpj * 0.1
    prob1 = U.maskunet_forward(mask_p, d_rpj)
    cond, d_crt, m = correct_and_condition(prob1, d_rpj, hit)
    pc = geo.param_vector(Kt)
    den = lambda x, t, c: U.unet_forward(unet_p, x, t, c)
    img = sample(sch, den, pc, cond, S, noise_fn, sampling_steps)
    prob2 = U.maskunet_forward(mask_p, img)
    out = torch.where(prob2 > MASK_THRESHOLD, img, torch.zeros_like(img))
    clouds = []
    for b in range(B):
        p = geo.point_cloud(out[b, 0].numpy() * 10, K[b], clip)
        clouds.append(geo.inverse_pose_apply(p, pose[b]))
    return {"depth_rpj": d_rpj, "hit": hit, "prob1": prob1, "img_cond": cond, "sampled": img,
            "prob2": prob2, "depth_out": out, "clouds": clouds}
