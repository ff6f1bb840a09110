"""Camera geometry of the generative data path: host-side mirror of the reference helpers.

Same names, argument meaning and error behaviour as the reference functions (cited per function,
sd = /root/reference/denoising_diffusion_pytorch/successive_ddnm_diffusion.py); the tensor ops run as HIP
kernels through the C-ABI (include/prg.h).  Tiny per-scene scalars (intrinsics, poses) stay on the host in
numpy exactly like the reference.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from . import _lib

DEPTH_UNIT_M = 10.0   # normalised depth 1.0 == 10 m (sd:2458, sd:2552)
K_CANDIDATES_F = (585.0, 572.0, 583.0, 540.021232, 570.342205, 533.069214)   # sd:358-366
K_WEIGHTS = (7, 8, 18, 5, 47, 5)                                             # sd:367


# ------------------------------------------------------------------------------------------------
# host helpers (numpy)
# ------------------------------------------------------------------------------------------------

def candidate_intrinsics() -> np.ndarray:
    out = np.zeros((len(K_CANDIDATES_F), 3, 3), dtype=np.float32)
    out[:, 0, 0] = out[:, 1, 1] = np.asarray(K_CANDIDATES_F, dtype=np.float32)
    out[:, 0, 2], out[:, 1, 2], out[:, 2, 2] = 320.0, 240.0, 1.0
    return out


def random_sample_intrinsic(batch_size: int) -> np.ndarray:
    """One of the six 3DMatch intrinsics per item, numpy legacy RNG (sd:354-374)."""
    p = np.asarray(K_WEIGHTS, dtype=np.float64)
    idx = np.random.choice(len(K_CANDIDATES_F), batch_size, replace=True, p=p / p.sum())
    return candidate_intrinsics()[idx]


def intrinsic_transform(intrinsic: np.ndarray, resize: Optional[int] = None,
                        centercrop: Optional[int] = None) -> np.ndarray:
    """Intrinsics after Resize(int) + CenterCrop(int) (sd:47-119; int arguments, the form sd:2436-2441 uses)."""
    K = np.asarray(intrinsic)
    fx, fy, cx, cy = K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]
    size_x, size_y = np.int32(cx * 2), np.int32(cy * 2)
    new_x, new_y = size_x, size_y
    nfx, nfy, ncx, ncy = fx, fy, cx, cx   # (sic) the reference seeds new_cy with old_cx when resize is None (sd:67)
    if resize is not None:
        if not isinstance(resize, (int, np.integer)):
            raise TypeError("only integer resize is supported on this path")
        if (size_x < size_y).all():
            new_x, new_y = int(resize), np.int32(np.floor(resize * size_y / size_x))
        else:
            new_x, new_y = np.int32(np.floor(resize * size_x / size_y)), np.int32(resize)
        nfx, nfy = np.float32(fx * new_x / size_x), np.float32(fy * new_y / size_y)
        ncx, ncy = np.float32(new_x / 2), np.float32(new_y / 2)
    if centercrop is not None:
        if not isinstance(centercrop, (int, np.integer)):
            raise TypeError("only integer centercrop is supported on this path")
        ncx = ncx - np.int32(np.round((new_x - centercrop) / 2.0))
        ncy = ncy - np.int32(np.round((new_y - centercrop) / 2.0))
    out = np.zeros_like(K)
    out[..., 0, 0], out[..., 1, 1], out[..., 0, 2], out[..., 1, 2], out[..., 2, 2] = nfx, nfy, ncx, ncy, 1.0
    return out


def param_vector(intrinsic: torch.Tensor) -> torch.Tensor:
    """(…,3,3) -> (…,4) [fx, fy, cx, cy]  (sd:343-351)."""
    return torch.stack([intrinsic[..., 0, 0], intrinsic[..., 1, 1], intrinsic[..., 0, 2], intrinsic[..., 1, 2]], -1)


def random_sample_pose(batch_size: int, center=(0, 0, 3), rng=None) -> np.ndarray:
    """Random camera motion about a pivot 3 m ahead, numpy legacy RNG, same draw order (sd:417-443).  `rng`: a
    `np.random.RandomState` to draw from instead of numpy's process-wide legacy stream (same algorithm, same values)."""
    np_random = np.random if rng is None else rng
    theta = np_random.rand(batch_size) * (np.pi / 12) - np.pi / 24
    phi = np_random.rand(batch_size) * (np.pi / 6) - np.pi / 12
    rot = Rotation.from_euler("XYZ", np.stack((theta, phi, np.zeros(batch_size)), axis=-1)).as_matrix()
    c = np.array(center)
    jitter = np_random.randn(batch_size, 3) / 3
    jitter[:, -1] = 0
    T = np.stack([np.eye(4) for _ in range(batch_size)])
    T[:, :3, :3] = rot
    T[:, :3, 3] = c - rot @ c + jitter
    return T.astype(np.float32)


# ------------------------------------------------------------------------------------------------
# device ops (HIP)
# ------------------------------------------------------------------------------------------------

def random_sample_transform(intrinsic: np.ndarray, image_size: int = 256) -> np.ndarray:
    """Random in-place camera rotation of Tester.generate (sd:377-415): pitch / yaw within the view frustum, any roll;
    translation drawn and multiplied by zero (the randn still consumes the legacy numpy stream).  float32 (B,4,4)."""
    K = np.asarray(intrinsic)
    batch = K.shape[0]
    h = w = image_size
    fx, fy, cx, cy = K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]
    th_min, th_max = -np.arctan((h - cy) / fy), np.arctan(cy / fy)
    ph_min, ph_max = -np.arctan(cx / fx), np.arctan((w - cx) / fx)
    theta = np.random.rand(batch) * (th_max - th_min) + th_min
    phi = np.random.rand(batch) * (ph_max - ph_min) + ph_min
    psi = np.random.rand(batch) * 2 * np.pi - np.pi
    R = Rotation.from_euler("XYZ", np.stack((theta, phi, psi), axis=-1), degrees=False).as_matrix()
    t = np.random.randn(batch, 3) / 3 * 0
    T = np.stack([np.eye(4) for _ in range(batch)])
    T[..., :3, :3] = R
    T[..., :3, 3] = t
    return T.astype(np.float32)


def _f32(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.PrgError("expected a tensor on the HIP device (this package has no CPU path)")
    return t.contiguous().to(torch.float32)


def depth2pc_tensor(depth: torch.Tensor, intrinsic: torch.Tensor, *, clip=(0, 10), invalid_num=None):
    """(B,1,H,W) depth, (B,3,3) K -> pc (B,HW,3) (invalid -> NaN / invalid_num), valid (B,HW) bool  (sd:176-209)."""
    lib = _lib.load()
    depth, K = _f32(depth), _f32(intrinsic)
    B, Cc, H, W = depth.shape
    assert Cc == 1
    pc = torch.empty((B, H * W, 3), dtype=torch.float32, device=depth.device)
    valid = torch.empty((B, H * W), dtype=torch.uint8, device=depth.device)
    lo, hi = (1.0, 0.0) if clip is None else (float(clip[0]), float(clip[1]))
    inv = float("nan") if invalid_num is None else float(invalid_num)
    _lib.check(lib.prg_depth2pc(_lib.ptr(depth), _lib.ptr(K), _lib.ptr(pc), _lib.ptr(valid), B, H, W, lo, hi, inv,
                                _lib.stream_ptr()), "prg_depth2pc")
    return pc, valid.view(torch.bool)


def pc2depth_tensor(pc: torch.Tensor, valid: Optional[torch.Tensor], intrinsic: torch.Tensor, *,
                    image_size=(480, 640)):
    """Z-buffer: (B,N,3), (B,N) bool, (B,3,3) -> depth (B,1,H,W) f32 (nearest z, 0 = empty), mask bool  (sd:212-265)."""
    lib = _lib.load()
    pc, K = _f32(pc), _f32(intrinsic)
    B, N, _ = pc.shape
    H, W = int(image_size[0]), int(image_size[1])
    v8 = None if valid is None else valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.contiguous().to(torch.uint8)
    depth = torch.empty((B, 1, H, W), dtype=torch.float32, device=pc.device)
    mask = torch.empty((B, 1, H, W), dtype=torch.uint8, device=pc.device)
    _lib.check(lib.prg_pc2depth(_lib.ptr(pc), _lib.ptr(v8), _lib.ptr(K), _lib.ptr(depth), _lib.ptr(mask), B, N, H, W,
                                _lib.stream_ptr()), "prg_pc2depth")
    return depth, mask.view(torch.bool)


def reproject_tensor(depth: torch.Tensor, intrinsic: torch.Tensor, relative_pose: torch.Tensor, *, clip=(0, 10),
                     invalid_num=None, depth_unit: float = 1.0, out_scale: float = 1.0):
    """Unproject, move by the SE(3) pose, z-buffer into the same camera — one fused kernel (sd:268-286).

    ``depth_unit`` / ``out_scale`` fold the caller's `depth * 10` and `images_rpj * 0.1` (sd:484, sd:2552)."""
    lib = _lib.load()
    depth, K, P = _f32(depth), _f32(intrinsic), _f32(relative_pose)
    B, Cc, H, W = depth.shape
    assert Cc == 1 and clip is not None
    out = torch.empty_like(depth)
    mask = torch.empty((B, 1, H, W), dtype=torch.uint8, device=depth.device)
    _lib.check(lib.prg_reproject_zbuffer(_lib.ptr(depth), _lib.ptr(K), _lib.ptr(P), _lib.ptr(out), _lib.ptr(mask), B,
                                         H, W, float(depth_unit), float(clip[0]), float(clip[1]), float(out_scale),
                                         _lib.stream_ptr()), "prg_reproject_zbuffer")
    return out, mask.view(torch.bool)


def project_clouds(clouds: Sequence[np.ndarray], poses: np.ndarray, intrinsic: np.ndarray, image_size: int,
                   device, depth_scale: float = 1.0):
    """Generator.generate's per-scene projection (sd:2531-2552) for a whole batch in one launch: ragged float32
    clouds (n_b,3) each moved by its (4,4) pose and z-buffered with its K; returns depth*depth_scale and mask."""
    lib = _lib.load()
    B = len(clouds)
    offs = np.zeros(B + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(c) for c in clouds])
    pts = torch.from_numpy(np.concatenate([np.asarray(c, dtype=np.float32).reshape(-1, 3) for c in clouds], 0)
                           if offs[-1] else np.zeros((1, 3), np.float32)).to(device)
    d_offs = torch.from_numpy(offs).to(device)
    P = torch.from_numpy(np.ascontiguousarray(poses, dtype=np.float32)).to(device)
    K = torch.from_numpy(np.ascontiguousarray(intrinsic, dtype=np.float32)).to(device)
    S = int(image_size)
    depth = torch.empty((B, 1, S, S), dtype=torch.float32, device=device)
    mask = torch.empty((B, 1, S, S), dtype=torch.uint8, device=device)
    _lib.check(lib.prg_project_points_zbuffer(_lib.ptr(pts), _lib.ptr(d_offs), _lib.ptr(P), _lib.ptr(K), _lib.ptr(depth),
                                              _lib.ptr(mask), B, S, S, float(depth_scale), _lib.stream_ptr()),
               "prg_project_points_zbuffer")
    return depth, mask.view(torch.bool)


def unproject_f64(depth: torch.Tensor, intrinsic: torch.Tensor, pose: Optional[torch.Tensor], *,
                  depth_unit: float = DEPTH_UNIT_M, clip=(0.5, 10.0)):
    """Batched float64 `point_cloud` + inverse pose (sd:122-143, sd:2623-2628) -> xyz (B,HW,3) f64, valid (B,HW)."""
    lib = _lib.load()
    depth, K = _f32(depth), _f32(intrinsic)
    P = None if pose is None else _f32(pose)
    B, Cc, H, W = depth.shape
    xyz = torch.empty((B, H * W, 3), dtype=torch.float64, device=depth.device)
    valid = torch.empty((B, H * W), dtype=torch.uint8, device=depth.device)
    _lib.check(lib.prg_unproject_f64(_lib.ptr(depth), _lib.ptr(K), _lib.ptr(P), _lib.ptr(xyz), _lib.ptr(valid), B, H, W,
                                     float(depth_unit), float(clip[0]), float(clip[1]), _lib.stream_ptr()),
               "prg_unproject_f64")
    return xyz, valid.view(torch.bool)


def point_clouds(depth: torch.Tensor, intrinsic: torch.Tensor, pose: Optional[torch.Tensor] = None, *,
                 depth_unit: float = DEPTH_UNIT_M, clip=(0.5, 10.0)) -> List[np.ndarray]:
    """Per image: the compacted (n_valid,3) float64 cloud in row-major pixel order — what `point_cloud(img*10, K,
    clip)` followed by `(pc - t) @ R` returns in the reference (sd:2623-2628)."""
    xyz, valid = unproject_f64(depth, intrinsic, pose, depth_unit=depth_unit, clip=clip)
    xyz, valid = xyz.cpu().numpy(), valid.cpu().numpy()
    return [xyz[b][valid[b]] for b in range(xyz.shape[0])]


def depth_augment(depth: torch.Tensor) -> torch.Tensor:
    """DepthAugment (dc:577-604): (B,1,H,W) -> (B,3,H,W)."""
    lib = _lib.load()
    depth = _f32(depth)
    B, _, H, W = depth.shape
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=depth.device)
    _lib.check(lib.prg_depth_augment(_lib.ptr(depth), _lib.ptr(out), B, H, W, _lib.stream_ptr()), "prg_depth_augment")
    return out


def apply_mask(prob: torch.Tensor, depth: torch.Tensor, hit: Optional[torch.Tensor], threshold: float = 0.99,
               want_cond: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """keep = prob > thr; depth[~keep] = 0; hit &= keep; img_cond = cat[depth, hit]*2-1  (sd:2564-2570)."""
    lib = _lib.load()
    prob, depth = _f32(prob), _f32(depth)
    B, _, H, W = depth.shape
    h8 = None if hit is None else (hit.contiguous().view(torch.uint8) if hit.dtype == torch.bool else hit.contiguous().to(torch.uint8))
    d_out = torch.empty_like(depth)
    h_out = torch.empty((B, 1, H, W), dtype=torch.uint8, device=depth.device)
    cond = torch.empty((B, 2, H, W), dtype=torch.float32, device=depth.device) if want_cond else None
    _lib.check(lib.prg_apply_mask(_lib.ptr(prob), _lib.ptr(depth), _lib.ptr(h8), float(threshold), _lib.ptr(d_out),
                                  _lib.ptr(h_out), _lib.ptr(cond), B, H, W, _lib.stream_ptr()), "prg_apply_mask")
    return d_out, h_out.view(torch.bool), cond


def occlusion_filter(depth_rpj: torch.Tensor, mask_rpj: torch.Tensor, threshold: float = 0.0375):
    """(sd:446-463) a reprojected pixel more than `threshold` metres behind the nearest valid depth of its 3x3 window
    takes that depth (thin foreground structures win over what shows through them).  Returns (depth, mask unchanged)."""
    lib = _lib.load()
    depth = _f32(depth_rpj)
    B, _, H, W = depth.shape
    m8 = mask_rpj.contiguous().view(torch.uint8) if mask_rpj.dtype == torch.bool else mask_rpj.contiguous().to(torch.uint8)
    out = torch.empty_like(depth)
    _lib.check(lib.prg_occlusion_filter(_lib.ptr(depth), _lib.ptr(m8), _lib.ptr(out), B, H, W, float(threshold),
                                        _lib.stream_ptr()), "prg_occlusion_filter")
    return out, mask_rpj
