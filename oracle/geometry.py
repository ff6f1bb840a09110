"""Oracle: camera geometry of the generative data path (TEST INFRASTRUCTURE — see oracle/__init__.py).

sd = /root/reference/denoising_diffusion_pytorch/successive_ddnm_diffusion.py
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial.transform import Rotation

DEPTH_UNIT_M = 10.0   # normalised depth 1.0 == 10 m (sd:2458, sd:2552)

# the six 3DMatch pinhole intrinsics and their sampling weights (sd:358-368)
K_CANDIDATES_F = (585.0, 572.0, 583.0, 540.021232, 570.342205, 533.069214)
K_WEIGHTS = (7, 8, 18, 5, 47, 5)


def candidate_intrinsics() -> np.ndarray:
    k = np.zeros((len(K_CANDIDATES_F), 3, 3), dtype=np.float32)
    for i, f in enumerate(K_CANDIDATES_F):
        k[i] = [[f, 0, 320.0], [0, f, 240.0], [0, 0, 1.0]]
    return k


def intrinsic_transform(K: np.ndarray, resize: int, centercrop: int) -> np.ndarray:
    """Intrinsics after torchvision Resize(int) (short side -> resize, long side floor) and CenterCrop(int).

    Follows sd:47-119 for the int/int case the generator uses (sd:2436-2441): image size is inferred as
    (2cx, 2cy) truncated to int32; focal lengths scale by new/old size; principal point moves to the
    new centre then shifts by the crop offset, rounded half-to-even (np.round).
    """
    K = np.asarray(K)
    fx, fy, cx, cy = K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]
    w0 = np.int32(cx * 2)
    h0 = np.int32(cy * 2)
    if (w0 < h0).all():
        w1 = int(resize)
        h1 = np.int32(np.floor(resize * h0 / w0))
    else:
        w1 = np.int32(np.floor(resize * w0 / h0))
        h1 = np.int32(resize)
    nfx = np.float32(fx * w1 / w0)
    nfy = np.float32(fy * h1 / h0)
    ncx = np.float32(w1 / 2)
    ncy = np.float32(h1 / 2)
    left = np.int32(np.round((w1 - centercrop) / 2.0))
    top = np.int32(np.round((h1 - centercrop) / 2.0))
    out = np.zeros_like(K)
    out[..., 0, 0] = nfx
    out[..., 1, 1] = nfy
    out[..., 0, 2] = ncx - left
    out[..., 1, 2] = ncy - top
    out[..., 2, 2] = 1.0
    return out


def param_vector(K: torch.Tensor) -> torch.Tensor:
    """(…,3,3) -> (…,4) = [fx, fy, cx, cy]  (sd:343-351)."""
    return torch.stack([K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]], dim=-1)


def random_sample_pose(batch: int, center=(0.0, 0.0, 3.0)) -> np.ndarray:
    """Random camera motion about a pivot 3 m ahead (sd:417-443).

    Draw order on numpy's legacy global RNG (pins the stream): rand(B) -> theta about x in +-pi/24,
    rand(B) -> phi about y in +-pi/12, randn(B,3)/3 -> translation jitter with z zeroed.  Rotation =
    intrinsic 'XYZ' Euler; t = c - R c + jitter.  Returned as float32 (B,4,4).
    """
    th = np.random.rand(batch) * (np.pi / 12) - np.pi / 24
    ph = np.random.rand(batch) * (np.pi / 6) - np.pi / 12
    R = Rotation.from_euler("XYZ", np.stack([th, ph, np.zeros(batch)], -1)).as_matrix()
    c = np.asarray(center, dtype=np.float64)
    jitter = np.random.randn(batch, 3) / 3
    jitter[:, 2] = 0
    T = np.tile(np.eye(4), (batch, 1, 1))
    T[:, :3, :3] = R
    T[:, :3, 3] = c - R @ c + jitter
    return T.astype(np.float32)


def point_cloud(depth: np.ndarray, K: np.ndarray, clip=(0.0, 10.0)) -> np.ndarray:
    """Pinhole unprojection of an (H,W) metric depth map to the (n_valid,3) cloud, row-major (sd:122-143).

    numpy promotion makes this float64: the int64 pixel grid minus the float32 principal point is
    float64, and everything downstream inherits it.
    """
    fx, fy, cx, cy = K[0][0], K[1][1], K[0][2], K[1][2]
    H, W = depth.shape
    r = np.arange(H)[:, None]
    c = np.arange(W)[None, :]
    ok = (depth > clip[0]) & (depth < clip[1])
    z = np.where(ok, depth, np.nan)
    x = (c - cx) * z / fx
    y = (r - cy) * z / fy
    pts = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    return pts[ok.reshape(-1)]


def inverse_pose_apply(pc: np.ndarray, pose: np.ndarray) -> np.ndarray:
    """(p - t) R  ==  R^T (p - t): camera frame -> common frame (sd:2627-2628)."""
    return (pc - pose[:3, 3]) @ pose[:3, :3]


def depth2pc_tensor(depth: torch.Tensor, K: torch.Tensor, clip=(0.0, 10.0)):
    """(B,1,H,W) -> (B,HW,3) points (NaN where invalid) and (B,HW) validity (sd:176-209)."""
    B, _, H, W = depth.shape
    fx, fy = K[:, 0, 0].view(B, 1, 1, 1), K[:, 1, 1].view(B, 1, 1, 1)
    cx, cy = K[:, 0, 2].view(B, 1, 1, 1), K[:, 1, 2].view(B, 1, 1, 1)
    rr = torch.arange(H).view(1, 1, H, 1)
    cc = torch.arange(W).view(1, 1, 1, W)
    ok = torch.ones_like(depth, dtype=torch.bool) if clip is None else (depth > clip[0]) & (depth < clip[1])
    nan = torch.tensor(float("nan"), dtype=depth.dtype)
    z = torch.where(ok, depth, nan)
    x = torch.where(ok, (cc - cx) * z / fx, nan)
    y = torch.where(ok, (rr - cy) * z / fy, nan)
    return torch.stack([x, y, z], -1).reshape(B, -1, 3), ok.reshape(B, -1)


def pc2depth_tensor(pc: torch.Tensor, valid: torch.Tensor, K: torch.Tensor, image_size):
    """Z-buffer projection (sd:212-265): nearest surviving z per pixel, untouched pixels 0, plus hit mask.

    c = round_half_even(x fx / z + cx), r likewise; a point survives iff it lands in the frame, is
    flagged valid and has z > 0 (NaN coordinates fail every comparison and drop out).
    """
    B, N, _ = pc.shape
    H, W = image_size
    x, y, z = pc[..., 0], pc[..., 1], pc[..., 2]
    c = torch.round(x * K[:, 0, 0, None] / z + K[:, 0, 2, None]).to(torch.long)
    r = torch.round(y * K[:, 1, 1, None] / z + K[:, 1, 2, None]).to(torch.long)
    keep = (c >= 0) & (c < W) & (r >= 0) & (r < H) & valid & (z > 0)
    b = torch.arange(B).view(B, 1).expand(B, N)
    lin = (b * H * W + r * W + c)[keep]
    depth = torch.zeros(B * H * W).scatter_reduce(0, lin, z[keep], reduce="amin", include_self=False)
    mask = torch.zeros(B * H * W, dtype=torch.bool)
    mask[lin] = True
    return depth.view(B, 1, H, W).to(torch.float32), mask.view(B, 1, H, W)


def reproject_tensor(depth: torch.Tensor, K: torch.Tensor, pose: torch.Tensor, clip=(0.0, 10.0)):
    """Unproject, move by the SE(3) pose (p R^T + t) and z-buffer back into the same camera (sd:268-286)."""
    H, W = depth.shape[-2:]
    pc, ok = depth2pc_tensor(depth, K, clip)
    pc = torch.matmul(pc, pose[:, :3, :3].transpose(-1, -2)) + pose[:, None, :3, 3]
    return pc2depth_tensor(pc, ok, K, (H, W))


def project_cloud(cloud: np.ndarray, pose: np.ndarray, K: np.ndarray, S: int):
    """The generator's per-scene form: float32 numpy rigid move, then a batch-of-one z-buffer (sd:2531-2547)."""
    moved = cloud @ pose[:3, :3].T + pose[:3, 3]
    pc = torch.tensor(moved[None])
    return pc2depth_tensor(pc, torch.ones(pc.shape[:2], dtype=torch.bool), torch.tensor(K[None]), (S, S))


def occlusion_filter(depth_rpj: torch.Tensor, mask_rpj: torch.Tensor):
    """3x3 min over valid neighbours; pixels >= 0.0375 m behind it take that minimum (sd:446-463)."""
    import torch.nn.functional as F
    pre = depth_rpj.clone()
    pre[~mask_rpj] = float("inf")
    mn = -F.max_pool2d(-pre, kernel_size=3, stride=1, padding=1)
    keep = (depth_rpj - mn) < 0.0375
    return torch.where(keep, depth_rpj, mn), mask_rpj


def random_sample_transform(K: np.ndarray, image_size: int = 256) -> np.ndarray:
    """Frustum-bounded random rotation, zero translation (sd:377-415).  Draw order on the legacy numpy stream:
    rand(B) theta, rand(B) phi, rand(B) psi, randn(B,3) (multiplied by 0)."""
    from scipy.spatial.transform import Rotation
    B = K.shape[0]
    fx, fy, cx, cy = K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]
    h = w = image_size
    th0, th1 = -np.arctan((h - cy) / fy), np.arctan(cy / fy)
    ph0, ph1 = -np.arctan(cx / fx), np.arctan((w - cx) / fx)
    theta = np.random.rand(B) * (th1 - th0) + th0
    phi = np.random.rand(B) * (ph1 - ph0) + ph0
    psi = np.random.rand(B) * 2 * np.pi - np.pi
    R = Rotation.from_euler("XYZ", np.stack((theta, phi, psi), -1)).as_matrix()
    t = np.random.randn(B, 3) / 3 * 0
    T = np.stack([np.eye(4) for _ in range(B)])
    T[..., :3, :3] = R
    T[..., :3, 3] = t
    return T.astype(np.float32)
