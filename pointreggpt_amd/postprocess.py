"""Host post-processing of the generated views: crop, voxel-grid mean down-sampling, rigid transforms, PLY I/O and the
overlap ratio of `generate_gt.py` — what the reference delegates to open3d (sd:2484-2500, 2641-2680;
generate_gt.py:68-102).  open3d is not in this image, so these follow Open3D 0.17 semantics *as recalled from its
C++ source* and are **parity-unpinned** (DESIGN.md §2): crop bounds inclusive; voxel index =
floor((p - (min_bound - voxel/2)) / voxel), output = per-voxel mean (output order unspecified in open3d: here sorted
by voxel index, deterministic); PLY binary_little_endian with `double x y z`; radius search = any neighbour with
squared distance < r^2.  Pure numpy, vectorised (no per-point Python loop: the reference's KD-tree loop is the
wall-clock tail of config 4).
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np

BBOX_MIN = np.array([-1.5, -1.5, 0.5])    # sd:2348
BBOX_MAX = np.array([1.5, 1.5, 3.5])


def crop_aabb(pts: np.ndarray, lo=BBOX_MIN, hi=BBOX_MAX) -> np.ndarray:
    """PointCloud.crop(AxisAlignedBoundingBox): keep lo <= p <= hi (inclusive)."""
    pts = np.asarray(pts)
    if len(pts) == 0:
        return pts.reshape(0, 3)
    keep = np.all((pts >= lo) & (pts <= hi), axis=1)
    return pts[keep]


def transform(pts: np.ndarray, T: np.ndarray) -> np.ndarray:
    """PointCloud.transform(T): p' = R p + t, float64."""
    pts = np.asarray(pts, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    return pts @ T[:3, :3].T + T[:3, 3]


def voxel_down_sample(pts: np.ndarray, voxel_size: float) -> np.ndarray:
    """PointCloud.voxel_down_sample: mean of the points of every occupied voxel (float64)."""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    if len(pts) == 0:
        return pts
    if voxel_size <= 0:
        raise ValueError("voxel_size <= 0")
    origin = pts.min(axis=0) - voxel_size * 0.5
    idx = np.floor((pts - origin) / voxel_size).astype(np.int64)
    dims = idx.max(axis=0) + 1
    if float(dims[0]) * float(dims[1]) * float(dims[2]) >= 2 ** 62:
        raise ValueError("voxel_size is too small")          # open3d raises the same way
    key = (idx[:, 0] * dims[1] + idx[:, 1]) * dims[2] + idx[:, 2]
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    starts = np.flatnonzero(np.r_[True, key_s[1:] != key_s[:-1]])
    counts = np.diff(np.r_[starts, len(key_s)])
    sums = np.add.reduceat(pts[order], starts, axis=0)
    return sums / counts[:, None]


# ------------------------------------------------------------------------------------------------------------------
# PLY (what o3d.io.write_point_cloud / read_point_cloud exchange; the dataloaders only need the N x 3 points)
# ------------------------------------------------------------------------------------------------------------------
def write_ply(path: str, pts: np.ndarray) -> None:
    pts = np.ascontiguousarray(np.asarray(pts, dtype="<f8").reshape(-1, 3))
    header = ("ply\nformat binary_little_endian 1.0\ncomment Created by pointreggpt_amd\n"
              f"element vertex {len(pts)}\nproperty double x\nproperty double y\nproperty double z\nend_header\n")
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(pts.tobytes())
    os.replace(tmp, path)


_PLY_TYPES = {"double": "<f8", "float64": "<f8", "float": "<f4", "float32": "<f4", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def read_ply(path: str) -> np.ndarray:
    """(N,3) float64 points of a binary-little-endian or ascii PLY (vertex element, x/y/z properties)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties on vertices are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if not all(k in names for k in "xyz"):
            raise ValueError(f"{path}: vertex element lacks x/y/z")
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(n * np.dtype(props).itemsize), dtype=np.dtype(props), count=n)
            return np.stack([data["x"], data["y"], data["z"]], axis=1).astype(np.float64)
        if fmt == "ascii":
            arr = np.loadtxt(f, max_rows=n, ndmin=2)
            return arr[:, [names.index("x"), names.index("y"), names.index("z")]].astype(np.float64)
        raise ValueError(f"{path}: unsupported PLY format {fmt}")


# ------------------------------------------------------------------------------------------------------------------
# overlap ratio (generate_gt.py:68-102) with a uniform grid instead of a per-point KD-tree loop
# ------------------------------------------------------------------------------------------------------------------
def _has_neighbour(query: np.ndarray, ref: np.ndarray, radius: float) -> np.ndarray:
    """bool[len(query)]: some ref point lies strictly within `radius` of the query point."""
    if len(query) == 0 or len(ref) == 0:
        return np.zeros(len(query), dtype=bool)
    cell = radius
    origin = np.minimum(query.min(0), ref.min(0)) - cell
    rc = np.floor((ref - origin) / cell).astype(np.int64)
    qc = np.floor((query - origin) / cell).astype(np.int64)
    dims = np.maximum(rc.max(0), qc.max(0)) + 2
    rkey = (rc[:, 0] * dims[1] + rc[:, 1]) * dims[2] + rc[:, 2]
    order = np.argsort(rkey, kind="stable")
    rkey_s, ref_s = rkey[order], ref[order]
    found = np.zeros(len(query), dtype=bool)
    r2 = radius * radius
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                todo = np.flatnonzero(~found)
                if len(todo) == 0:
                    return found
                c = qc[todo] + (dx, dy, dz)
                key = (c[:, 0] * dims[1] + c[:, 1]) * dims[2] + c[:, 2]
                lo = np.searchsorted(rkey_s, key, side="left")
                hi = np.searchsorted(rkey_s, key, side="right")
                # cells hold few points after the 0.025 voxel grid (cell 0.0375): walk them in lock step
                width = int((hi - lo).max()) if len(lo) else 0
                for k in range(width):
                    j = lo + k
                    ok = j < hi
                    jj = np.where(ok, j, 0)
                    d2 = ((ref_s[jj] - query[todo]) ** 2).sum(1)
                    hit = ok & (d2 < r2)
                    found[todo[hit]] = True
    return found


def compute_overlap_ratio(pc1: np.ndarray, pc2: np.ndarray, voxel_size: float = 0.025, overlap_factor: float = 1.5,
                          is_down_sample: bool = True) -> Tuple[float, float]:
    """Fraction of (down-sampled) points of each cloud with a neighbour of the other within 1.5 voxels."""
    if is_down_sample:
        pc1, pc2 = voxel_down_sample(pc1, voxel_size), voxel_down_sample(pc2, voxel_size)
    r = voxel_size * overlap_factor
    with np.errstate(invalid="ignore", divide="ignore"):
        o1 = np.float64(_has_neighbour(pc1, pc2, r).sum()) / np.float64(len(pc1))
        o2 = np.float64(_has_neighbour(pc2, pc1, r).sum()) / np.float64(len(pc2))
    return float(o1), float(o2)


# ------------------------------------------------------------------------------------------------------------------
# image side files (torchvision.utils.save_image / cv2.imwrite equivalents; real-data parity unpinned)
# ------------------------------------------------------------------------------------------------------------------
def save_image01(img01: np.ndarray, path: str) -> None:
    """utils.save_image of a (1,H,W) or (H,W) tensor in [0,1]: x*255 + 0.5, clamp, uint8, grey replicated to RGB."""
    from PIL import Image
    a = np.asarray(img01, dtype=np.float32)
    a = a.reshape(a.shape[-2], a.shape[-1])
    u8 = np.clip(a * 255.0 + 0.5, 0, 255).astype(np.uint8)
    Image.fromarray(np.stack([u8] * 3, axis=-1), mode="RGB").save(path)


def save_depth16(depth01: np.ndarray, path: str) -> None:
    """cv2.imwrite(path, (depth * 1e4).astype(uint16)) (sd:2618-2620): 0.1 mm units... of the 10 m-normalised depth."""
    from PIL import Image
    a = np.asarray(depth01, dtype=np.float32)
    a = a.reshape(a.shape[-2], a.shape[-1])
    Image.fromarray((a * 1e4).astype(np.uint16)).save(path)
