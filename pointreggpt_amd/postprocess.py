"""Host post-processing of the generated views: crop, voxel-grid mean down-sampling, rigid transforms, PLY I/O and the
overlap ratio of `generate_gt.py` — what the reference delegates to open3d (sd:2484-2500, 2641-2680;
generate_gt.py:68-102).  open3d is not in this image, so these follow Open3D 0.17 semantics *as recalled from its
C++ source* and are **parity-unpinned** (DESIGN.md §2): crop bounds inclusive; voxel index =
floor((p - (min_bound - voxel/2)) / voxel), output = per-voxel mean (output order unspecified in open3d: here sorted
by voxel index, deterministic); PLY binary_little_endian with `double x y z`; radius search = any neighbour with
squared distance < r^2.

Two layers: the numpy forms below are the specification (and what the CPU tests compare with); `native_*` and `WriterPool`
run the same arithmetic in the library's C++ (csrc/hostpool.cpp, include/prg.h "Host post-processing") — the writer pool
produces a batch's files on worker threads while the GPU samples the next batch.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

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


# ------------------------------------------------------------------------------------------------------------------
# native (C++) forms: csrc/hostpool.cpp behind the C-ABI
# ------------------------------------------------------------------------------------------------------------------
def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, 3))


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def native_voxel_down_sample(pts, voxel_size: float) -> np.ndarray:
    from . import _lib
    lib = _lib.load()
    p = _f64(pts)
    out = np.empty_like(p)
    n = C.c_int64(0)
    _lib.check(lib.prg_host_voxel_down_sample(_dp(p), len(p), float(voxel_size), _dp(out), C.byref(n)),
               "prg_host_voxel_down_sample")
    return out[:n.value].copy()


def native_crop_aabb(pts, lo=BBOX_MIN, hi=BBOX_MAX) -> np.ndarray:
    from . import _lib
    lib = _lib.load()
    p = _f64(pts)
    out = np.empty_like(p)
    n = C.c_int64(0)
    lo64, hi64 = np.ascontiguousarray(lo, dtype=np.float64), np.ascontiguousarray(hi, dtype=np.float64)
    _lib.check(lib.prg_host_crop_aabb(_dp(p), len(p), _dp(lo64), _dp(hi64), _dp(out), C.byref(n)), "prg_host_crop_aabb")
    return out[:n.value].copy()


def native_write_ply(path: str, pts) -> None:
    from . import _lib
    p = _f64(pts)
    _lib.check(_lib.load().prg_host_write_ply(os.fsencode(path), _dp(p), len(p)), "prg_host_write_ply")


def rank_cpu_budget() -> int:
    """CPUs this rank may count on: the affinity mask when sharding.pin_rank_cpus narrowed it to this rank's private share (it
    leaves the marker PRG_PINNED_CPUS), otherwise the CPUs the process may use — the whole host, or a container cpuset / job-wide
    taskset that EVERY rank shares — divided by the ranks of the job (LOCAL_WORLD_SIZE, else WORLD_SIZE)."""
    total = os.cpu_count() or 4
    mine = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else total
    pinned = os.environ.get("PRG_PINNED_CPUS", "")
    if pinned.isdigit() and int(pinned) == mine:
        return max(1, mine)
    ranks = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return max(1, mine // max(1, ranks))


class WriterPool:
    """Asynchronous per-scene output (prg_pool_*): every submit copies its inputs and returns at once; `wait()` blocks
    until all files exist and raises on the first failed job."""

    def __init__(self, threads: int = 0):
        from . import _lib
        self._lib = _lib
        lib = _lib.load()
        if threads <= 0:
            threads = max(2, min(16, rank_cpu_budget() // 2))
        self.threads = threads
        self._h = C.c_void_p()
        _lib.check(lib.prg_pool_create(int(threads), C.byref(self._h)), "prg_pool_create")

    def cloud(self, path: str, xyz: np.ndarray, valid: Optional[np.ndarray] = None, *, pre: Optional[np.ndarray] = None,
              crop: bool = True, voxel: float = 0.025, post: Optional[np.ndarray] = None, lo=BBOX_MIN, hi=BBOX_MAX):
        """xyz (n,3) float64 [valid (n,) bool] -> pre (4x4) -> crop -> voxel mean -> post (4x4) -> PLY at `path`."""
        p = _f64(xyz)
        v = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
        t0 = None if pre is None else np.ascontiguousarray(pre, dtype=np.float64)
        t1 = None if post is None else np.ascontiguousarray(post, dtype=np.float64)
        lo64, hi64 = np.ascontiguousarray(lo, dtype=np.float64), np.ascontiguousarray(hi, dtype=np.float64)
        self._lib.check(self._lib.load().prg_pool_submit_cloud(self._h, os.fsencode(path), _dp(p), len(p), _dp(v), _dp(t0),
                                                               int(crop), _dp(lo64), _dp(hi64), float(voxel), _dp(t1)),
                        "prg_pool_submit_cloud")

    def image01(self, path: str, img: np.ndarray):
        a = np.ascontiguousarray(np.asarray(img, dtype=np.float32).reshape(img.shape[-2], img.shape[-1]))
        self._lib.check(self._lib.load().prg_pool_submit_image(self._h, os.fsencode(path), _dp(a), a.shape[0], a.shape[1], 0))

    def depth16(self, path: str, img: np.ndarray):
        a = np.ascontiguousarray(np.asarray(img, dtype=np.float32).reshape(img.shape[-2], img.shape[-1]))
        self._lib.check(self._lib.load().prg_pool_submit_image(self._h, os.fsencode(path), _dp(a), a.shape[0], a.shape[1], 1))

    def text(self, path: str, values: np.ndarray):
        a = np.ascontiguousarray(np.atleast_2d(np.asarray(values, dtype=np.float64)))
        self._lib.check(self._lib.load().prg_pool_submit_text(self._h, os.fsencode(path), _dp(a), a.shape[0], a.shape[1]))

    def wait(self) -> int:
        n = C.c_int64(0)
        self._lib.check(self._lib.load().prg_pool_wait(self._h, C.byref(n)), "prg_pool_wait")
        return int(n.value)

    def close(self):
        if self._h:
            self._lib.load().prg_pool_destroy(self._h)
            self._h = C.c_void_p()

    # `with WriterPool() as pool:` — leaving the block normally waits for every file and raises the first worker error;
    # leaving it on an exception lets the already queued jobs finish (nothing is half-written), reports a worker error
    # on stderr without masking the original exception, and tears the pool down.  Callers submit a batch's resume marker
    # only after `wait()` succeeded, so an aborted run never leaves a marker behind.
    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        worker_error = None
        try:
            if self._h:
                self.wait()
        except Exception as e:          # noqa: BLE001 — surfaced below
            worker_error = e
        self.close()
        if worker_error is not None:
            if exc_type is None:
                raise worker_error
            import sys
            print(f"writer pool: a queued job also failed while unwinding: {worker_error}", file=sys.stderr)
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------------------
# overlap ratios on the GPU (generate_gt.py:68-102): voxel grids in C++, neighbour-existence counts in one HIP launch
# ------------------------------------------------------------------------------------------------------------------
def overlap_ratios_hip(pairs, voxel_size: float = 0.025, overlap_factor: float = 1.5, is_down_sample: bool = True,
                       device="cuda"):
    """[(src (n,3), tgt (m,3)), ...] -> [(overlap_src, overlap_tgt), ...] exactly as compute_overlap_ratio defines them:
    both clouds voxel-down-sampled, then the fraction of points with a point of the other cloud strictly within
    overlap_factor * voxel_size.  One prg_overlap_counts launch for the whole list (float64 all-pairs test)."""
    import torch

    from . import _lib
    lib = _lib.load()
    _lib.require_gpu()
    if not pairs:
        return []
    clouds = []
    for a, b in pairs:
        for c in (a, b):
            clouds.append(native_voxel_down_sample(c, voxel_size) if is_down_sample else _f64(c))
    sizes = np.array([len(c) for c in clouds], dtype=np.int64)
    offs = np.zeros(len(clouds) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(sizes)
    out = []
    if offs[-1] == 0 or sizes.max() == 0:
        return [(float("nan"), float("nan"))] * len(pairs)
    pts = torch.from_numpy(np.concatenate([c for c in clouds if len(c)], axis=0)).to(device)
    d_offs = torch.from_numpy(offs).to(device)
    counts = torch.empty((len(pairs), 2), dtype=torch.int32, device=device)
    _lib.check(lib.prg_overlap_counts(_lib.ptr(pts), _lib.ptr(d_offs), len(pairs), int(sizes.max()),
                                      float(voxel_size * overlap_factor), _lib.ptr(counts), _lib.stream_ptr()),
               "prg_overlap_counts")
    cnt = counts.cpu().numpy().astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(len(pairs)):
            out.append((float(cnt[i, 0] / np.float64(sizes[2 * i])), float(cnt[i, 1] / np.float64(sizes[2 * i + 1]))))
    return out
