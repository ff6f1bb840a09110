"""Oracle for the host post-processing (TEST INFRASTRUCTURE — see oracle/__init__.py): the obvious slow forms of
voxel-mean down-sampling (dictionary of voxel -> points) and of the overlap ratio (scipy cKDTree radius queries,
generate_gt.py:68-102).  open3d itself is absent, so this pins the product against the *recalled* Open3D 0.17
semantics, not against open3d: parity unpinned."""
from collections import defaultdict

import numpy as np
from scipy.spatial import cKDTree


def voxel_down_sample(pts, voxel):
    pts = np.asarray(pts, dtype=np.float64)
    origin = pts.min(0) - voxel * 0.5
    cells = defaultdict(list)
    for p in pts:
        cells[tuple(np.floor((p - origin) / voxel).astype(np.int64))].append(p)
    return np.array([np.mean(v, axis=0) for v in cells.values()])


def overlap_ratio(pc1, pc2, voxel=0.025, factor=1.5):
    a, b = voxel_down_sample(pc1, voxel), voxel_down_sample(pc2, voxel)
    r = voxel * factor
    ta, tb = cKDTree(a), cKDTree(b)
    # strictly-inside radius, like nanoflann's RadiusResultSet (dist < r^2)
    da, _ = tb.query(a, k=1)
    db, _ = ta.query(b, k=1)
    return float((da < r).sum() / len(a)), float((db < r).sum() / len(b))
