"""CPU tests of the host post-processing (crop / voxel grid / PLY / overlap / gt.log) against the slow oracle forms."""
import os

import numpy as np
import pytest

from oracle import postprocess as OP
from pointreggpt_amd import postprocess as PP
from pointreggpt_amd.generator import gather_gt, generate_gt


def _cloud(seed, n=6000):
    rng = np.random.default_rng(seed)
    plane = np.c_[rng.uniform(-1.2, 1.2, n), rng.uniform(-1.0, 1.0, n), 2.0 + 0.1 * rng.standard_normal(n)]
    return plane


def _as_set(a, nd=9):
    return {tuple(np.round(p, nd)) for p in a}


def test_voxel_down_sample_matches_dictionary_form():
    for seed, voxel in ((0, 0.025), (1, 0.05), (2, 0.002)):
        pts = _cloud(seed, 3000)
        got, ref = PP.voxel_down_sample(pts, voxel), OP.voxel_down_sample(pts, voxel)
        assert got.shape == ref.shape and _as_set(got) == _as_set(ref)
    assert PP.voxel_down_sample(np.zeros((0, 3)), 0.025).shape == (0, 3)
    one = PP.voxel_down_sample(np.array([[0.1, 0.2, 0.3]] * 5), 0.025)
    assert one.shape == (1, 3) and np.allclose(one[0], [0.1, 0.2, 0.3])
    with pytest.raises(ValueError):
        PP.voxel_down_sample(_cloud(0, 10), 0.0)


def test_crop_is_inclusive_and_transform_roundtrips():
    pts = np.array([[-1.5, 0, 0.5], [1.5, 1.5, 3.5], [1.5000001, 0, 1], [0, 0, 0.4999], [0, 0, 2]])
    assert len(PP.crop_aabb(pts)) == 3
    T = np.eye(4)
    T[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
    T[:3, 3] = [0.5, -0.25, 1.0]
    q = PP.transform(PP.transform(pts, T), np.linalg.inv(T))
    assert np.allclose(q, pts, atol=1e-12)


def test_ply_roundtrip_and_float_reader(tmp_path):
    pts = _cloud(3, 1234)
    path = str(tmp_path / "a.ply")
    PP.write_ply(path, pts)
    head = open(path, "rb").read(200)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0") and b"property double x" in head
    assert np.array_equal(PP.read_ply(path), pts)
    # a float32 PLY with an extra property (what other writers produce) still yields N x 3
    rec = np.zeros(7, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "u1")])
    rec["x"], rec["y"], rec["z"] = np.arange(7), 1.5, -2.0
    with open(tmp_path / "b.ply", "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 7\nproperty float x\nproperty float y\n"
                b"property float z\nproperty uchar intensity\nend_header\n" + rec.tobytes())
    got = PP.read_ply(str(tmp_path / "b.ply"))
    assert got.shape == (7, 3) and np.allclose(got[:, 0], np.arange(7)) and np.allclose(got[:, 2], -2.0)


def test_overlap_ratio_matches_kdtree_form():
    a = _cloud(5)
    b = _cloud(6) + np.array([0.6, 0.0, 0.0])            # partial overlap
    got, ref = PP.compute_overlap_ratio(a, b), OP.overlap_ratio(a, b)
    assert abs(got[0] - ref[0]) < 1e-12 and abs(got[1] - ref[1]) < 1e-12 and 0.2 < got[0] < 0.9
    far = PP.compute_overlap_ratio(a, b + 10.0)
    assert far == (0.0, 0.0)
    same = PP.compute_overlap_ratio(a, a)
    assert same == (1.0, 1.0)


def test_gt_log_format_is_what_the_dataloaders_parse(tmp_path):
    """example_dataloader/*: line.split('\\t') -> scene_name, int(src), int(tgt), float(overlaps); scene_name.split('-')."""
    root = str(tmp_path)
    for scene in (3, 4):
        d = tmp_path / "ds" / "data" / "scene-{:0>6d}".format(scene)
        d.mkdir(parents=True)
        PP.write_ply(str(d / "sample-000000.cloud.ply"), _cloud(scene))
        PP.write_ply(str(d / "sample-000001.cloud.ply"), _cloud(scene + 10) + [0.3 * (scene - 3), 0, 0])
    d5 = tmp_path / "ds" / "data" / "scene-000005"
    d5.mkdir()
    PP.write_ply(str(d5 / "sample-000000.cloud.ply"), _cloud(1, 500))      # < 1000 points: pair skipped
    PP.write_ply(str(d5 / "sample-000001.cloud.ply"), _cloud(2, 500))
    generate_gt("ds", 3, 6, 2, root=root, overlap="numpy-spec")
    gather_gt("ds", 3, 6, root=root)
    lines = open(tmp_path / "ds" / "metadata" / "gt.log").read().splitlines()
    assert len(lines) == 2
    for line, scene in zip(lines, (3, 4)):
        name, s, t, o1, o2 = line.split("\t")
        assert name == "scene-{:0>6d}".format(scene) and name.split("-")[1] == "{:0>6d}".format(scene)
        assert (int(s), int(t)) == (0, 1) and 0.0 <= float(o1) <= 1.0 and len(o1.split(".")[1]) == 4
    assert open(tmp_path / "ds" / "data" / "scene-000005" / "gt.log").read() == ""
    # idempotent: a second run skips finished scenes and rewrites the same metadata file
    generate_gt("ds", 3, 6, 2, root=root, overlap="numpy-spec")
    gather_gt("ds", 3, 6, root=root)
    assert open(tmp_path / "ds" / "metadata" / "gt.log").read().splitlines() == lines


# ------------------------------------------------------------------------------------------------------------------
# native (C++) host pipeline behind the C-ABI: same arithmetic as the numpy forms above, asynchronous writer pool
# ------------------------------------------------------------------------------------------------------------------
def test_native_voxel_and_crop_match_numpy_forms():
    for seed, voxel in ((0, 0.025), (1, 0.05), (2, 0.002)):
        pts = _cloud(seed, 5000)
        a, b = PP.voxel_down_sample(pts, voxel), PP.native_voxel_down_sample(pts, voxel)
        # same voxels in the same (ascending key) order; means agree to the last bit or two (numpy's reduceat may add pairwise)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-15
    dense = np.random.default_rng(3).uniform(-0.2, 0.2, (40000, 3))          # ~8 points per voxel
    a, b = PP.voxel_down_sample(dense, 0.025), PP.native_voxel_down_sample(dense, 0.025)
    assert a.shape == b.shape and np.abs(a - b).max() <= 1e-15
    assert PP.native_voxel_down_sample(np.zeros((0, 3)), 0.025).shape == (0, 3)
    pts = np.array([[-1.5, 0, 0.5], [1.5, 1.5, 3.5], [1.5000001, 0, 1], [0, 0, 0.4999], [0, 0, 2]])
    assert np.array_equal(PP.native_crop_aabb(pts), PP.crop_aabb(pts))
    from pointreggpt_amd._lib import PrgError
    with pytest.raises(PrgError):
        PP.native_voxel_down_sample(_cloud(0, 10), 0.0)
    with pytest.raises(PrgError):
        PP.native_voxel_down_sample(np.array([[0.0, np.nan, 1.0]]), 0.025)


def test_writer_pool_files_equal_the_python_writers(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(5)
    T = np.eye(4)
    T[:3, :3] = [[0.96, -0.28, 0], [0.28, 0.96, 0], [0, 0, 1]]
    T[:3, 3] = [0.1, -0.2, 0.3]
    pool = PP.WriterPool(3)
    clouds, imgs = [], []
    for i in range(12):
        pts = _cloud(10 + i, 4000)
        valid = rng.random(len(pts)) > 0.3
        img = rng.random((48, 64)).astype(np.float32) * 1.1 - 0.05           # exercises both clamps
        clouds.append((pts, valid))
        imgs.append(img)
        pool.cloud(str(tmp_path / f"c{i}.ply"), pts, valid, pre=T, crop=True, voxel=0.025, post=np.linalg.inv(T))
        pool.cloud(str(tmp_path / f"m{i}.ply"), pts.astype(np.float32), None, crop=False, voxel=0.025)
        pool.image01(str(tmp_path / f"i{i}.png"), img)
        pool.depth16(str(tmp_path / f"d{i}.png"), np.clip(img, 0, 1))
        pool.text(str(tmp_path / f"t{i}.txt"), T.astype(np.float32))
    assert pool.wait() == 12 * 5
    for i, ((pts, valid), img) in enumerate(zip(clouds, imgs)):
        ref = PP.transform(PP.voxel_down_sample(PP.crop_aabb(PP.transform(pts[valid], T)), 0.025), np.linalg.inv(T))
        got = PP.read_ply(str(tmp_path / f"c{i}.ply"))
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-12
        ref = PP.voxel_down_sample(pts.astype(np.float32), 0.025)
        got = PP.read_ply(str(tmp_path / f"m{i}.ply"))
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-15
        PP.save_image01(img, str(tmp_path / "ref.png"))
        assert np.array_equal(np.asarray(Image.open(tmp_path / f"i{i}.png")), np.asarray(Image.open(tmp_path / "ref.png")))
        PP.save_depth16(np.clip(img, 0, 1), str(tmp_path / "refd.png"))
        d16 = np.asarray(Image.open(tmp_path / f"d{i}.png"))
        assert d16.dtype == np.uint16 and np.array_equal(d16, np.asarray(Image.open(tmp_path / "refd.png")))
        np.savetxt(str(tmp_path / "ref.txt"), T.astype(np.float32))
        assert open(tmp_path / f"t{i}.txt").read() == open(tmp_path / "ref.txt").read()
    # a failing job (unwritable directory) surfaces at wait(), and the pool stays usable
    pool.text(str(tmp_path / "no_such_dir" / "x.txt"), T)
    from pointreggpt_amd._lib import PrgError
    with pytest.raises(PrgError):
        pool.wait()
    pool.text(str(tmp_path / "ok.txt"), T)
    pool.wait()
    assert (tmp_path / "ok.txt").is_file()
    pool.close()
