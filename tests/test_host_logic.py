"""CPU tests (-m "not gpu"): host-side logic of the product, the C-ABI library's exports, multi-rank sharding."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from pointreggpt_amd import _lib, geometry as G, sharding, synthetic, weights as W
from pointreggpt_amd.diffusion import GaussianDiffusion, ddim_times, make_schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    """The built .so loads (no GPU needed) and exports exactly the entry points include/prg.h declares."""
    hdr = open(os.path.join(ROOT, "include", "prg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(prg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().prg_abi_version() == 1


def test_param_count_matches_library():
    lib = _lib.load()
    from pointreggpt_amd.unet import _cfg_c
    for cfg in (W.unet_config(64), W.maskunet_config(64), W.unet_config(8), W.maskunet_config(16)):
        assert lib.prg_unet_param_count(ctypes.byref(_cfg_c(cfg))) == W.num_params(cfg)
    bad = _cfg_c(W.unet_config(64))
    bad.dim = 7
    assert lib.prg_unet_param_count(ctypes.byref(bad)) < 0
    assert b"dim" in lib.prg_last_error()


def test_product_has_no_cpu_path():
    from pointreggpt_amd.unet import Unet
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.PrgError):
        Unet(16).init_synthetic(0)
    with pytest.raises(_lib.PrgError):
        G.pc2depth_tensor(torch.zeros((1, 4, 3)), None, torch.eye(3)[None], image_size=(8, 8))
    # and nothing under the package imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "pointreggpt_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_host_geometry_against_reference_goldens(golden):
    g = golden("G2_intrinsic_transform")
    for S in (32, 64, 128, 256):
        assert np.array_equal(G.intrinsic_transform(g["K"], S, S), g[f"S{S}_batched"])
        assert np.array_equal(G.intrinsic_transform(g["K"][0], S, S), g[f"S{S}"][0])
    assert np.array_equal(G.intrinsic_transform(g["K"][4]), g["none"])
    assert np.array_equal(G.candidate_intrinsics(), g["K"])
    g = golden("G3_random_sample_pose")
    for s in (0, 1, 12345):
        np.random.seed(s)
        assert np.array_equal(G.random_sample_pose(4), g[f"pose_seed{s}"])
        assert np.array_equal(np.random.rand(2), g[f"after_seed{s}"])
        np.random.seed(s)
        assert np.array_equal(G.random_sample_intrinsic(16), g[f"intr_seed{s}"])
    g18 = golden("G18_refine_occlusion_transform")
    for s_ in (0, 7):
        np.random.seed(s_)
        assert np.array_equal(G.random_sample_transform(g18["rst_K"], 64), g18[f"rst_seed{s_}"])
        assert np.array_equal(np.random.rand(2), g18[f"rst_after_seed{s_}"])
    K = torch.tensor(g["pose_seed0"][:, :3, :3])
    assert torch.equal(G.param_vector(K), torch.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], -1))


class _FakeNet:
    channels = out_dim = 1
    random_or_learned_sinusoidal_cond = False


def test_schedule_and_step_table_against_reference_goldens(golden):
    g = golden("G1_schedule")
    for T, pre in ((1000, ""), (8, "T8_")):
        for k, v in make_schedule(T).items():
            assert np.array_equal(v.numpy(), g[pre + k]), k
    for n in (5, 50, 250):
        assert ddim_times(1000, n) == g[f"ddim_times_{n}"].tolist()
    d = GaussianDiffusion(_FakeNet(), image_size=32, timesteps=1000)
    rows = d.step_table()
    assert len(rows) == 1000 and [r["t"] for r in rows] == list(range(999, -1, -1)) and d.n_draws == 1000
    assert rows[-1]["sigma"] == 0.0 and rows[-1]["c_x0"] == 1.0 and rows[-1]["c_x"] == 0.0     # t = 0
    assert rows[0]["c_x0"] == float(g["posterior_mean_coef1"][999]) and rows[0]["c_eps"] == 0.0
    assert all(r["clip_pred"] == 2 for r in rows)          # ancestral: clamp x0 after the DDNM replacement (sd:1250)
    assert rows[3]["sigma"] == float(np.exp(np.float32(0.5) * g["posterior_log_variance_clipped"][996]))
    d = GaussianDiffusion(_FakeNet(), image_size=32, timesteps=1000, sampling_timesteps=250)
    rows = d.step_table()
    assert len(rows) == 250 and rows[0]["t"] == 999 and rows[1]["t"] == 995 and d.n_draws == 250
    assert rows[-1] == dict(rows[-1], c_x0=1.0, c_x=0.0, c_eps=0.0, sigma=0.0) and all(r["clip_pred"] == 1 for r in rows)
    a, an = g["alphas_cumprod"][999], g["alphas_cumprod"][995]
    sigma = np.sqrt((1 - a / an) * (1 - an) / (1 - a), dtype=np.float32)
    assert abs(rows[0]["sigma"] - float(sigma)) <= 1e-7 and abs(rows[0]["c_x0"] - float(np.sqrt(an))) <= 1e-7
    with pytest.raises(ValueError):
        GaussianDiffusion(_FakeNet(), image_size=32, objective="pred_noise")


def test_synthetic_scenes_are_index_keyed():
    d1, K1, P1 = synthetic.synth_batch(0, [5, 9], 64)
    d2, K2, P2 = synthetic.synth_batch(0, [9, 5, 7], 64)
    assert np.array_equal(d1[0], d2[1]) and np.array_equal(K1[1], K2[0]) and np.array_equal(P1[0], P2[1])
    assert d1.dtype == np.float32 and d1.shape == (2, 1, 64, 64) and 0.0 <= d1.min() and d1.max() <= 0.35
    assert 0.02 < (d1 == 0).mean() < 0.1
    assert synthetic.noise_seed(0, 5) != synthetic.noise_seed(0, 6) and synthetic.noise_seed(0, 5) == synthetic.noise_seed(0, 5)
    keys = {synthetic.noise_seed(3, i, s) for i in range(50) for s in range(50)}
    assert len(keys) == 2500 and synthetic.noise_seed(3, 1, 0) == synthetic.noise_seed(3, 1)   # no scene/sample aliasing


def test_flatten_state_dict_and_checkpoint_layouts():
    from pointreggpt_amd.unet import flatten_state_dict
    cfg = W.unet_config(8)
    sd = W.synth_state_dict(cfg, 3)
    flat = flatten_state_dict(cfg, sd)
    assert flat.dtype == np.float32 and flat.size == W.num_params(cfg)
    assert np.array_equal(flat[:8 * 49], sd["init_conv.weight"].reshape(-1).numpy())
    # EMA-style checkpoint (ema_model. + model. prefixes) and plain 'model' fallback both resolve
    ema = {"initted": torch.tensor(True), **{"ema_model.model." + k: v for k, v in sd.items()}}
    got = W.unet_state_from_checkpoint({"ema": ema, "model": {}}, cfg)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    got = W.unet_state_from_checkpoint({"model": {"model." + k: v for k, v in sd.items()}}, cfg)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    with pytest.raises(KeyError):
        W.unet_state_from_checkpoint({"model": {}}, cfg)
    bad = dict(sd)
    bad["init_conv.bias"] = torch.zeros(3)
    with pytest.raises(ValueError):
        flatten_state_dict(cfg, bad)


def test_shard_range_partitions_exactly():
    for start, stop, batch in ((0, 10000, 64), (7, 8, 4), (0, 0, 4), (3, 1003, 1), (0, 130, 64)):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                a, b = sharding.shard_range(start, stop, r, world, batch)
                assert start <= a <= b <= stop and ((a - start) % batch == 0 or a == stop)
                got += list(range(a, b))
            assert got == list(range(start, stop))
    assert sharding.num_to_groups(10, 4) == [4, 4, 2] and sharding.num_to_groups(8, 4) == [4, 4]


_WORKER = r"""
import os, sys, json
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pointreggpt_amd import sharding, synthetic
rank, world, _ = sharding.rank_world()
dist.init_process_group("gloo", rank=rank, world_size=world)
a, b = sharding.shard_range(0, 37, rank, world, batch=4)
mine = torch.zeros(37, dtype=torch.int64); mine[a:b] = 1
# per-scene noise keys and inputs must not depend on the rank that owns the scene
keys = torch.tensor([synthetic.noise_seed(0, i) % (2**62) for i in range(a, b)] + [0] * (37 - (b - a)), dtype=torch.int64)
dist.all_reduce(mine)
t = torch.tensor([float(b - a)], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"cover": mine.tolist(), "max_load": t.item()}))
dist.barrier(); dist.destroy_process_group()
"""


def test_two_rank_sharding_over_gloo(tmp_path):
    """world_size 2 on CPU (gloo): the ranks' scene ranges tile the request exactly, MAX-reduction works."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29731", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["cover"] == [1] * 37 and res["max_load"] == 20.0


def test_real_data_input_side(tmp_path, monkeypatch):
    """the 3DMatch input side of Generator.generate (sd:2352-2361, 2397-2459) on a synthetic
    layout — train_info.pkl -> <cloud>.info.txt -> frame-XXXXXX.depth.png (uint16 mm) -> Resize(S, NEAREST) ->
    CenterCrop(S) -> x 1e-4 -> (> 1 -> 0), and the matching intrinsics.  torchvision is absent from the image, so the
    expectation is a direct restatement of its documented arithmetic (resize: short side -> S, long side int(S*long/short),
    nearest source pixel floor((i + 0.5) * scale); crop offset int(round((n - S) / 2))): parity-unpinned, but pinned to that."""
    import pickle
    from PIL import Image
    from pointreggpt_amd.generator import PAIRS_PER_LAP, Generator
    S = 64
    rng = np.random.default_rng(22)
    raw = rng.integers(300, 9000, size=(480, 640)).astype(np.uint16)
    raw[:40, :40] = 0                          # holes
    raw[100:120, 200:260] = 12000              # > 10 m: dropped by the (> 1 -> 0) rule
    root = tmp_path / "3dmatch"
    (root / "sceneA" / "seq-01").mkdir(parents=True)
    Image.fromarray(raw).save(root / "sceneA" / "seq-01" / "frame-000012.depth.png")
    (root / "sceneB" / "seq-02").mkdir(parents=True)
    Image.fromarray(raw[::-1].copy()).save(root / "sceneB" / "seq-02" / "frame-000007.depth.png")
    Kraw = np.array([[585.0, 0, 320.0], [0, 585.0, 240.0], [0, 0, 1.0]])
    np.savetxt(root / "sceneA" / "camera-intrinsics.txt", Kraw)
    np.savetxt(root / "sceneB" / "camera-intrinsics.txt", Kraw)
    monkeypatch.chdir(tmp_path)
    (tmp_path / "dataset/indoor/metadata").mkdir(parents=True)
    (tmp_path / "dataset/indoor/data/train/sceneA").mkdir(parents=True)
    (tmp_path / "dataset/indoor/data/train/sceneB").mkdir(parents=True)
    info = {"src": ["train/sceneA/cloud_bin_0.pth"], "tgt": ["train/sceneB/cloud_bin_3.pth"]}
    (tmp_path / "dataset/indoor/data/train/sceneA/cloud_bin_0.info.txt").write_text("sceneA seq-01 12 62\n")
    (tmp_path / "dataset/indoor/data/train/sceneB/cloud_bin_3.info.txt").write_text("sceneB seq-02 7 57\n")

    class _Diff:
        image_size = S

    gen = Generator(_Diff(), str(root), samples_folder=str(tmp_path / "out"), synthetic_seed=None, device="cpu")
    # the index wraps every PAIRS_PER_LAP scenes; the fake list has one entry, so patch the modulus via a 1-entry view
    lst = {"src": info["src"] * 1, "tgt": info["tgt"] * 1}
    depth, K = gen._real_scene(0, {k: v * PAIRS_PER_LAP for k, v in lst.items()}, tmp_path)
    # expectation
    nw, nh = int(S * 640 / 480), S
    xs = np.floor((np.arange(nw) + 0.5) * 640 / nw).astype(int)
    ys = np.floor((np.arange(nh) + 0.5) * 480 / nh).astype(int)
    left = int(round((nw - S) / 2.0))
    exp = raw[ys][:, xs][:, left:left + S].astype(np.float32) * np.float32(1e-4)
    exp[exp > 1] = 0
    assert depth.dtype == np.float32 and depth.shape == (S, S) and np.array_equal(depth, exp)
    assert (depth == 0).sum() > 0 and depth.max() <= 1.0
    assert np.array_equal(K, G.intrinsic_transform(Kraw, resize=S, centercrop=S).astype(np.float32))
    # odd laps swap src and tgt (sd:2397-2410)
    depth2, _ = gen._real_scene(PAIRS_PER_LAP, {k: v * PAIRS_PER_LAP for k, v in lst.items()}, tmp_path)
    assert np.array_equal(depth2, (raw[::-1][ys][:, xs][:, left:left + S].astype(np.float32) * np.float32(1e-4)) *
                          ((raw[::-1][ys][:, xs][:, left:left + S].astype(np.float32) * np.float32(1e-4)) <= 1))


def test_downsample_equals_two_by_two_taps_over_space_to_depth():
    """The identity conv_w256.hip's Downsample mode rests on (csrc/conv.hip: s2d_equivalent_weights): Conv2d(C, Co, 4, 2, 1)
    (sd:596-597) of x equals a 3x3 / pad 1 convolution, with only taps (1..2, 1..2) non-zero, of the space-to-depth view
    [4C, H/2, W/2] of x shifted by (1, 1) — virtual channel (2 dy + dx) C + c of block (y', x') is x[c, 2y'+dy-1, 2x'+dx-1]
    (zero outside), weight [co][(2 dy + dx) C + c][1 + by][1 + bx] = W[co][c][2 by + dy][2 bx + dx]."""
    import torch
    g = torch.Generator().manual_seed(42)
    B, C, Co, H, Wd = 2, 3, 5, 8, 12
    x = torch.randn((B, C, H, Wd), generator=g, dtype=torch.float64)
    w = torch.randn((Co, C, 4, 4), generator=g, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(x, w, None, stride=2, padding=1)
    xs = torch.nn.functional.pad(x, (1, 1, 1, 1))[:, :, :H + 1, :Wd + 1]      # xs[y, x] = x[y - 1, x - 1]; one extra block row / col
    xs = torch.nn.functional.pad(xs, (0, 1, 0, 1))                              # even size: (H + 2, Wd + 2)
    v = torch.stack([xs[:, :, dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)], dim=1).reshape(B, 4 * C, H // 2 + 1, Wd // 2 + 1)
    weq = torch.zeros((Co, 4 * C, 3, 3), dtype=torch.float64)
    for ky in range(4):
        for kx in range(4):
            by, dy, bx, dx = ky >> 1, ky & 1, kx >> 1, kx & 1
            weq[:, (2 * dy + dx) * C:(2 * dy + dx + 1) * C, 1 + by, 1 + bx] = w[:, :, ky, kx]
    got = torch.nn.functional.conv2d(v, weq, None, padding=1)[:, :, :H // 2, :Wd // 2]
    assert float((got - ref).abs().max()) < 1e-12


def test_job_seed_is_drawn_once_and_published_to_every_rank():
    """the job seed is drawn on rank 0 (secrets.randbits) and published through a c10d store on
    MASTER_ADDR:MASTER_PORT — identical on every rank of a launch, fresh on the next launch even with the SAME rendezvous id
    and port, independent of launcher pids; PRG_JOB_SEED pins it.  (Round 5: the store keys carry the attempt's
    TORCHELASTIC_RESTART_COUNT, which torchrun gives every rank of an attempt alike.)"""
    import socket
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); from pointreggpt_amd import sharding; print(sharding.job_seed())" % ROOT

    def launch(extra=None):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       TORCHELASTIC_RUN_ID="fixed-id", TORCHELASTIC_RESTART_COUNT="0")
            env.pop("PRG_JOB_SEED", None)
            env.update(extra or {})
            procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=300) for p in procs]
        assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
        return [int(o[0].strip().splitlines()[-1]) for o in outs]

    a, b = launch(), launch()
    assert a[0] == a[1] and b[0] == b[1], (a, b)
    assert a[0] != b[0], "two launches with the same rendezvous id must not share a seed"
    assert 0 <= a[0] < (1 << 63)
    assert launch({"PRG_JOB_SEED": "1234"}) == [1234, 1234]


def test_job_seed_survives_a_store_that_outlives_the_attempt():
    """under torchrun MASTER_PORT is the AGENT's store and outlives worker restarts; rank 0 of the restarted
    attempt is then a client of a store that still holds the previous attempt's seed and reader count.  The keys are namespaced
    per attempt (TORCHELASTIC_RESTART_COUNT): with a pre-existing store that already holds attempt 0's keys, the ranks of
    attempt 1 agree on a NEW seed (never the stale one, no early return on the stale reader count)."""
    import socket
    import subprocess
    import sys
    from datetime import timedelta

    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    agent = dist.TCPStore("127.0.0.1", port, None, is_master=True, timeout=timedelta(seconds=60), wait_for_workers=False)
    stale = 4242424242
    agent.set("prg_job_seed/fixed-id/0/0", str(stale))            # what attempt 0 left behind
    agent.add("prg_job_seed/fixed-id/0/0/readers", 2)
    agent.set("prg_job_seed", str(stale))                          # (and the round-4 key names)
    agent.add("prg_job_seed_readers", 2)
    code = "import sys; sys.path.insert(0, %r); from pointreggpt_amd import sharding; print(sharding.job_seed())" % ROOT
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TORCHELASTIC_RUN_ID="fixed-id", TORCHELASTIC_RESTART_COUNT="1")
        env.pop("PRG_JOB_SEED", None)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    seeds = [int(o[0].strip().splitlines()[-1]) for o in outs]
    assert seeds[0] == seeds[1] != stale, seeds
    assert int(agent.get("prg_job_seed/fixed-id/1/0").decode()) == seeds[0]
    del agent


def test_rank_cpu_sets_partition_the_host():
    """Round 5: per-rank CPU placement.  Eight ranks on a 2-socket, 256-thread host whose GPUs 0-3 / 4-7 hang off
    NUMA nodes 0 / 1: every rank gets a disjoint quarter of ITS GPU's node, all sets together cover the host; without topology
    the allowed CPUs are cut evenly; fewer CPUs than ranks = everything shared; PRG_NO_AFFINITY / a single rank leave the process
    alone."""
    from pointreggpt_amd import sharding as S
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    allowed = list(range(256))
    sets = []
    for r in range(8):
        node = r // 4
        cp = S.rank_cpu_set(r, 8, allowed, node_cpus[node], peers_on_node=4, index_on_node=r % 4)
        assert len(cp) == 32 and set(cp) <= set(node_cpus[node])
        sets.append(set(cp))
    assert len(set().union(*sets)) == 256 and sum(len(x) for x in sets) == 256
    flat = [S.rank_cpu_set(r, 8, allowed, []) for r in range(8)]
    assert [len(x) for x in flat] == [32] * 8 and sorted(sum(flat, [])) == allowed
    assert S.rank_cpu_set(5, 8, [0, 1, 2], []) == [0, 1, 2]
    assert S._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert S.pin_rank_cpus(0, 1)["pinned"] is False
    before = os.sched_getaffinity(0)
    os.environ["PRG_NO_AFFINITY"] = "1"
    try:
        assert S.pin_rank_cpus(1, 2)["pinned"] is False and os.sched_getaffinity(0) == before
    finally:
        del os.environ["PRG_NO_AFFINITY"]


def test_pin_rank_cpus_in_a_subprocess():
    """pin_rank_cpus really narrows the calling process (and so every thread it starts later) to its share of the allowed CPUs."""
    import subprocess
    import sys
    code = ("import os, sys, json; sys.path.insert(0, %r); from pointreggpt_amd import sharding as S; "
            "a = sorted(os.sched_getaffinity(0)); i = S.pin_rank_cpus(1, 2); print(json.dumps([a, sorted(os.sched_getaffinity(0)), i]))" % ROOT)
    env = dict(os.environ)
    env.pop("PRG_NO_AFFINITY", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    import json
    before, after, info = json.loads(r.stdout.strip().splitlines()[-1])
    if len(before) >= 2:
        assert info["pinned"] and after == before[len(before) // 2 + (len(before) % 2 > 0 and 0):] or set(after) < set(before)
        assert len(after) == len(before) // 2


def test_bench_contract_line_is_short_for_a_full_result():
    """bench.py's stdout line is built key by key from the full result and stays under 4 KB (round 5's 22 KB line was not parsed by
    the driver): fed with the full round-5 result of the driver's own command (profiles/r05_end_bench_driver_command.json) plus an
    8-rank table, the line keeps every contract key, `roofline`, `cpu_baseline`, and names the sidecar."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05_end_bench_driver_command.json")).read().strip().splitlines()[-1])
    full["per_rank"] = [{"rank": r, "setup_s": 40.0 + r, "warmup_s": 20.0, "timed_s": 86.0 + 0.1 * r, "device_bytes_in_use": 9 << 30,
                         "pinned_cpus": 32, "numa_node": r // 4} for r in range(8)]
    line = bench.contract_line(full, bench.SIDECAR)
    assert len(line) < bench.LINE_LIMIT == 4096 and "\n" not in line
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_mode", "configs4", "e2e_files", "full"):
        assert k in c, k
    assert c["value"] == full["value"] and c["roofline"]["frac"] == full["roofline"]["frac"] and abs(c["roofline"]["traffic"] / full["roofline"]["traffic"] - 1) < 1e-4
    assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-12
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    assert "workload" in c["config"] and "model" not in c["config"]
    assert c["per_rank_timed_s"] == {"min": 86.0, "max": 86.7}
    # a pathological leg cannot break the contract: the optional objects are dropped before the limit is crossed
    full["config"]["workload"] = "w" * 3000
    assert len(bench.contract_line(full, bench.SIDECAR)) < 4096


def test_rank_cpu_budget_divides_a_shared_mask_but_not_a_pinned_one():
    """a mask narrower than the host that pin_rank_cpus did NOT set (container cpuset, job-wide taskset,
    PRG_NO_AFFINITY) is shared by every rank and must be divided by the rank count; a mask pin_rank_cpus set is the rank's own."""
    import json
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip("needs >= 4 CPUs")
    code = ("import os, sys, json; sys.path.insert(0, %r); from pointreggpt_amd import postprocess as PP, sharding as S; "
            "os.sched_setaffinity(0, %r); a = PP.rank_cpu_budget(); i = S.pin_rank_cpus(1, 2); b = PP.rank_cpu_budget(); "
            "print(json.dumps([a, b, i, len(os.sched_getaffinity(0))]))" % (ROOT, allowed[:4]))
    env = {k: v for k, v in os.environ.items() if k not in ("PRG_NO_AFFINITY", "PRG_PINNED_CPUS")}
    env.update(WORLD_SIZE="2", LOCAL_WORLD_SIZE="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    shared, own, info, n_after = json.loads(r.stdout.strip().splitlines()[-1])
    assert shared == 2                       # 4 shared CPUs / 2 ranks (the old code returned 4 per rank)
    assert info["pinned"] and n_after == 2 and own == 2 and info["threads_moved"] >= 1
