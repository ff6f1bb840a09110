"""The CPU oracle (oracle/) against golden vectors produced by the real reference (tools/make_goldens.py).

This is what pins the oracle: every function on the hot path is compared with the reference's own output on
the same inputs.  torch-CPU arithmetic is shared with the reference, so most comparisons are bit-exact; where
the oracle's expression order legitimately differs (functional vs module code paths never do here) a tolerance
is written next to the assert."""
import json
import os

import numpy as np
import torch

from oracle import diffusion as OD
from oracle import geometry as OG
from oracle import unet as OU
from pointreggpt_amd import weights as W

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
T = torch.tensor


def test_state_dict_spec_matches_reference():
    spec = json.load(open(os.path.join(GOLDEN, "state_dict_spec.json")))
    for key, cfg in (("unet64", W.unet_config(64)), ("mask64", W.maskunet_config(64)), ("unet8", W.unet_config(8))):
        mine = [[k, list(v)] for k, v in W.param_spec(cfg).items()]
        assert mine == spec[key]
    assert W.num_params(W.unet_config(64)) == 38376833       # SURVEY §2 [probe]
    assert W.num_params(W.maskunet_config(64)) == 34096769


def test_G1_schedule(golden):
    g = golden("G1_schedule")
    for T_, pre in ((1000, ""), (8, "T8_")):
        s = OD.schedule(T_)
        for k, v in s.items():
            assert np.array_equal(v.numpy(), g[pre + k]), k
    s = OD.schedule(1000)   # known-answer values quoted in SURVEY §8a
    assert abs(float(s["betas"][0]) - 3.0027920729e-04) < 1e-12
    assert float(s["betas"][999]) == np.float32(0.999)
    assert float(s["posterior_mean_coef1"][0]) == 1.0 and float(s["posterior_mean_coef2"][0]) == 0.0
    for steps in (5, 50, 250):
        pairs = OD.ddim_time_pairs(1000, steps)
        times = [p[0] for p in pairs] + [pairs[-1][1]]
        assert times == g[f"ddim_times_{steps}"].tolist()


def test_G2_intrinsic_transform(golden):
    g = golden("G2_intrinsic_transform")
    for S in (32, 64, 128, 256):
        for i, k in enumerate(g["K"]):
            assert np.array_equal(OG.intrinsic_transform(k, S, S), g[f"S{S}"][i])
        assert np.array_equal(OG.intrinsic_transform(g["K"], S, S), g[f"S{S}_batched"])
    assert np.allclose(OG.intrinsic_transform(g["K"][4], 64, 64)[[0, 1, 0, 1], [0, 1, 2, 2]],
                       [75.7486, 76.0456, 32.5, 32.0], atol=1e-4)   # SURVEY §8a a21 KAT
    assert np.array_equal(OG.candidate_intrinsics(), g["K"])


def test_G3_random_sample_pose(golden):
    g = golden("G3_random_sample_pose")
    for s in (0, 1, 12345):
        np.random.seed(s)
        assert np.array_equal(OG.random_sample_pose(4), g[f"pose_seed{s}"])
        assert np.array_equal(np.random.rand(2), g[f"after_seed{s}"])   # same amount of stream consumed


def test_G4_pc2depth(golden):
    g = golden("G4_pc2depth")
    d, m = OG.pc2depth_tensor(T(g["pc"]), T(g["valid"]), T(g["K"]), (64, 64))
    assert np.array_equal(d.numpy(), g["depth"]) and np.array_equal(m.numpy(), g["mask"])
    d, m = OG.pc2depth_tensor(T(g["pc"][:, :5000]), T(g["valid"][:, :5000]), T(g["K"]), (48, 80))
    assert np.array_equal(d.numpy(), g["depth_48x80"]) and np.array_equal(m.numpy(), g["mask_48x80"])


def test_G5_G6_reproject_unproject(golden):
    g = golden("G5_G6_reproject_unproject")
    depth, K, pose = T(g["depth"]), T(g["K"]), T(g["pose"])
    d, m = OG.reproject_tensor(depth * 10, K, pose, (0, 10))
    assert np.array_equal(d.numpy(), g["rpj_depth"]) and np.array_equal(m.numpy(), g["rpj_mask"])
    d, m = OG.reproject_tensor(depth * 10, K, pose, (0.5, 10))
    assert np.array_equal(d.numpy(), g["rpj05_depth"]) and np.array_equal(m.numpy(), g["rpj05_mask"])
    # identity pose: the round trip returns the clipped source depth bit-exactly (SURVEY §8c [probe])
    src = (depth[2] * 10)
    assert np.array_equal(g["rpj_depth"][2], torch.where((src > 0) & (src < 10), src, torch.zeros(())).numpy())
    pc, ok = OG.depth2pc_tensor(depth * 10, K, (0.5, 10))
    assert np.array_equal(pc.numpy(), g["pc"], equal_nan=True) and np.array_equal(ok.numpy(), g["pc_valid"])
    for b in range(3):
        p = OG.point_cloud(g["depth"][b, 0] * 10, g["K"][b], (0.5, 10))
        assert p.dtype == np.float64 and np.array_equal(p, g[f"cloud{b}"])
        assert np.array_equal(OG.inverse_pose_apply(p, g["pose"][b]), g[f"cloud{b}_common"])
        dd, mm = OG.project_cloud(g[f"cloud{b}"].astype(np.float32), g["pose"][b], g["K"][b], 64)
        assert np.array_equal(dd[0].numpy(), g[f"gen_depth{b}"]) and np.array_equal(mm[0].numpy(), g[f"gen_mask{b}"])


def test_G7_unet_small_with_taps(golden):
    g = golden("G7_unet_small_taps")
    for dim in (8, 16):
        p = W.synth_state_dict(W.unet_config(dim), 7)
        taps = {}
        y = OU.unet_forward(p, T(g[f"d{dim}_x"]), T(g[f"d{dim}_t"]), T(g[f"d{dim}_pc"]), taps=taps)
        assert np.array_equal(y.numpy(), g[f"d{dim}_y"])
        for k in ("init_conv", "down0_block0", "down0_attn", "down0_out", "mid_attn", "up0_out", "final_res"):
            assert np.array_equal(taps[k].numpy(), g[f"d{dim}_tap_{k}"]), k


def test_G8_unet_dim64(golden):
    g = golden("G8_unet_dim64")
    p = W.synth_state_dict(W.unet_config(64), 8)
    y = OU.unet_forward(p, T(g["x"]), T(g["t"]), T(g["pc"]))
    assert np.array_equal(y.numpy(), g["y"])


def _denoiser(dim, seed):
    p = W.synth_state_dict(W.unet_config(dim), seed)
    return lambda x, t, c: OU.unet_forward(p, x, t, c)


def test_G9_single_transitions(golden):
    g = golden("G9_G10_sampler")
    sch = OD.schedule(1000)
    den = _denoiser(16, 9)
    x, pc, cond = T(g["x"]), T(g["pc"]), T(g["cond"])
    for t in (999, 500, 1, 0):
        img, x0 = OD.p_sample(sch, den, x, t, pc, cond, T(g[f"ps{t}_noise"]))
        assert np.array_equal(img.numpy(), g[f"ps{t}_img"]), t
        assert np.array_equal(x0.numpy(), g[f"ps{t}_x0"]), t
    img, _ = OD.p_sample(sch, den, x, 500, pc, None, T(g["ps500_nocond_noise"]))
    assert np.array_equal(img.numpy(), g["ps500_nocond_img"])


def test_G10_short_chains(golden):
    g = golden("G9_G10_sampler")
    den = _denoiser(16, 9)
    pc, cond = T(g["pc"]), T(g["cond"])
    out = OD.p_sample_loop(OD.schedule(8), den, pc, cond, (2, 1, 32, 32), OD.stored_noise(T(g["chain8_noise"])))
    assert np.array_equal(out.numpy(), g["chain8_out"])
    sch = OD.schedule(1000)
    out = OD.sample(sch, den, pc, cond, 32, OD.stored_noise(T(g["ddim5_noise"])), sampling_steps=5)
    assert np.array_equal(out.numpy(), g["ddim5_out"])
    out = OD.sample(sch, den, pc, None, 32, OD.stored_noise(T(g["ddim5_nocond_noise"])), sampling_steps=5)
    assert np.array_equal(out.numpy(), g["ddim5_nocond_out"])
    # the reference's own draw order reproduced from a seeded torch generator
    gen = torch.Generator().manual_seed(4321)
    torch.manual_seed(4321)
    out = OD.sample(sch, den, pc, cond, 32, lambda k: torch.randn((2, 1, 32, 32)), sampling_steps=5)
    assert np.array_equal(out.numpy(), g["ddim5_out"])


def test_G11_maskunet(golden):
    g = golden("G11_maskunet")
    x = T(g["depth"])
    assert np.array_equal(OU.depth_augment(x).numpy(), g["augment"])
    for dim in (8, 16):
        p = W.synth_state_dict(W.maskunet_config(dim), 11, final_bias=4.0)
        assert np.array_equal(OU.maskunet_forward(p, x).numpy(), g[f"d{dim}_prob"])
    old = OD.MASK_THRESHOLD
    try:
        OD.MASK_THRESHOLD = float(g["thr"])
        cond, d, m = OD.correct_and_condition(T(g["d16_prob"]), x, T(g["hit"]))
    finally:
        OD.MASK_THRESHOLD = old
    assert np.array_equal(cond.numpy(), g["img_cond"]) and np.array_equal(d.numpy(), g["corrected"])
    assert np.array_equal(m.numpy(), g["mask_out"])


def test_G12_end_to_end_pair(golden):
    """Config 1 of BASELINE.json: 64x64, 50-step DDIM, dim-64 networks (CPU plumbing case)."""
    g = golden("G12_end_to_end_64")
    unet_p = W.synth_state_dict(W.unet_config(64), 12)
    mask_p = W.synth_state_dict(W.maskunet_config(64), 13, final_bias=6.0)
    d_rpj, hit = OG.reproject_tensor(T(g["depth"]) * 10, T(g["K"]), T(g["pose"]), (0, 10))
    assert np.array_equal((d_rpj * 0.1).numpy(), g["rpj_depth"]) and np.array_equal(hit.numpy(), g["rpj_mask"])
    prob1 = OU.maskunet_forward(mask_p, d_rpj * 0.1)
    assert np.array_equal(prob1.numpy(), g["prob1"])
    old = OD.MASK_THRESHOLD
    try:
        OD.MASK_THRESHOLD = float(g["thr1"])
        cond, _, _ = OD.correct_and_condition(prob1, d_rpj * 0.1, hit)
    finally:
        OD.MASK_THRESHOLD = old
    assert np.array_equal(cond.numpy(), g["img_cond"])
    den = lambda x, t, c: OU.unet_forward(unet_p, x, t, c)
    img = OD.sample(OD.schedule(1000), den, OG.param_vector(T(g["K"])), cond, 64, OD.stored_noise(T(g["noise"])),
                    sampling_steps=50)
    assert np.array_equal(img.numpy(), g["sampled"])
    prob2 = OU.maskunet_forward(mask_p, img)
    assert np.array_equal(prob2.numpy(), g["prob2"])
    out = torch.where(prob2 > float(g["thr2"]), img, torch.zeros_like(img))
    cloud = OG.inverse_pose_apply(OG.point_cloud(out[0, 0].numpy() * 10, g["K"][0], (0.5, 10)), g["pose"][0])
    assert np.array_equal(cloud, g["cloud"])


# ------------------------------------------------------------------------------------------------------------------
# Round-2 fixtures: the benchmarked configuration (dim 64, 128x128), the reference's shipped 256x256 resolution, DDIM
# with known pixels > 1, and the float64 "exact arithmetic" envelopes (tools/make_goldens.py g12b..g17)
# ------------------------------------------------------------------------------------------------------------------
TAP_NAMES = ("init_conv", "down0_block0", "down0_attn", "down0_out", "mid_attn", "up0_out", "final_res")


def _check_unet_fixture(g, wseed):
    p = W.synth_state_dict(W.unet_config(64), wseed)
    taps = {}
    y = OU.unet_forward(p, T(g["x"]), T(g["t"]), T(g["pc"]), taps=taps)
    assert np.array_equal(y.numpy(), g["y"])
    for k in TAP_NAMES:
        assert np.array_equal(taps[k].reshape(-1)[T(g[f"tap_{k}_idx"])].numpy(), g[f"tap_{k}"]), k
    # the stored float64 evaluation is the same function without roundoff: fp32 sits within a few 1e-6 of it
    assert np.abs(g["y"].astype(np.float64) - g["y64"]).max() < 2e-5


def test_G13_unet_dim64_128(golden):
    _check_unet_fixture(golden("G13_unet_dim64_128"), 13)


def test_G16_unet_dim64_256(golden):
    """One forward at the reference's shipped resolution (generate_dataset.py:34-49)."""
    _check_unet_fixture(golden("G16_unet_dim64_256"), 16)


def test_G14_chain8_dim64_128(golden):
    g = golden("G14_chain8_dim64_128")
    den = _denoiser(64, 14)
    out = OD.p_sample_loop(OD.schedule(8), den, T(g["pc"]), T(g["cond"]), (2, 1, 128, 128), OD.stored_noise(T(g["noise"])))
    assert np.array_equal(out.numpy(), g["out"])
    assert np.abs(g["out"].astype(np.float64) - g["out64"]).max() < 1e-5     # noise floor of this chain: 2.6e-6


def test_G15_maskunet_dim64_128(golden):
    g = golden("G15_maskunet_dim64_128")
    p = W.synth_state_dict(W.maskunet_config(64), 15, final_bias=6.0)
    assert np.array_equal(OU.maskunet_forward(p, T(g["depth"])).numpy(), g["prob"])


def test_G17_ddim_known_pixels_above_one(golden):
    """ddim_sample clamps only the raw network output; replaced pixels > 1 enter the state unclamped (sd:1197-1218,
    1371) — the ancestral sampler clamps after the replacement (sd:1250).  Both on the same condition."""
    g = golden("G17_ddim_cond_gt1")
    den = _denoiser(16, 9)
    pc, cond = T(g["pc"]), T(g["cond"])
    assert float(cond[:, 0].max()) > 1.0
    out = OD.sample(OD.schedule(1000), den, pc, cond, 32, OD.stored_noise(T(g["ddim5_noise"])), sampling_steps=5)
    assert np.array_equal(out.numpy(), g["ddim5_out"]) and float(out.max()) > 1.0
    out = OD.p_sample_loop(OD.schedule(8), den, pc, cond, (2, 1, 32, 32), OD.stored_noise(T(g["chain8_noise"])))
    assert np.array_equal(out.numpy(), g["chain8_out"]) and float(out.max()) <= 1.0


def test_G12b_noise_floor_of_the_parity_metric(golden):
    """The measured floor of BASELINE's metric on the G12 chain: the reference's fp32 result moves by 1.0e-4 m in XYZ
    when torch runs one thread instead of eight (another oneDNN summation order) and sits 1.4e-4 m from exact arithmetic.
    (The two stored chains take ~40 s to regenerate: tools/make_goldens.py g12b; here only their consistency.)"""
    g, e = golden("G12_end_to_end_64"), golden("G12b_envelope")
    assert abs(np.abs(e["sampled_1thread"] - g["sampled"]).max() - float(e["depth_1thread"])) < 1e-12
    assert abs(np.abs(e["sampled_exact"] - g["sampled"].astype(np.float64)).max() - float(e["depth_exact"])) < 1e-12
    assert 0.9e-4 < float(e["xyz_1thread"]) < 1.2e-4 and 1.2e-4 < float(e["xyz_exact"]) < 1.6e-4
    known = OD.cond_mask(T(g["img_cond"])).numpy()
    assert np.array_equal(e["sampled_1thread"][known], g["sampled"][known])    # DDNM pixels never move


def test_G18_refine_step_occlusion_filter_random_transform(golden):
    """What Tester.sample / Tester.generate add to the generator's path (sd:1307-1314, 1374-1388, 446-463, 377-415)."""
    g = golden("G18_refine_occlusion_transform")
    den = _denoiser(16, 9)
    pc, cond = T(g["pc"]), T(g["cond"])
    out = OD.sample(OD.schedule(1000), den, pc, cond, 32, OD.stored_noise(T(g["ddim5_refine_noise"])), sampling_steps=5,
                    has_refine_step=True)
    assert np.array_equal(out.numpy(), g["ddim5_refine_out"])
    out = OD.p_sample_loop(OD.schedule(8), den, pc, cond, (2, 1, 32, 32), OD.stored_noise(T(g["chain8_refine_noise"])),
                           has_refine_step=True)
    assert np.array_equal(out.numpy(), g["chain8_refine_out"])
    d, m = OG.occlusion_filter(T(g["of_depth_in"]), T(g["of_mask_in"]))
    assert np.array_equal(d.numpy(), g["of_depth_out"]) and np.array_equal(m.numpy(), g["of_mask_out"])
    assert (g["of_depth_out"] != g["of_depth_in"]).sum() > 10             # the filter does something on this input
    for s_ in (0, 7):
        np.random.seed(s_)
        assert np.array_equal(OG.random_sample_transform(g["rst_K"], 64), g[f"rst_seed{s_}"])
        assert np.array_equal(np.random.rand(2), g[f"rst_after_seed{s_}"])


def test_G0_host_tables_are_this_hosts():
    """The SinusoidalPosEmb frequency table of the fixtures' host (the reference's float32 torch.exp, sd:645-657): the CPU
    tests run on that host, so torch reproduces it bit for bit here; the GPU tests hand it to the library explicitly."""
    import math
    g = dict(np.load(os.path.join(GOLDEN, "G0_host_tables.npz")))
    for dim in (8, 16, 64):
        half = dim // 2
        f = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
        assert np.array_equal(f.numpy(), g[f"freqs_dim{dim}"])
        assert np.array_equal(OU.sinusoidal(torch.tensor([999.0]), dim)[0, :half].numpy(), (999.0 * f).sin().numpy())


def test_G19_G20_long_chain_prefix_and_clouds(golden):
    """Round-3 chains of real length (1000-step ancestral at 64x64, 250-step DDIM at 128x128 and at 256x256, calibrated denoiser).
    tools/make_goldens.py asserted the oracle bit-exact over the WHOLE chain when the fixture was made (minutes of CPU);
    here: the regenerated noise matches its sha256, the oracle reproduces the reference's state after 20 transitions
    bit-exactly, and the stored cloud is the oracle's unprojection of the stored image."""
    from conftest import LONG_CHAINS, regenerate_chain_noise
    sch = OD.schedule(1000)
    for name, c in LONG_CHAINS.items():
        g = golden(name)
        nz = regenerate_chain_noise(g)
        sd = W.synth_state_dict(W.unet_config(64), int(g["wseed"]), calibrated=True)
        den = lambda x, t, cc: OU.unet_forward(sd, x, t, cc)
        if c["S"] <= 128:       # (256x256: 2.4 s per oracle evaluation — the whole-chain assert of make_goldens.py stands alone)
            x20 = OD.sample(sch, den, T(g["pc"]), T(g["img_cond"]), c["S"], OD.stored_noise(nz), sampling_steps=c["steps"],
                            stop_after=20)
            assert np.array_equal(x20.numpy(), g["x20"]), name
        cloud = OG.inverse_pose_apply(OG.point_cloud(g["sampled"][0, 0] * 10, g["K"][0], (0.5, 10.0)), g["pose"][0])
        assert np.array_equal(cloud, g["cloud"])
        # the fixture is a non-degenerate workload: the in-painted pixels do not sit on the clamp
        assert float(g["saturated_fraction_inpainted"]) < 0.2 and float(g["inpainted_fraction"]) > 0.1
        # known pixels are returned bit-exactly (DDNM replacement at the last transition)
        known = OD.cond_mask(T(g["img_cond"])).numpy()
        assert np.array_equal(g["sampled"][known], ((g["img_cond"][:, 0:1] + 1) * 0.5)[known])
