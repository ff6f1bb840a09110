"""GPU parity tests: the HIP hot path, called through the C-ABI (ctypes), against (a) the golden vectors the real
reference produced and (b) the CPU oracle on the same seeded inputs.  Run with `-m gpu` on an MI355X.

Tolerances (stated per assert):
  * geometry / z-buffer / masks / DDNM known pixels: bit-exact;
  * fp32 mode (exact-f32 MFMA): |err| <= 1e-4 on O(1..10) activations — accumulation-order roundoff only
    (observed 3e-6 .. 1.4e-5);
  * bf16 mode: |err| <= 0.125 max, <= 0.022 mean on O(1..10) activations (observed 0.062 / 0.0104 at the benchmarked
    size): bf16 storage of ~100 chained layers; bounds are <= 2x what is observed, and the drift is printed.
"""
import numpy as np
import pytest
import torch

from pointreggpt_amd import weights as W

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4
BF16_MAX, BF16_MEAN = 0.125, 0.022       # <= 2x observed (dim 64: max 0.062 / mean 0.0104 at 128x128, 0.056 / 0.0094 at 256x256)
MXFP8_MAX, MXFP8_MEAN = 0.95, 0.17          # MX-fp8 3x3 convs (3-bit mantissas): <= 2x observed (0.47 / 0.085 on O(5) outputs, 8x bf16)
NORTH_STAR = 1e-5          # point-XYZ L-infinity 1e-4 m == 1e-5 in normalised depth (1.0 == 10 m)
XYZ_FLOOR_FACTOR = 1.0
BF16_CHAIN_MAX, BF16_CHAIN_MEAN = 0.08, 0.004    # few-transition chains, in-painted pixels, normalised depth: <= 2x observed (0.038 / 0.002)


@pytest.fixture(scope="module")
def hip():
    from pointreggpt_amd import _lib, geometry
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import MaskUnet, Unet
    _lib.load()

    class NS:
        pass

    ns = NS()
    ns.G, ns.GaussianDiffusion, ns.MaskUnet, ns.Unet, ns.lib = geometry, GaussianDiffusion, MaskUnet, Unet, _lib
    return ns


def D(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def golden_unet(hip, golden, dim, dtype, sd):
    """A U-Net for comparisons with the golden fixtures: the SinusoidalPosEmb frequency table is the one of the host the
    fixtures were generated on (G0_host_tables; the reference's outputs depend on that float32 exp, see include/prg.h)."""
    net = hip.Unet(dim, dtype=dtype).load_state_dict(sd)
    return net.set_time_freqs(golden("G0_host_tables")[f"freqs_dim{dim}"])


def maxerr(a, b):
    return float(np.nanmax(np.abs(a.detach().cpu().double().numpy() - np.asarray(b, dtype=np.float64))))


def meanerr(a, b):
    return float(np.nanmean(np.abs(a.detach().cpu().double().numpy() - np.asarray(b, dtype=np.float64))))


def golden_ddim(hip, golden, net, S, steps):
    """GaussianDiffusion(timesteps=1000, sampling_timesteps=steps) whose transition coefficients are those of the host the
    fixtures were generated on: sigma and c = sqrt(1 - a' - sigma^2) (sd:1357-1368) are float32 expressions with
    cancellation that two x86 hosts evaluate 8.7e-6 apart at the first transition (G0_host_tables, tools/_dump_chain.py)."""
    d = hip.GaussianDiffusion(net, image_size=S, timesteps=1000, sampling_timesteps=steps)
    g0 = golden("G0_host_tables")
    rows = d.step_table()
    tab, tt = g0[f"ddim{steps}_rows"], g0[f"ddim{steps}_t"]
    assert [r["t"] for r in rows] == tt.tolist()
    worst = 0.0
    for r, v in zip(rows, tab):
        for j, k in enumerate(("c_x0", "c_x", "c_eps", "sigma", "sqrt_recip", "sqrt_recipm1")):
            if v[j] != 0:
                worst = max(worst, abs(r[k] - float(v[j])) / abs(float(v[j])))
            r[k] = float(v[j])
    print(f"DDIM-{steps} coefficients: this host vs the fixtures' host differ by up to {worst:.2e} (relative)")
    d.step_table = lambda: rows
    return d


# ------------------------------------------------------------------------------------------------------------------
# geometry: bit-exact
# ------------------------------------------------------------------------------------------------------------------
def test_zbuffer_bit_exact(hip, golden):
    g = golden("G4_pc2depth")
    d, m = hip.G.pc2depth_tensor(D(g["pc"]), D(g["valid"]), D(g["K"]), image_size=(64, 64))
    assert np.array_equal(d.cpu().numpy(), g["depth"]) and np.array_equal(m.cpu().numpy(), g["mask"])
    d, m = hip.G.pc2depth_tensor(D(g["pc"][:, :5000]), D(g["valid"][:, :5000]), D(g["K"]), image_size=(48, 80))
    assert np.array_equal(d.cpu().numpy(), g["depth_48x80"]) and np.array_equal(m.cpu().numpy(), g["mask_48x80"])
    # empty cloud and all-invalid cloud: every pixel 0 / False
    d, m = hip.G.pc2depth_tensor(torch.zeros((2, 0, 3), device="cuda"), None, D(g["K"]), image_size=(16, 16))
    assert float(d.abs().sum()) == 0 and not bool(m.any())
    d, m = hip.G.pc2depth_tensor(D(g["pc"]), torch.zeros((2, 20000), dtype=torch.bool, device="cuda"), D(g["K"]),
                                 image_size=(64, 64))
    assert float(d.abs().sum()) == 0 and not bool(m.any())


def test_reproject_unproject_bit_exact(hip, golden):
    g = golden("G5_G6_reproject_unproject")
    depth, K, pose = D(g["depth"]), D(g["K"]), D(g["pose"])
    d, m = hip.G.reproject_tensor(depth, K, pose, clip=(0, 10), depth_unit=10.0)
    assert np.array_equal(d.cpu().numpy(), g["rpj_depth"]) and np.array_equal(m.cpu().numpy(), g["rpj_mask"])
    d, m = hip.G.reproject_tensor(depth, K, pose, clip=(0.5, 10), depth_unit=10.0)
    assert np.array_equal(d.cpu().numpy(), g["rpj05_depth"]) and np.array_equal(m.cpu().numpy(), g["rpj05_mask"])
    pc, ok = hip.G.depth2pc_tensor(depth * 10, K, clip=(0.5, 10))
    assert np.array_equal(pc.cpu().numpy(), g["pc"], equal_nan=True) and np.array_equal(ok.cpu().numpy(), g["pc_valid"])
    pc, ok = hip.G.depth2pc_tensor(depth * 10, K, clip=(0, 10), invalid_num=0.0)
    assert np.array_equal(pc.cpu().numpy(), g["pc0"]) and np.array_equal(ok.cpu().numpy(), g["pc0_valid"])
    cam = hip.G.point_clouds(depth, K, None)
    com = hip.G.point_clouds(depth, K, pose)
    for b in range(3):
        assert cam[b].dtype == np.float64 and np.array_equal(cam[b], g[f"cloud{b}"])
        assert np.array_equal(com[b], g[f"cloud{b}_common"])       # the parity metric's quantity: 0 m
    d, m = hip.G.project_clouds([g[f"cloud{b}"].astype(np.float32) for b in range(3)], g["pose"], g["K"], 64, "cuda")
    for b in range(3):
        assert np.array_equal(d[b].cpu().numpy(), g[f"gen_depth{b}"]) and np.array_equal(m[b].cpu().numpy(), g[f"gen_mask{b}"])


def test_augment_and_mask_bit_exact(hip, golden):
    g = golden("G11_maskunet")
    assert np.array_equal(hip.G.depth_augment(D(g["depth"])).cpu().numpy(), g["augment"])
    dd, hh, cond = hip.G.apply_mask(D(g["d16_prob"]), D(g["depth"]), D(g["hit"]), float(g["thr"]))
    assert np.array_equal(dd.cpu().numpy(), g["corrected"]) and np.array_equal(hh.cpu().numpy(), g["mask_out"])
    assert np.array_equal(cond.cpu().numpy(), g["img_cond"])


def test_depth_augment_vector_and_scalar_forms_against_oracle(hip):
    """Round 6: DepthAugment (dc:577-604) runs four pixels per thread when W % 4 == 0 and one pixel per thread otherwise; both
    forms against the oracle's max-pool restatement, bit for bit, on images with holes, all-zero windows, borders and a
    constant-zero image (the `no valid neighbour` branch)."""
    from oracle import unet as OU
    g = torch.Generator().manual_seed(4)
    for (B, H, W) in [(3, 32, 32), (2, 30, 30), (1, 8, 4), (2, 5, 7), (1, 128, 128)]:
        d = torch.rand((B, 1, H, W), generator=g) * 3
        d[torch.rand((B, 1, H, W), generator=g) < 0.4] = 0.0
        d[0, 0, : H // 2, : W // 2] = 0.0                 # a hole larger than the window
        if B > 1:
            d[1] = 0.0
        got = hip.G.depth_augment(d.cuda()).cpu()
        assert torch.equal(got, OU.depth_augment(d)), (B, H, W)


def test_geometry_properties_full_size(hip):
    """BASELINE size (B=64, 128x128): identity pose round-trips the clipped depth bit-exactly; the z-buffer keeps
    the minimum; project(unproject(d)) with any pose never invents depth outside the source range."""
    from pointreggpt_amd import synthetic
    B, S = 64, 128
    depth, K, pose = synthetic.synth_batch(0, range(B), S)
    d, Kd = D(depth), D(K)
    eye = torch.eye(4, device="cuda").repeat(B, 1, 1)
    r, m = hip.G.reproject_tensor(d, Kd, eye, clip=(0, 10), depth_unit=10.0, out_scale=1.0)
    src = d * 10
    assert torch.equal(r, torch.where((src > 0) & (src < 10), src, torch.zeros_like(src)))
    assert torch.equal(m, (src > 0) & (src < 10))
    r, m = hip.G.reproject_tensor(d, Kd, D(pose), clip=(0, 10), depth_unit=10.0, out_scale=0.1)
    assert bool((r[m] > 0).all()) and float(r[~m].abs().sum()) == 0
    # two copies of a cloud, the second pushed 1.5x farther along its rays: the z-buffer equals the first alone
    pc, ok = hip.G.depth2pc_tensor(src, Kd, clip=(0.5, 10))
    d1, m1 = hip.G.pc2depth_tensor(pc, ok, Kd, image_size=(S, S))
    d2, m2 = hip.G.pc2depth_tensor(torch.cat([pc * 1.5, pc], 1), torch.cat([ok, ok], 1), Kd, image_size=(S, S))
    assert torch.equal(d1[m1], d2[m1]) and bool((m2 | ~m1).all())


# ------------------------------------------------------------------------------------------------------------------
# U-Nets
# ------------------------------------------------------------------------------------------------------------------
TAPS = ("init_conv", "down0_block0", "down0_attn", "down0_out", "mid_attn", "up0_out", "final_res")


@pytest.mark.parametrize("dim", [8, 16])
def test_unet_small_taps_fp32(hip, golden, dim):
    g = golden("G7_unet_small_taps")
    net = golden_unet(hip, golden, dim, "fp32", W.synth_state_dict(W.unet_config(dim), 7))
    net.set_taps(True)
    y = net(D(g[f"d{dim}_x"]), D(g[f"d{dim}_t"]), D(g[f"d{dim}_pc"]))
    for k in TAPS:
        assert maxerr(net.get_tap(k, 2), g[f"d{dim}_tap_{k}"]) <= FP32_TOL, k
    assert maxerr(y, g[f"d{dim}_y"]) <= FP32_TOL
    assert maxerr(net.get_tap("init_conv", 2), g[f"d{dim}_tap_init_conv"]) == 0.0   # same fmaf chain order


@pytest.mark.parametrize("dim", [8, 16])
def test_unet_small_bf16(hip, golden, dim):
    g = golden("G7_unet_small_taps")
    net = golden_unet(hip, golden, dim, "bf16", W.synth_state_dict(W.unet_config(dim), 7))
    y = net(D(g[f"d{dim}_x"]), D(g[f"d{dim}_t"]), D(g[f"d{dim}_pc"]))
    print(f"dim {dim} bf16: max {maxerr(y, g[f'd{dim}_y']):.3e} mean {meanerr(y, g[f'd{dim}_y']):.3e}")
    assert maxerr(y, g[f"d{dim}_y"]) <= BF16_MAX and meanerr(y, g[f"d{dim}_y"]) <= BF16_MEAN


def test_unet_dim64(hip, golden):
    g = golden("G8_unet_dim64")
    sd = W.synth_state_dict(W.unet_config(64), 8)
    y = golden_unet(hip, golden, 64, "fp32", sd)(D(g["x"]), D(g["t"]), D(g["pc"]))
    assert maxerr(y, g["y"]) <= FP32_TOL
    y = golden_unet(hip, golden, 64, "bf16", sd)(D(g["x"]), D(g["t"]), D(g["pc"]))
    print(f"dim 64 @64 bf16: max {maxerr(y, g['y']):.3e} mean {meanerr(y, g['y']):.3e}")
    assert maxerr(y, g["y"]) <= BF16_MAX and meanerr(y, g["y"]) <= BF16_MEAN


def test_time_frequency_table_defaults(hip, golden):
    """set_time_freqs: the default (torch on this host's CPU = the reference's CPU path here), 'device' (torch on the HIP
    device = the reference on an accelerator) and the fixtures' host's table; the default path reproduces G8 in fp32."""
    g, g0 = golden("G8_unet_dim64"), golden("G0_host_tables")
    sd = W.synth_state_dict(W.unet_config(64), 8)
    net = hip.Unet(64, dtype="fp32").load_state_dict(sd)               # default table
    cpu_tab = net._freqs.copy()
    y = net(D(g["x"]), D(g["t"]), D(g["pc"]))
    dev_tab = net.set_time_freqs("device")._freqs.copy()
    ulp = lambda a, b: int(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)).max())
    print(f"time frequencies: this host's CPU vs the fixtures' host {ulp(cpu_tab, g0['freqs_dim64'])} ulp, "
          f"HIP device vs the fixtures' host {ulp(dev_tab, g0['freqs_dim64'])} ulp")
    assert cpu_tab.shape == (32,) and dev_tab.shape == (32,) and cpu_tab[0] == 1.0 and dev_tab[0] == 1.0
    assert ulp(cpu_tab, g0["freqs_dim64"]) <= 1 and ulp(dev_tab, g0["freqs_dim64"]) <= 2
    assert maxerr(y, g["y"]) <= FP32_TOL                                   # t <= 999: one ulp of a frequency is <= 6e-5 in an embedding
    with pytest.raises(ValueError):
        net.set_time_freqs("gpu")


_FASTPATH_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from pointreggpt_amd import weights as W
from pointreggpt_amd.unet import Unet
g = np.load({gold!r})
sd = W.synth_state_dict(W.unet_config(64), 8)
net = Unet(64, dtype="bf16").load_state_dict(sd)
D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
y64 = net(D(g["x"]), D(g["t"]), D(g["pc"])).float().cpu().numpy()
gen = torch.Generator().manual_seed(5)
x = torch.randn((2, 1, 128, 128), generator=gen)
y128 = net(x.cuda(), torch.tensor([3, 900]).cuda(), D(g["pc"][:1]).repeat(2, 1)).float().cpu().numpy()
x40 = torch.randn((3, 1, 40, 40), generator=gen)     # 1600 / 400 pixels: partial 64-pixel attention tiles, ragged conv tiles
y40 = net(x40.cuda(), torch.tensor([0, 500, 999]).cuda(), D(g["pc"][:1]).repeat(3, 1)).float().cpu().numpy()
x96 = torch.randn((2, 1, 96, 96), generator=gen)     # 3 x 12 tiles of 8x32 at 96, 3 x 3 tiles of 16x16 at 48: tile counts that are not powers of two
y96 = net(x96.cuda(), torch.tensor([7, 640]).cuda(), D(g["pc"][:1]).repeat(2, 1)).float().cpu().numpy()
np.savez({out!r}, y64=y64, y128=y128, y40=y40, y96=y96)
"""


_QK_EDGE_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from pointreggpt_amd import weights as W
from pointreggpt_amd.unet import Unet
sd = W.synth_state_dict(W.unet_config(64), 8)
worst = 0.0
for pre in [k[:-len(".fn.fn.to_qkv.weight")] for k in sd if k.endswith(".fn.fn.to_qkv.weight") and not k.startswith("mid_attn")]:
    w = sd[pre + ".fn.fn.to_qkv.weight"]
    g = sd[pre + ".fn.norm.g"].reshape(-1).double()
    C = w.shape[1]
    wg = w.reshape(384, C).double() * g[None, :] * 1.4426950408889634
    bound = float((1.02 * torch.sqrt((wg[:256] ** 2).sum(1) * C)).max())       # what unet.hip computes (log2 units)
    f = {target} / bound
    w[:256] *= f                                                                # q and k rows: the static bound becomes `target`
    worst = max(worst, bound * f)
net = Unet(64, dtype="bf16").load_state_dict(sd)
gen = torch.Generator().manual_seed(11)
x = torch.randn((2, 1, 64, 64), generator=gen)
pc = torch.tensor([[75.7, 76.0, 32.5, 32.0]] * 2)
y = net(x.cuda(), torch.tensor([5, 800]).cuda(), pc.cuda()).float().cpu().numpy()
np.savez({out!r}, y=y, bound=np.float64(worst))
"""


def test_linear_attention_without_shift_at_the_edge_of_the_static_bound(tmp_path):
    """Round 3 dropped the softmax shift where the static bound |q|, |k| <= 57.7 (log2 units) holds.  Here the q and k rows
    of every linear-attention block are scaled until that bound is 56 — the static path is still taken, exp2 runs on
    arguments an order of magnitude larger than with the synthetic weights — and the result must stay finite and agree
    with (a) the measured-maximum path (PRG_LA_KSHIFT=0: la_kmax pass + shift), (b) the unfused kernels (their own
    shifted softmax).  At 59 the bound fails and the library itself falls back to (a): same comparison."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for target in (56.0, 59.0):
        outs = {}
        for name, env in {"static": {}, "measured": {"PRG_LA_KSHIFT": "0"}, "unfused": {"PRG_FUSED_ATTN": "0"}}.items():
            out = str(tmp_path / f"{name}_{target}.npz")
            r = subprocess.run([sys.executable, "-c", _QK_EDGE_SCRIPT.format(root=root, out=out, target=target)],
                               env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs[name] = np.load(out)
        assert abs(float(outs["static"]["bound"]) - target) < 1e-6
        assert np.isfinite(outs["static"]["y"]).all()
        for name in ("measured", "unfused"):
            d = np.abs(outs["static"]["y"].astype(np.float64) - outs[name]["y"].astype(np.float64))
            print(f"bound {target}: static vs {name}: max {d.max():.3e} mean {d.mean():.3e}")
            assert d.max() <= 0.12 and d.mean() <= 0.016, (target, name, d.max(), d.mean())   # the fast-path variants' bound


def test_bf16_fast_paths_match_generic_kernels(tmp_path):
    """The wave-specialised 3x3 conv and the fused linear attention are alternative schedules of the same arithmetic:
    switching either off (generic implicit-GEMM conv / unfused LayerNorm-qkv-attention kernels) must give the same
    U-Net output up to bf16 rounding of intermediates.  64x64 exercises tiles 8x32x64 / 4x32x128 / 8x16x128 at widths
    64 / 32 / 16, 128x128 the full-size ones; the attention blocks run at C = 64 and 128 (fused) and 256, 512 (unfused)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden", "G8_unet_dim64.npz")
    outs = {}
    variants = {"fast": {}, "no_ws": {"PRG_CONV_WS": "0"}, "no_fused_attn": {"PRG_FUSED_ATTN": "0"},
                "no_kshift": {"PRG_LA_KSHIFT": "0"},       # measured column maxima instead of the static softmax shift
                "no_c64": {"PRG_CONV_C64": "0"},           # 64 -> 64 convs through the wave-specialised kernel instead
                "gn_fold": {"PRG_GN_FOLD": "1", "PRG_GN_ACC": "0"},           # GroupNorm coefficients folded inside the c64 conv (per-image ticket)
                "c64_contiguous": {"PRG_C64_INTERLEAVE": "0"},   # contiguous instead of interleaved tile runs
                # 256-pixel x 128-channel tiles wherever the shape allows (at these batch sizes the default dispatch keeps
                # the 128-pixel tiles): fused prologue, x2 gather, two sources, statistics, 8x32 and 16x16 tiles
                "w256_all": {"PRG_W256_MIN_TILES": "1"}, "no_w256": {"PRG_CONV_W256": "0", "PRG_CONV_DOWN_W256": "0"},
                # ResnetBlock tail of up levels 0-1 as a separate pass instead of the res_conv's epilogue
                "no_res_epilogue": {"PRG_RES_EPILOGUE": "0"},
                "no_head_fuse": {"PRG_HEAD_FUSE": "0"},            # the 1x1 head as its own launch instead of the final tail's epilogue
                # GroupNorm statistics as per-tile slabs + gn_coeff launches (rounds 1-2) instead of the fixed-point accumulators
                # folded by the consumers (round 3)
                "no_gn_acc": {"PRG_GN_ACC": "0"},
                # sum_n p of la_ctx on the matrix pipe at every width / as float additions at every width
                "la_psum": {"PRG_LA_PSUM": "1"}, "la_ssum": {"PRG_LA_PSUM": "0"},
                # ... and the accumulators with every eligible shape on the 256-pixel kernel (its in-kernel fold at every width)
                "gn_acc_w256_all": {"PRG_W256_MIN_TILES": "1", "PRG_GN_ACC": "1"},
                # round 4: h1 (the tensor between a ResnetBlock's two convs) as bf16 with the float32 prologue instead of f16 with
                # the packed-f16 prologue and f16 MFMA operands in conv2 (conv.h, "h16")
                "no_h16": {"PRG_H16": "0"},
                # Upsample convs in the nine-tap gather form instead of the four 2 x 2-tap sub-pixel convolutions (conv_w256.hip MODE 2);
                # and the sub-pixel form wherever the shape allows (at these batch sizes the default keeps the gather form)
                "no_up2x2": {"PRG_UP2X2": "0"}, "up2x2_all": {"PRG_W256_MIN_TILES": "1", "PRG_UP2X2": "1"}}
    for name, env in variants.items():
        out = str(tmp_path / f"{name}.npz")
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", _FASTPATH_SCRIPT.format(root=root, gold=gold, out=out)], env=e,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    for name in ("no_ws", "no_fused_attn", "no_kshift", "no_c64", "gn_fold", "c64_contiguous", "w256_all", "no_w256", "no_res_epilogue", "no_head_fuse",
                 "no_gn_acc", "gn_acc_w256_all", "la_psum", "la_ssum", "no_h16", "no_up2x2", "up2x2_all"):
        for k in ("y64", "y128", "y40", "y96"):
            d = np.abs(outs["fast"][k].astype(np.float64) - outs[name][k].astype(np.float64))
            assert np.isfinite(outs[name][k]).all()
            # bf16 re-rounding of ~100 chained layers; observed max 0.06 / mean 0.008 on O(7) activations
            print(f"fast vs {name} [{k}]: max {d.max():.3e} mean {d.mean():.3e}")
            assert d.max() <= 0.12 and d.mean() <= 0.016, (name, k, d.max(), d.mean())


def test_unet_vs_oracle_odd_batch_and_size(hip):
    """Ragged shapes the tiles do not divide: B=3, 48x48 (M = 6912, bottom level 6x6), per-image timesteps."""
    from oracle import unet as OU
    sd = W.synth_state_dict(W.unet_config(16), 3)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((3, 1, 48, 48), generator=g)
    t = torch.tensor([0, 417, 999])
    pc = torch.tensor([[56.8, 57.0, 24.4, 24.0]] * 3) + torch.randn((3, 4), generator=g)
    ref = OU.unet_forward(sd, x, t, pc)
    net = hip.Unet(16, dtype="fp32").load_state_dict(sd)
    y = net(x.cuda(), t.cuda(), pc.cuda())
    assert maxerr(y, ref.numpy()) <= FP32_TOL
    # B = 67 at 32x32: more batch rows than one 64-row block of the conditioning Linear kernel, every image its own timestep
    B = 67
    x = torch.randn((B, 1, 32, 32), generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    pc = torch.tensor([[37.9, 38.0, 16.2, 16.0]] * B) + torch.randn((B, 4), generator=g)
    ref = OU.unet_forward(sd, x, t, pc)
    y = net(x.cuda(), t.cuda(), pc.cuda())
    assert maxerr(y, ref.numpy()) <= FP32_TOL
    for b in (0, 63, 64, 66):                                # batch-slot invariance across the row-block boundary
        yb = net(x[b:b + 1].cuda(), t[b:b + 1].cuda(), pc[b:b + 1].cuda())
        assert torch.equal(yb.cpu(), y[b:b + 1].cpu()), b


@pytest.mark.parametrize("dim", [8, 16])
def test_maskunet(hip, golden, dim):
    g = golden("G11_maskunet")
    sd = W.synth_state_dict(W.maskunet_config(dim), 11, final_bias=4.0)
    p = hip.MaskUnet(dim, dtype="fp32").load_state_dict(sd)(D(g["depth"]))
    assert maxerr(p, g[f"d{dim}_prob"]) <= 1e-5           # probabilities: observed 6e-7
    p = hip.MaskUnet(dim, dtype="bf16").load_state_dict(sd)(D(g["depth"]))
    e = maxerr(p, g[f"d{dim}_prob"])
    print(f"MaskUnet dim {dim} bf16: {e:.3e}")
    assert e <= 0.01                                           # observed 4e-3


def test_bad_arguments_fail_loudly(hip):
    net = hip.Unet(16, dtype="fp32")
    with pytest.raises(hip.lib.PrgError):
        net(torch.zeros((1, 1, 32, 32), device="cuda"), torch.zeros(1, dtype=torch.long, device="cuda"),
            torch.zeros((1, 4), device="cuda"))                       # no weights loaded
    sd = W.synth_state_dict(W.unet_config(16), 0)
    del sd["mid_attn.fn.norm.g"]
    with pytest.raises(KeyError):
        net.load_state_dict(sd)
    net.init_synthetic(0)
    with pytest.raises(hip.lib.PrgError):                              # 20x20 cannot be halved three times to >= 2
        net(torch.zeros((1, 1, 20, 20), device="cuda"), torch.zeros(1, dtype=torch.long, device="cuda"),
            torch.zeros((1, 4), device="cuda"))
    with pytest.raises(hip.lib.PrgError):
        hip.G.pc2depth_tensor(torch.zeros((1, 4, 3)), None, torch.eye(3)[None], image_size=(8, 8))   # CPU tensors


# ------------------------------------------------------------------------------------------------------------------
# sampler
# ------------------------------------------------------------------------------------------------------------------
def _one_step_diffusion(hip, net, S, row_index, T=1000):
    d = hip.GaussianDiffusion(net, image_size=S, timesteps=T)
    rows = [d.step_table()[row_index]]
    d.step_table = lambda: rows
    return d


def test_single_transitions_fp32(hip, golden):
    """p_sample at t in {999, 500, 1, 0} from a supplied x (G9): x' compared after the (x+1)/2 output map."""
    g = golden("G9_G10_sampler")
    net = golden_unet(hip, golden, 16, "fp32", W.synth_state_dict(W.unet_config(16), 9))
    for t in (999, 500, 1, 0):
        d = _one_step_diffusion(hip, net, 32, 999 - t)
        noise = torch.from_numpy(np.stack([g["x"], g[f"ps{t}_noise"]]))
        out = d.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=noise.cuda())
        assert maxerr(out, (g[f"ps{t}_img"] + 1) * 0.5) <= FP32_TOL, t
        d.close()
    d = _one_step_diffusion(hip, net, 32, 499)
    out = d.sample(param_cond=D(g["pc"]), img_cond=None,
                   noise=torch.from_numpy(np.stack([g["x"], g["ps500_nocond_noise"]])).cuda())
    assert maxerr(out, (g["ps500_nocond_img"] + 1) * 0.5) <= FP32_TOL


def test_sampler_step_alone_against_oracle_p_sample(hip):
    """prg_debug_sampler_step = the transition update alone (sd:1257-1281 after model_predictions): the t = 0 ancestral row (no
    noise drawn) on a ragged-size batch equals the oracle's p_sample fed the same network output, bit for bit; a noisy row at a
    streaming size (B = 256 @128x128) keeps the DDNM contract (known pixels pulled towards the condition by exactly c_x0 * cond +
    c_x * x + sigma * n with |n| finite) and never writes outside x."""
    import ctypes as C
    from oracle import diffusion as OD
    from pointreggpt_amd import _lib
    lib = _lib.load()
    B, S = 3, 12
    g = torch.Generator().manual_seed(7)
    x = torch.randn((B, 1, S, S), generator=g)
    u = torch.randn((B, 1, S, S), generator=g) * 1.5
    cond = torch.cat([torch.rand((B, 1, S, S), generator=g) * 2 - 1, (torch.rand((B, 1, S, S), generator=g) > 0.5).float() * 2 - 1], 1)
    sch = OD.schedule(1000)
    ref, _ = OD.p_sample(sch, lambda x_, t_, c_: u, x, 0, None, cond, None)
    net = hip.Unet(8, dtype="fp32").init_synthetic(seed=1)
    row = hip.GaussianDiffusion(net, image_size=S, timesteps=1000).step_table()[-1]
    net.close()
    assert row["t"] == 0 and row["sigma"] == 0.0
    rc = _lib.StepC(row["t"], row["clip_pred"], row["c_x0"], row["c_x"], row["c_eps"], row["sigma"], row["sqrt_recip"], row["sqrt_recipm1"])
    xd, ud, cd = x.cuda().contiguous(), u.cuda().contiguous(), cond.cuda().contiguous()
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device="cuda")
    us = C.c_float()
    _lib.check(lib.prg_debug_sampler_step(_lib.ptr(xd), _lib.ptr(ud), _lib.ptr(cd), _lib.ptr(seeds), C.byref(rc), B, S * S, 1, C.byref(us), None))
    assert torch.equal(xd.cpu(), ref), float((xd.cpu() - ref).abs().max())
    # streaming size, noisy row: x' - (c_x0 * x0 + c_x * x) = sigma * n on every pixel, n ~ N(0, 1) from the on-chip Philox
    B, S = 256, 128
    x = torch.randn((B, S * S), device="cuda")
    u = torch.randn((B, S * S), device="cuda")
    cond = torch.cat([torch.rand((B, 1, S * S), device="cuda") * 2 - 1, (torch.rand((B, 1, S * S), device="cuda") > 0.5).float() * 2 - 1], 1).contiguous()
    guard = torch.full((B, S * S), 7.0, device="cuda")
    buf = torch.cat([guard[:1], x, guard[:1]]).contiguous()
    xin = buf[1:B + 1]
    x0 = torch.where(cond[:, 1] > 0, cond[:, 0], u).clamp(-1, 1)
    rc = _lib.StepC(500, 2, 0.25, 0.75, 0.0, 0.5, 1.2, 0.7)
    mean = 0.25 * x0 + 0.75 * xin.clone()
    seeds = torch.arange(1, B + 1, dtype=torch.int64, device="cuda")
    _lib.check(lib.prg_debug_sampler_step(_lib.ptr(xin), _lib.ptr(u), _lib.ptr(cond), _lib.ptr(seeds), C.byref(rc), B, S * S, 1, C.byref(us), None))
    n = (xin - mean) / 0.5
    assert bool(torch.isfinite(n).all()) and abs(float(n.mean())) < 2e-3 and abs(float(n.std()) - 1) < 2e-3 and float(n.abs().max()) < 7
    assert bool((buf[0] == 7.0).all()) and bool((buf[-1] == 7.0).all())
    assert not torch.equal(n[0], n[1])                    # per-scene keys


@pytest.mark.parametrize("graph", [False, True])
def test_short_chains_fp32(hip, golden, graph):
    g = golden("G9_G10_sampler")
    net = golden_unet(hip, golden, 16, "fp32", W.synth_state_dict(W.unet_config(16), 9))
    known = (g["cond"][:, 1:2] + 1) * 0.5 > 0.5
    d8 = hip.GaussianDiffusion(net, image_size=32, timesteps=8)
    out = d8.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["chain8_noise"]), use_graph=graph)
    assert maxerr(out, g["chain8_out"]) <= FP32_TOL
    assert np.array_equal(out.cpu().numpy()[known], g["chain8_out"][known])      # DDNM known pixels: exact
    d5 = golden_ddim(hip, golden, net, 32, 5)
    out = d5.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["ddim5_noise"]), use_graph=graph)
    assert maxerr(out, g["ddim5_out"]) <= FP32_TOL
    assert np.array_equal(out.cpu().numpy()[known], g["ddim5_out"][known])
    out = d5.sample(param_cond=D(g["pc"]), img_cond=None, noise=D(g["ddim5_nocond_noise"]), use_graph=graph)
    assert maxerr(out, g["ddim5_nocond_out"]) <= FP32_TOL
    with pytest.raises((AssertionError, hip.lib.PrgError)):     # too few stored draws for the table
        d5.sample(param_cond=D(g["pc"]), img_cond=None, noise=D(g["ddim5_nocond_noise"][:2]), use_graph=graph)


def test_short_chains_bf16_drift_reported(hip, golden):
    g = golden("G9_G10_sampler")
    net = golden_unet(hip, golden, 16, "bf16", W.synth_state_dict(W.unet_config(16), 9))
    known = (g["cond"][:, 1:2] + 1) * 0.5 > 0.5
    d5 = golden_ddim(hip, golden, net, 32, 5)
    out = d5.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["ddim5_noise"]))
    assert np.array_equal(out.cpu().numpy()[known], g["ddim5_out"][known])       # exact even in bf16
    e = maxerr(out, g["ddim5_out"])
    print(f"bf16 5-step DDIM drift on in-painted pixels: max {e:.3e} (normalised depth; x10 for metres)")
    assert e <= 0.07                                           # <= 2x observed (3.3e-2)


def test_philox_noise_is_shard_invariant(hip):
    """A scene's result depends on (seed key, scene inputs) only: not on its batch slot or batch composition."""
    net = hip.Unet(16, dtype="fp32").init_synthetic(4)
    d = hip.GaussianDiffusion(net, image_size=32, timesteps=1000, sampling_timesteps=4)
    pc = torch.tensor([[37.9, 38.0, 16.25, 16.0], [36.5, 36.7, 16.25, 16.0], [40.0, 40.0, 16.0, 16.0]], device="cuda")
    a = d.sample(param_cond=pc, seeds=[101, 202, 303])
    b = d.sample(param_cond=pc.flip(0).contiguous(), seeds=[303, 202, 101])
    assert torch.equal(a, b.flip(0))
    c = d.sample(param_cond=pc[1:2].contiguous(), seeds=[202])
    assert maxerr(c[0], a[1].cpu().numpy()) <= 1e-5       # other tile shape -> same math, roundoff only
    assert not torch.equal(a[0], a[1])
    # the start image is a standard normal field
    d1 = hip.GaussianDiffusion(net, image_size=32, timesteps=1000, sampling_timesteps=1)
    # (sigma = 0 on the single transition -> output is x0 only; check the Philox field through a 2-step table)
    rows = d.step_table()[:1]
    rows[0].update(c_x0=0.0, c_x=1.0, c_eps=0.0, sigma=0.0)      # identity transition: out = (start + 1)/2
    d1.step_table = lambda: rows
    z = d1.sample(param_cond=pc.repeat(22, 1)[:64].contiguous(), seeds=list(range(64))) * 2 - 1
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02


def test_end_to_end_pair_64(hip, golden):
    """BASELINE configs[0] shape: one synthetic pair, 64x64, 50-step DDIM, dim-64 networks, stored noise.
    The parity metric: L-infinity over point XYZ (metres) between HIP (fp32 mode) and the reference."""
    g = golden("G12_end_to_end_64")
    unet = golden_unet(hip, golden, 64, "fp32", W.synth_state_dict(W.unet_config(64), 12))
    mask = hip.MaskUnet(64, dtype="fp32").load_state_dict(W.synth_state_dict(W.maskunet_config(64), 13, final_bias=6.0))
    diff = golden_ddim(hip, golden, unet, 64, 50)
    K, pose = D(g["K"]), D(g["pose"])
    rpj, hit = hip.G.reproject_tensor(D(g["depth"]), K, pose, clip=(0, 10), depth_unit=10.0, out_scale=0.1)
    assert np.array_equal(rpj.cpu().numpy(), g["rpj_depth"]) and np.array_equal(hit.cpu().numpy(), g["rpj_mask"])
    prob1 = mask(rpj)
    assert maxerr(prob1, g["prob1"]) <= 1e-5
    _, _, cond = hip.G.apply_mask(prob1, rpj, hit, float(g["thr1"]))
    flips = int((cond.cpu().numpy() != g["img_cond"]).sum())
    assert flips == 0, f"{flips} condition pixels flipped at the threshold"
    img = diff.sample(param_cond=hip.G.param_vector(K), img_cond=cond, noise=D(g["noise"]))
    e_img = maxerr(img, g["sampled"])
    prob2 = mask(img)
    out, _, _ = hip.G.apply_mask(prob2, img, None, float(g["thr2"]), want_cond=False)
    same_mask = np.array_equal((out.cpu().numpy() > 0), (g["depth_out"] > 0))
    cloud = hip.G.point_clouds(out, K, pose)[0]
    env = golden("G12b_envelope")
    floor = float(env["xyz_exact"])        # reference fp32 vs exact arithmetic on this very chain: 1.41e-4 m
    e_exact = maxerr(img, env["sampled_exact"])
    print(f"end-to-end 64x64/50-step: |depth err|max = {e_img:.3e} vs reference, {e_exact:.3e} vs exact arithmetic "
          f"(reference vs exact {float(env['depth_exact']):.3e}, reference 1 thread vs 8 threads {float(env['depth_1thread']):.3e}); "
          f"mask identical = {same_mask}, points {len(cloud)} vs {len(g['cloud'])}")
    # The parity metric, unconditionally: same mask, same point count, point-XYZ L-infinity.
    assert same_mask and len(cloud) == len(g["cloud"]), "a depth-correction mask pixel flipped"
    linf = float(np.abs(cloud - g["cloud"]).max())
    print(f"point-XYZ L-infinity vs reference: {linf:.3e} m  (north star 1e-4 m; measured floor of this chain: the reference "
          f"moves by {float(env['xyz_1thread']):.2e} m with another thread count and sits {floor:.2e} m from exact arithmetic)")
    # BASELINE's tolerance is 1e-4 m.  On THIS chain the reference itself is not reproducible to 1e-4 m (G12b: 1.00e-4 m
    # between thread counts, 1.41e-4 m to exact arithmetic), so the bound is the north star or, where the measured floor
    # is above it, that floor: the HIP result may not be further from the reference than the reference is from exact
    # arithmetic (observed 1.12e-4 m), and must itself be at most HALF as far from exact arithmetic as the reference is
    # (observed 4.3e-6 vs 1.33e-5 normalised) — i.e. what remains is the reference's own roundoff.
    assert linf <= max(1e-4, XYZ_FLOOR_FACTOR * floor), linf
    assert e_exact <= 0.5 * float(env["depth_exact"])


# ------------------------------------------------------------------------------------------------------------------
# the benchmarked configuration: dim 64 at 128x128 (and the reference's shipped 256x256), against the real reference's
# fixtures and their float64 "exact arithmetic" twins
# ------------------------------------------------------------------------------------------------------------------
def _tap_report(net, g, B, label):
    rows = []
    for k in TAPS:
        t = net.get_tap(k, B).reshape(-1)[D(g[f"tap_{k}_idx"])]
        rows.append((k, maxerr(t, g[f"tap_{k}"]), maxerr(t, g[f"tap_{k}_64"]),
                     float(np.abs(g[f"tap_{k}"].astype(np.float64) - g[f"tap_{k}_64"]).max()), float(np.abs(g[f"tap_{k}_64"]).max())))
    print(f"\n{label}: tap | hip-ref32 | hip-exact | ref32-exact | scale")
    for r in rows:
        print("   %-14s %.3e  %.3e  %.3e  %.2f" % r)
    return rows


@pytest.mark.parametrize("fixture,wseed,B", [("G13_unet_dim64_128", 13, 2), ("G16_unet_dim64_256", 16, 1)])
def test_unet_dim64_full_size_taps(hip, golden, fixture, wseed, B):
    """dim-64 U-Net at 128x128 (the benchmarked resolution) and 256x256 (the shipped one): output + seven taps."""
    g = golden(fixture)
    sd = W.synth_state_dict(W.unet_config(64), wseed)
    net = golden_unet(hip, golden, 64, "fp32", sd)
    net.set_taps(True)
    y = net(D(g["x"]), D(g["t"]), D(g["pc"]))
    rows = _tap_report(net, g, B, fixture + " fp32")
    e_ref, e_exact, floor = maxerr(y, g["y"]), maxerr(y, g["y64"]), float(np.abs(g["y"].astype(np.float64) - g["y64"]).max())
    print(f"   output: hip-ref32 {e_ref:.3e}  hip-exact {e_exact:.3e}  ref32-exact {floor:.3e}")
    for k, e, _, _, _ in rows:
        assert e <= FP32_TOL, k
    assert e_ref <= 1e-5 and e_exact <= floor          # output O(5): observed 4.8e-6 vs the reference, 2.3e-6 vs exact (reference: 4.9e-6)
    net.close()
    net = golden_unet(hip, golden, 64, "bf16", sd)
    net.set_taps(True)
    y = net(D(g["x"]), D(g["t"]), D(g["pc"]))
    _tap_report(net, g, B, fixture + " bf16")
    print(f"   output bf16: max {maxerr(y, g['y']):.3e} mean {meanerr(y, g['y']):.3e}")
    assert maxerr(y, g["y"]) <= BF16_MAX and meanerr(y, g["y"]) <= BF16_MEAN


def test_chain8_dim64_128(hip, golden):
    """dim-64 ancestral chain (8 transitions, mixed DDNM mask, stored noise) at 128x128: the north-star tolerance
    (1e-5 normalised = 1e-4 m) holds in fp32 mode; bf16 drift of the same chain is reported and bounded."""
    g = golden("G14_chain8_dim64_128")
    sd = W.synth_state_dict(W.unet_config(64), 14)
    known = (g["cond"][:, 1:2] + 1) * 0.5 > 0.5
    for graph in (True, False):
        net = golden_unet(hip, golden, 64, "fp32", sd)
        d8 = hip.GaussianDiffusion(net, image_size=128, timesteps=8)
        out = d8.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["noise"]), use_graph=graph)
        e, ex = maxerr(out, g["out"]), maxerr(out, g["out64"])
        print(f"chain8@128 fp32 (graph={graph}): hip-ref32 {e:.3e}  hip-exact {ex:.3e}  ref32-exact "
              f"{float(np.abs(g['out'] - g['out64']).max()):.3e}")
        assert e <= NORTH_STAR
        assert np.array_equal(out.cpu().numpy()[known], g["out"][known])
        net.close()
    net = golden_unet(hip, golden, 64, "bf16", sd)
    d8 = hip.GaussianDiffusion(net, image_size=128, timesteps=8)
    out = d8.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["noise"]))
    assert np.array_equal(out.cpu().numpy()[known], g["out"][known])
    e, em = maxerr(out, g["out"]), meanerr(out, g["out"])
    print(f"chain8@128 bf16: in-painted pixels max {e:.3e} mean {em:.3e} (normalised depth; x10 for metres)")
    assert e <= BF16_CHAIN_MAX and em <= BF16_CHAIN_MEAN


def test_maskunet_dim64_128(hip, golden):
    g = golden("G15_maskunet_dim64_128")
    sd = W.synth_state_dict(W.maskunet_config(64), 15, final_bias=6.0)
    p = hip.MaskUnet(64, dtype="fp32").load_state_dict(sd)(D(g["depth"]))
    e = maxerr(p, g["prob"])
    print(f"MaskUnet dim64@128 fp32: {e:.3e} (exact {maxerr(p, g['prob64']):.3e})")
    assert e <= 1e-5
    p = hip.MaskUnet(64, dtype="bf16").load_state_dict(sd)(D(g["depth"]))
    e = maxerr(p, g["prob"])
    print(f"MaskUnet dim64@128 bf16: max {e:.3e} mean {meanerr(p, g['prob']):.3e}")
    assert e <= 1e-3                                           # observed 2.3e-4


def test_ddim_known_pixels_above_one(hip, golden):
    """ddim_sample does not clamp the DDNM-replaced pixels (sd:1197-1218, 1371); p_sample does (sd:1250)."""
    g = golden("G17_ddim_cond_gt1")
    net = golden_unet(hip, golden, 16, "fp32", W.synth_state_dict(W.unet_config(16), 9))
    known = (g["cond"][:, 1:2] + 1) * 0.5 > 0.5
    d5 = golden_ddim(hip, golden, net, 32, 5)
    out = d5.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["ddim5_noise"]))
    assert maxerr(out, g["ddim5_out"]) <= FP32_TOL and float(out.max()) > 1.0
    assert np.array_equal(out.cpu().numpy()[known], g["ddim5_out"][known])
    d8 = hip.GaussianDiffusion(net, image_size=32, timesteps=8)
    out = d8.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["chain8_noise"]))
    assert maxerr(out, g["chain8_out"]) <= FP32_TOL and float(out.max()) <= 1.0


def test_benchmark_batch_bf16_against_oracle(hip):
    """The benchmarked launch shapes themselves: B = 64, 128x128, dim 64, bf16, four ancestral DDNM transitions
    (t = 999, 500, 1, 0 rows of the 1000-step table) with stored noise and a mixed mask, compared per image with the
    CPU oracle (batch-invariant) for three batch slots."""
    from oracle import diffusion as OD
    from oracle import unet as OU
    from pointreggpt_amd import synthetic
    B, S = 64, 128
    sd = W.synth_state_dict(W.unet_config(64), 64)
    depth, K, pose = synthetic.synth_batch(64, range(B), S)
    Kd = D(K)
    rpj, hit = hip.G.reproject_tensor(D(depth), Kd, D(pose), clip=(0, 10), depth_unit=10.0, out_scale=0.1)
    _, _, cond = hip.G.apply_mask(torch.ones_like(rpj), rpj, hit, 0.5)
    pc = hip.G.param_vector(Kd)
    gen = torch.Generator().manual_seed(64)
    noise = torch.randn((5, B, 1, S, S), generator=gen)
    net = hip.Unet(64, dtype="bf16").load_state_dict(sd)
    d = hip.GaussianDiffusion(net, image_size=S, timesteps=1000)
    table = d.step_table()
    rows = [table[0], table[499], table[998], table[999]]
    d.step_table = lambda: rows
    out = d.sample(param_cond=pc, img_cond=cond, noise=noise.cuda()).cpu()
    sch = OD.schedule(1000)
    den = lambda x, t, c: OU.unet_forward(sd, x, t, c)
    cond_h, pc_h = cond.cpu(), pc.cpu()
    worst = 0.0
    for b in (0, 31, 63):
        x = noise[0, b:b + 1]
        for k, r in enumerate(rows):
            x, _ = OD.p_sample(sch, den, x, r["t"], pc_h[b:b + 1], cond_h[b:b + 1], noise[k + 1, b:b + 1])
        ref = (x + 1) * 0.5
        known = OD.cond_mask(cond_h[b:b + 1])
        assert torch.equal(out[b:b + 1][known], ref[known])
        e, em = float((out[b:b + 1] - ref).abs().max()), float((out[b:b + 1] - ref).abs().mean())
        print(f"B=64 bf16 slot {b}: 4 transitions, in-painted max {e:.3e} mean {em:.3e}; known pixels exact")
        worst = max(worst, e)
        assert e <= BF16_CHAIN_MAX and em <= BF16_CHAIN_MEAN
    assert worst > 0.0          # bf16 is not expected to be exact: a zero would mean the comparison is vacuous


# ------------------------------------------------------------------------------------------------------------------
# chains of REAL length against the real reference, on the calibrated (non-saturating) synthetic denoiser:
#   G19  1000-step ancestral DDNM chain @64x64 (sd:1283-1317)      G20  250-step DDIM chain @128x128 (sd:1319-1392)
# fp32 mode is held to the north star: point-XYZ L-infinity <= max(1e-4 m, the REFERENCE's own 1-thread-vs-8-thread
# spread on that chain).  bf16 and mxfp8 drift is reported in metres against the reference and bounded at <= 2x observed.
# ------------------------------------------------------------------------------------------------------------------
def golden_ancestral(hip, golden, net, S):
    """GaussianDiffusion(timesteps=1000) with the fixtures' host's float32 table (sigma = exp(0.5 logvar) is a host exp)."""
    d = hip.GaussianDiffusion(net, image_size=S, timesteps=1000)
    g0 = golden("G0_host_tables")
    rows = d.step_table()
    assert [r["t"] for r in rows] == g0["anc1000_t"].tolist()
    worst = 0.0
    for r, v in zip(rows, g0["anc1000_rows"]):
        for j, k in enumerate(("c_x0", "c_x", "c_eps", "sigma", "sqrt_recip", "sqrt_recipm1")):
            if v[j] != 0:
                worst = max(worst, abs(r[k] - float(v[j])) / abs(float(v[j])))
            r[k] = float(v[j])
    print(f"ancestral coefficients: this host vs the fixtures' host differ by up to {worst:.2e} (relative)")
    d.step_table = lambda: rows
    return d


def _run_long_chain(hip, golden, name, dtype, batch=1):
    """The fixture's chain through the C-ABI.  batch > 1: the scene replicated `batch` times (same condition, same noise) so
    that the launches have the benchmarked shapes and take the benchmarked kernels (the 256-pixel / MX kernels only run
    where a launch fills the chip); every slot must then give the same image, and slot 0 is compared."""
    from conftest import LONG_CHAINS, regenerate_chain_noise
    c = LONG_CHAINS[name]
    g = golden(name)
    sd = W.synth_state_dict(W.unet_config(64), int(g["wseed"]), calibrated=True)
    net = golden_unet(hip, golden, 64, dtype, sd)
    d = golden_ddim(hip, golden, net, c["S"], c["steps"]) if c["steps"] else golden_ancestral(hip, golden, net, c["S"])
    nz = regenerate_chain_noise(g).reshape(-1, 1, 1, c["S"], c["S"]).cuda()
    if batch > 1:
        nz = nz.expand(-1, batch, -1, -1, -1).contiguous()
    out = d.sample(param_cond=D(g["pc"]).repeat(batch, 1), img_cond=D(g["img_cond"]).repeat(batch, 1, 1, 1), noise=nz)
    if batch > 1:
        assert torch.equal(out[0], out[batch - 1]) and torch.equal(out[0], out[batch // 2]), "result depends on the batch slot"
        out = out[:1].contiguous()
    del nz
    K, pose = D(g["K"]), D(g["pose"])
    cloud = hip.G.point_clouds(out, K, pose)[0]
    img = out.cpu().numpy()
    known = (g["img_cond"][:, 1:2] + 1) * 0.5 > 0.5
    assert np.array_equal(img[known], g["sampled"][known]), "DDNM known pixels must be bit-exact in every precision mode"
    valid = lambda a: (a[0, 0] * 10 > 0.5) & (a[0, 0] * 10 < 10)
    v_hip, v_ref = valid(img), valid(g["sampled"])
    free = ~known
    dd = np.abs(img.astype(np.float64) - g["sampled"])[free] * 10.0          # metres, in-painted pixels
    rep = {"depth_max_m": float(dd.max()), "depth_mean_m": float(dd.mean()), "depth_median_m": float(np.median(dd)),
           "same_valid_mask": bool(np.array_equal(v_hip, v_ref)), "points": (len(cloud), len(g["cloud"]))}
    if rep["same_valid_mask"]:
        rep["xyz_linf_m"] = float(np.abs(cloud - g["cloud"]).max())
    else:                                                                          # compare the points both keep
        both = (v_hip & v_ref).reshape(-1)
        ch = np.full((v_hip.size, 3), np.nan); ch[v_hip.reshape(-1)] = cloud
        cr = np.full((v_ref.size, 3), np.nan); cr[v_ref.reshape(-1)] = g["cloud"]
        rep["xyz_linf_m"] = float(np.abs(ch[both] - cr[both]).max())
    rep["saturated_fraction"] = float(((img <= 0) | (img >= 1))[free].mean())
    d.close(); net.close()
    return g, rep, img


@pytest.mark.parametrize("name", ["G19_chain1000_ancestral_64", "G20_ddim250_128", "G21_ddim250_256", "G21b_ddim250_256", "G22_chain1000_ancestral_128"])
def test_long_chain_fp32_north_star(hip, golden, name):
    g, rep, img = _run_long_chain(hip, golden, name, "fp32")
    spread = float(g["xyz_spread_1_vs_8_threads_m"])
    print(f"{name} fp32: point-XYZ L-inf vs reference {rep['xyz_linf_m']:.3e} m (north star 1e-4 m; the reference moves by "
          f"{spread:.3e} m between 1 and 8 threads and sits {float(g['xyz_ref_to_exact_m']):.3e} m from its float64 twin); "
          f"in-painted depth max {rep['depth_max_m']:.3e} m mean {rep['depth_mean_m']:.3e} m; "
          f"|hip - exact|max {maxerr(torch.from_numpy(img), g['sampled_exact']) * 10:.3e} m; saturated {rep['saturated_fraction']:.4f}")
    assert rep["same_valid_mask"] and rep["points"][0] == rep["points"][1]
    assert rep["saturated_fraction"] < 0.2
    if name != "G21_ddim250_256":
        # the north star, literally (round 4): observed G19 4.5e-6 m, G20 2.8e-5 m, G21b 3.6e-5 m, G22 (the headline chain) 6.0e-6 m
        assert rep["xyz_linf_m"] <= 1e-4, rep
    else:
        # G21 (seed-21 weights at 256x256, kept as the stress case; G21b is the same chain on a calibrated seed): 16.6 % of the
        # REFERENCE's own in-painted pixels end on the clamp and the reference sits 4.3e-4 m from its float64 twin, farther than its
        # 1-vs-8-thread spread (1.8e-4 m) — an implementation that is CLOSER to exact arithmetic cannot also be inside that spread.
        # There: at most half as far from exact arithmetic as the reference is, and no farther from it than it is from exact.
        e_exact = maxerr(torch.from_numpy(img), g["sampled_exact"])
        print(f"{name}: |hip - exact| {e_exact:.3e} vs |reference - exact| {float(g['depth_ref_to_exact']):.3e} (depth)")
        assert e_exact <= 0.5 * float(g["depth_ref_to_exact"]), (e_exact, float(g["depth_ref_to_exact"]))
        assert rep["xyz_linf_m"] <= 1.1 * float(g["xyz_ref_to_exact_m"]), rep


# <= 2x observed (printed by the test; B = 64, the benchmarked launch shapes): (depth max, depth mean) of the in-painted pixels
# and point-XYZ L-infinity, metres.  The MAXIMUM over ~5-10 k in-painted pixels is a heavy-tailed statistic (a pixel near a
# depth discontinuity of the network's own output moves by decimetres under any perturbation); the mean is the stable one.
# observed (B = 64): bf16 G19 0.038 / 0.0093 / 0.038, G20 0.62 / 0.025 / 0.61; mxfp8 G19 0.129 / 0.031 / 0.130, G20 1.22 / 0.046 / 1.20
@pytest.mark.parametrize("name,nb", [("G22_chain1000_ancestral_128", 64), ("G21b_ddim250_256", 16)])
def test_long_chain_fp32_north_star_at_the_benchmarked_batch(hip, golden, name, nb):
    """Round 6: the fp32 parity mode on the headline chain (G22: 1000-step ancestral DDNM @128x128) and on
    the shipped setting's chain (G21b: 250-step DDIM @256x256) at the batch `parity_mode.fp32` / the configs[4] legs are TIMED on
    (B = 64 / B = 16): the launches then take the kernels of the benchmarked shapes (sd:1283-1317, sd:1319-1392).  The scene is
    replicated over the batch; every slot must give the same image (asserted in _run_long_chain) and slot 0 holds the north star."""
    g, rep, img = _run_long_chain(hip, golden, name, "fp32", batch=nb)
    print(f"{name} fp32 (B={nb}): point-XYZ L-inf vs reference {rep['xyz_linf_m']:.3e} m (north star 1e-4 m); in-painted depth max "
          f"{rep['depth_max_m']:.3e} m mean {rep['depth_mean_m']:.3e} m; saturated {rep['saturated_fraction']:.4f}")
    assert rep["same_valid_mask"] and rep["points"][0] == rep["points"][1]
    assert rep["xyz_linf_m"] <= 1e-4, rep


LONG_DRIFT_BOUNDS = {("G19_chain1000_ancestral_64", "bf16"): (0.08, 0.02, 0.08), ("G19_chain1000_ancestral_64", "mxfp8"): (0.26, 0.062, 0.26),
                     ("G20_ddim250_128", "bf16"): (1.3, 0.05, 1.3), ("G20_ddim250_128", "mxfp8"): (2.5, 0.093, 2.4),
                     # BASELINE configs[4]'s own chain and format (256x256, 250-step DDIM; B = 16)
                     # (16.6 % of this chain's in-painted pixels sit on the clamp in the reference itself: a pixel that saturates in one
                     # run and not in the other differs by metres, so the maximum / L-infinity are not bounded here — the mean is;
                     # observed bf16 mean 4.5 cm / median 1.2 cm, mxfp8 mean 8.9 cm / median 4.0 cm)
                     ("G21_ddim250_256", "bf16"): (None, 0.09, None), ("G21_ddim250_256", "mxfp8"): (None, 0.18, None),
                     # G21 on a calibrated seed (round 4): bounds set from the first run below
                     # observed: bf16 max 0.84 / mean 0.0257 / xyz 0.88 m, mxfp8 0.92 / 0.0439 / 0.90 (the maximum is a depth-discontinuity pixel)
                     ("G21b_ddim250_256", "bf16"): (None, 0.052, None), ("G21b_ddim250_256", "mxfp8"): (None, 0.09, None),
                     # the headline chain itself (round 4): 1000-step ancestral DDNM @128x128, B = 64
                     # observed: bf16 0.052 / 0.0107 / 0.053, mxfp8 0.277 / 0.059 / 0.289
                     ("G22_chain1000_ancestral_128", "bf16"): (0.11, 0.022, 0.11), ("G22_chain1000_ancestral_128", "mxfp8"): (0.56, 0.12, 0.58)}


@pytest.mark.parametrize("name", ["G19_chain1000_ancestral_64", "G20_ddim250_128", "G21_ddim250_256", "G21b_ddim250_256", "G22_chain1000_ancestral_128"])
@pytest.mark.parametrize("dtype", ["bf16", "mxfp8"])
def test_long_chain_reduced_precision_drift_in_metres(hip, golden, name, dtype):
    """The throughput modes on the same chains, at the benchmarked batch sizes (B = 64 at 64x64 / 128x128, B = 16 at 256x256:
    the kernels bench.py times), against the REFERENCE (not against this library's fp32 mode)."""
    from conftest import LONG_CHAINS
    nb = LONG_CHAINS[name]["batch"]
    g, rep, _ = _run_long_chain(hip, golden, name, dtype, batch=nb)
    print(f"{name} {dtype} (B={nb}): in-painted depth vs reference max {rep['depth_max_m']:.3e} m mean {rep['depth_mean_m']:.3e} m "
          f"median {rep['depth_median_m']:.3e} m; point-XYZ L-inf {rep['xyz_linf_m']:.3e} m; same valid mask "
          f"{rep['same_valid_mask']}; saturated {rep['saturated_fraction']:.4f}")
    assert rep["saturated_fraction"] < 0.2
    mx, mean, xyz = LONG_DRIFT_BOUNDS[(name, dtype)]
    assert rep["depth_mean_m"] <= mean and rep["depth_median_m"] <= mean, rep
    assert (mx is None or rep["depth_max_m"] <= mx) and (xyz is None or rep["xyz_linf_m"] <= xyz), rep


# ------------------------------------------------------------------------------------------------------------------
# successive multi-view generation (Tester.sample / Tester.generate, sd:1960-2247): refine step, occlusion filter
# ------------------------------------------------------------------------------------------------------------------
def test_refine_step_and_occlusion_filter(hip, golden):
    g = golden("G18_refine_occlusion_transform")
    net = golden_unet(hip, golden, 16, "fp32", W.synth_state_dict(W.unet_config(16), 9))
    known = (g["cond"][:, 1:2] + 1) * 0.5 > 0.5
    d5 = golden_ddim(hip, golden, net, 32, 5)
    out = d5.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["ddim5_refine_noise"]), has_refine_step=True)
    assert maxerr(out, g["ddim5_refine_out"]) <= FP32_TOL
    # the refine step rewrites exactly the KNOWN pixels (with the network's own prediction); without it they hold the condition
    plain = d5.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["ddim5_refine_noise"]))
    assert torch.equal(plain.cpu()[torch.from_numpy(~known)], out.cpu()[torch.from_numpy(~known)])
    assert not torch.equal(plain.cpu()[torch.from_numpy(known)], out.cpu()[torch.from_numpy(known)])
    d8 = hip.GaussianDiffusion(net, image_size=32, timesteps=8)
    out = d8.sample(param_cond=D(g["pc"]), img_cond=D(g["cond"]), noise=D(g["chain8_refine_noise"]), has_refine_step=True)
    assert maxerr(out, g["chain8_refine_out"]) <= FP32_TOL
    d, m = hip.G.occlusion_filter(D(g["of_depth_in"]), D(g["of_mask_in"]))
    assert np.array_equal(d.cpu().numpy(), g["of_depth_out"]) and np.array_equal(m.cpu().numpy(), g["of_mask_out"])


def test_tester_successive_views_against_oracle_sequence(hip, tmp_path):
    """Tester.sample (sd:1960-2093): unconditional view, then two views 0.5 m further each, conditioned on the reprojected,
    occlusion-filtered previous view — the same sequence restated with the oracle's functions on the stored noise."""
    from oracle import diffusion as OD
    from oracle import geometry as OG
    from oracle import unet as OU
    from pointreggpt_amd.tester import Tester
    from pointreggpt_amd import postprocess as PP
    S, B, steps, views = 32, 2, 4, 3
    sd = W.synth_state_dict(W.unet_config(16), 31)
    net = hip.Unet(16, dtype="fp32").load_state_dict(sd)
    diff = hip.GaussianDiffusion(net, image_size=S, timesteps=1000, sampling_timesteps=steps)
    gen = torch.Generator().manual_seed(31)
    noise = [torch.randn((steps, B, 1, S, S), generator=gen) for _ in range(views)]
    np.random.seed(31)
    strips = Tester(diff, batch_size=B, samples_folder=str(tmp_path / "s")).sample(B, views, noise=noise)
    assert len(strips) == 1 and tuple(strips[0].shape) == (B, 1, S, S * views)
    # oracle restatement
    np.random.seed(31)
    K = hip.G.intrinsic_transform(hip.G.random_sample_intrinsic(B), resize=S, centercrop=S).astype(np.float32)
    sch, Kt = OD.schedule(1000), torch.from_numpy(K)
    pc = OG.param_vector(Kt)
    den = lambda x, t, c: OU.unet_forward(sd, x, t, c)
    img = OD.sample(sch, den, pc, None, S, OD.stored_noise(noise[0]), sampling_steps=steps)
    got = strips[0].cpu()
    assert float((got[..., :S] - img).abs().max()) <= FP32_TOL
    absolute = np.stack([np.eye(4)] * B).astype(np.float32)
    for k in range(1, views):
        rel = np.stack([np.eye(4)] * B)
        rel[:, :3, 3] = [0, 0, 0.5]
        rel = rel.astype(np.float32)
        absolute = rel @ absolute
        # condition built from the HIP path's own previous view: a last-bit difference upstream must not move a z-buffer pixel
        prev = got[..., (k - 1) * S:k * S].contiguous()
        d_rpj, hit = OG.reproject_tensor(prev * 10, Kt, torch.from_numpy(rel), (0, 10))
        d_rpj, hit = OG.occlusion_filter(d_rpj, hit)
        cond = torch.cat([d_rpj * 0.1, hit.float()], 1) * 2 - 1
        img = OD.sample(sch, den, pc, cond, S, OD.stored_noise(noise[k]), sampling_steps=steps)
        view = got[..., k * S:(k + 1) * S]
        known = OD.cond_mask(cond)
        assert float((view - img).abs().max()) <= FP32_TOL, k
        assert int((view[known] != img[known]).sum()) <= 2        # (2-ulp host-BLAS effect on z)
        cloud = PP.read_ply(str(tmp_path / "s" / f"scene-1-sample-{k}.ply"))
        ref = OG.inverse_pose_apply(OG.point_cloud(view[1, 0].numpy() * 10, K[1], (0.5, 3.5)), absolute[1])
        assert cloud.shape == ref.shape and np.abs(cloud - ref).max() < 1e-12
    assert (tmp_path / "s" / "scene-0-camera-intrinsics.txt").is_file() and (tmp_path / "s" / "scene-1-sample-2.png").is_file()


def test_tester_generate_accumulates_a_scene(hip, tmp_path):
    """Tester.generate (sd:2095-2247): random in-place rotations, accumulated cloud re-projected as the condition."""
    from pointreggpt_amd.tester import Tester
    from pointreggpt_amd import postprocess as PP
    net = hip.Unet(16, dtype="fp32").init_synthetic(32)
    diff = hip.GaussianDiffusion(net, image_size=32, timesteps=1000, sampling_timesteps=3)
    np.random.seed(5)
    t = Tester(diff, batch_size=2, samples_folder=str(tmp_path / "g"), seed=9)
    scenes = t.generate(3, 3, voxel_size=0.005)
    assert [len(b) for b in scenes] == [2, 1]
    for i in range(3):
        final = PP.read_ply(str(tmp_path / "g" / f"scene-{i}.ply"))
        acc = scenes[i // 2][i % 2]
        assert acc.dtype == np.float32 and len(acc) > 0 and np.isfinite(final).all()
        ref = PP.voxel_down_sample(acc, 0.025)
        assert final.shape == ref.shape and np.abs(final - ref).max() < 1e-12
    # same seeds, same scenes: the run is reproducible (Philox keys per scene and view, numpy stream for the rotations)
    np.random.seed(5)
    again = Tester(diff, batch_size=2, samples_folder=str(tmp_path / "g2"), seed=9).generate(3, 3, voxel_size=0.005)
    assert all(np.array_equal(a, b) for ba, bb in zip(scenes, again) for a, b in zip(ba, bb))


def test_overlap_counts_against_kdtree_oracle(hip):
    """generate_gt's overlap ratio (generate_gt.py:68-102): HIP all-pairs kernel vs scipy cKDTree (oracle) and the numpy
    grid specification, on overlapping, disjoint, tiny and empty clouds."""
    from oracle import postprocess as OP
    from pointreggpt_amd import postprocess as PP
    rng = np.random.default_rng(8)

    def cloud(n, shift):
        return np.c_[rng.uniform(-1.2, 1.2, n) + shift, rng.uniform(-1.0, 1.0, n), 2.0 + 0.05 * rng.standard_normal(n)]

    pairs = [(cloud(6000, 0.0), cloud(5000, 0.4)), (cloud(3000, 0.0), cloud(3000, 5.0)), (cloud(40, 0.0), cloud(4000, 0.0)),
             (cloud(2500, 0.0), cloud(2500, 0.02))]
    got = PP.overlap_ratios_hip(pairs)
    for (a, b), (o1, o2) in zip(pairs, got):
        r1, r2 = OP.overlap_ratio(a, b)
        s1, s2 = PP.compute_overlap_ratio(a, b)
        assert (o1, o2) == (s1, s2), ((o1, o2), (s1, s2))          # identical counts -> identical float64 ratios
        assert abs(o1 - r1) < 1e-12 and abs(o2 - r2) < 1e-12
    assert got[1] == (0.0, 0.0) and 0.2 < got[0][0] < 1.0
    e = PP.overlap_ratios_hip([(np.zeros((0, 3)), cloud(100, 0.0))])
    assert np.isnan(e[0][0]) and e[0][1] == 0.0


# ------------------------------------------------------------------------------------------------------------------
# conv kernels one at a time (prg_debug_conv3x3) and the MX-fp8 operand path of BASELINE configs[4]
# ------------------------------------------------------------------------------------------------------------------
def _debug_conv(hip, x, w, bias, dtype):
    import ctypes as C
    lib = hip.lib.load()
    B, Cin, H, Wd = x.shape
    Cout = w.shape[0]
    out = torch.empty((B, Cout, H, Wd), dtype=torch.float32, device="cuda")
    wh = np.ascontiguousarray(w.numpy(), dtype=np.float32)
    bh = None if bias is None else np.ascontiguousarray(bias.numpy(), dtype=np.float32)
    hip.lib.check(lib.prg_debug_conv3x3(hip.lib.ptr(x.cuda().contiguous()), wh.ctypes.data_as(C.c_void_p),
                                        None if bh is None else bh.ctypes.data_as(C.c_void_p), hip.lib.ptr(out), B, Cin, Cout, H,
                                        Wd, dtype, hip.lib.stream_ptr()), "prg_debug_conv3x3")
    return out.cpu()


CONV_SHAPES = [(2, 64, 64, 32, 64),     # 64 -> 64: the weights-stationary kernel (16 tiles)
               (2, 64, 64, 16, 32),     # 64 -> 64, too few tiles for it: wave-specialised 8x32x64
               (1, 128, 128, 8, 32),    # 4x32x128 tiles, two channel chunks
               (2, 64, 128, 16, 16),    # 8x16x128 tiles
               (1, 192, 64, 8, 32)]     # three chunks
CONV_SHAPES_BF16 = CONV_SHAPES + [(16, 128, 128, 64, 64),  # 256 tiles of 8x32x128: the 256-pixel kernel, two chunks
                                  (8, 64, 256, 64, 64),    # ... two output-channel tiles (XCDs pinned to one each)
                                  (256, 64, 128, 16, 16)]  # ... 16x16 tiles


@pytest.mark.parametrize("B,Cin,Cout,H,Wd", CONV_SHAPES_BF16)
def test_conv3x3_kernels_one_at_a_time(hip, B, Cin, Cout, H, Wd):
    """Each bf16 3x3 kernel of the dispatch against a float64 convolution of the SAME bf16-rounded operands: what remains
    is fp32 accumulation order and the bf16 rounding of the output (half an ulp = 2^-9 relative)."""
    g = torch.Generator().manual_seed(Cin * 1000 + Cout + H)
    x = torch.randn((B, Cin, H, Wd), generator=g) * torch.exp(torch.randn((1, Cin, 1, 1), generator=g))
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / (3.0 * Cin ** 0.5)
    bias = torch.randn((Cout,), generator=g)
    got = _debug_conv(hip, x, w, bias, hip.lib.PRG_BF16)
    xb, wb = x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double()
    ref = torch.nn.functional.conv2d(xb, wb, bias.double(), padding=1)
    err = (got.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 1e-4 * float(ref.abs().max())
    assert bool((err <= tol).all()), float((err - tol).max())


UP_SHAPES = [(16, 512, 256, 16, 16),   # 16 x 16 source tiles, two output-channel tiles (up level 0 -> 1)
             (8, 256, 128, 32, 32),    # 8 x 32 source tiles (up level 1 -> 2)
             (8, 128, 64, 64, 64),     # Cout = 64: the second channel half of the consumers idles (up level 2 -> 3)
             (16, 64, 128, 16, 32),    # one 64-channel chunk per phase, 16 x 16 tiles on a non-square image
             (1, 128, 64, 16, 16)]     # too small for it: the nine-tap gather form of the older kernels


@pytest.mark.parametrize("B,Cin,Cout,H,Wd", UP_SHAPES)
def test_upsample_conv_subpixel_form(hip, B, Cin, Cout, H, Wd):
    """Upsample = nn.Upsample(x2, nearest) + Conv2d(C, Cout, 3, pad 1) (sd:592-594) through the dispatch: the 256-pixel
    kernel's MODE 2 (four 2 x 2-tap sub-pixel convolutions of the source image with pre-summed weights: 4 / 9 of the MACs)
    against a float64 convolution of the nearest-upsampled bf16-rounded input with the UNROUNDED weights.  What remains: bf16
    rounding of the (pre-summed) weights and of the output."""
    import ctypes as C
    lib = hip.lib.load()
    g = torch.Generator().manual_seed(Cin * 53 + Cout + H)
    x = torch.randn((B, Cin, H, Wd), generator=g) * torch.exp(0.5 * torch.randn((1, Cin, 1, 1), generator=g))
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / (3.0 * Cin ** 0.5)
    bias = torch.randn((Cout,), generator=g)
    out = torch.empty((B, Cout, 2 * H, 2 * Wd), dtype=torch.float32, device="cuda")
    wh, bh = np.ascontiguousarray(w.numpy(), dtype=np.float32), np.ascontiguousarray(bias.numpy(), dtype=np.float32)
    hip.lib.check(lib.prg_debug_upsample_conv3x3(hip.lib.ptr(x.cuda().contiguous()), wh.ctypes.data_as(C.c_void_p), bh.ctypes.data_as(C.c_void_p),
                                                 hip.lib.ptr(out), B, Cin, Cout, H, Wd, hip.lib.stream_ptr()), "prg_debug_upsample_conv3x3")
    xu = torch.nn.functional.interpolate(x.to(torch.bfloat16).double(), scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(xu, w.double(), bias.double(), padding=1)
    err = (out.cpu().double() - ref).abs()
    scale = float(ref.abs().max())
    print(f"upsample conv {Cin}->{Cout} @{H}x{Wd} B={B}: max {float(err.max()):.3e} mean {float(err.mean()):.3e} (|ref| max {scale:.2f})")
    # bf16 output rounding (2^-9 relative) + bf16 weight rounding summed over K = 9 Cin terms (random signs: ~2^-9 |ref| rms)
    tol = 2.0 ** -7 * ref.abs() + 2.0 ** -8 * scale
    assert bool((err <= tol).all()), float((err - tol).max())
    assert float(err.mean()) <= 2.0 ** -9 * scale


# (B, Cin, C, H, W): conv1 / conv2 kernels of the pair
PAIR_SHAPES = [(4, 64, 64, 64, 64),      # c64 -> c64 (down levels 0-1)
               (8, 64, 64, 128, 128),    # ... at two tiles per CU: the one-wave-per-SIMD c64w kernel on both sides (round 6)
               (4, 128, 64, 64, 64),     # wave-specialised 8x32x64 (Cin = 128) -> c64 (up level 3, final block)
               (64, 128, 128, 32, 32),   # 256-pixel kernel on both sides, 8x32 tiles, one channel tile (level 2)
               (128, 256, 256, 16, 16)]  # ... 16x16 tiles, two channel tiles (256 workgroups): every input pixel transformed twice


@pytest.mark.parametrize("B,Cin,C,H,Wd", PAIR_SHAPES)
def test_block_pair_h16_against_float64(hip, B, Cin, C, H, Wd):
    """Round 4, `h16`: conv1 -> [GroupNorm + SiLU in conv2's prologue] -> conv2 with the tensor in between stored as
    f16 (packed-f16 prologue, f16 MFMA operands) and as bf16 (float32 prologue, bf16 operands), each against the float64
    evaluation of the same chain on the same bf16-rounded inputs and conv1 weights.  The f16 form must be the MORE accurate
    one (11-bit tensor and operands instead of 8) and stay inside a bf16-output bound."""
    import ctypes as C_
    lib = hip.lib.load()
    g = torch.Generator().manual_seed(Cin * 131 + C + H)
    x = torch.randn((B, Cin, H, Wd), generator=g) * torch.exp(0.5 * torch.randn((1, Cin, 1, 1), generator=g))
    w1 = torch.randn((C, Cin, 3, 3), generator=g) / (3.0 * Cin ** 0.5)
    w2 = torch.randn((C, C, 3, 3), generator=g) / (3.0 * C ** 0.5)
    b1, b2 = torch.randn((C,), generator=g), torch.randn((C,), generator=g)
    gamma, beta = 1.0 + 0.3 * torch.randn((C,), generator=g), 0.3 * torch.randn((C,), generator=g)
    groups = 8
    # float64 chain: bf16-rounded x and w1 (what both forms read); h, the norm and conv2 exact
    h = torch.nn.functional.conv2d(x.to(torch.bfloat16).double(), w1.to(torch.bfloat16).double(), b1.double(), padding=1)
    a = torch.nn.functional.silu(torch.nn.functional.group_norm(h, groups, gamma.double(), beta.double(), eps=1e-5))
    ref = torch.nn.functional.conv2d(a, w2.double(), b2.double(), padding=1)
    hp = lambda t: np.ascontiguousarray(t.numpy(), dtype=np.float32)
    arrs = [hp(t) for t in (w1, b1, gamma, beta, w2, b2)]
    xd = x.cuda().contiguous()
    errs = {}
    for h16 in (0, 1):
        out = torch.empty((B, C, H, Wd), dtype=torch.float32, device="cuda")
        hip.lib.check(lib.prg_debug_block_pair(hip.lib.ptr(xd), *[a_.ctypes.data_as(C_.c_void_p) for a_ in arrs], hip.lib.ptr(out),
                                               B, Cin, C, H, Wd, groups, h16, hip.lib.stream_ptr()), "prg_debug_block_pair")
        e = (out.cpu().double() - ref).abs()
        assert bool(torch.isfinite(out).all())
        errs[h16] = (float(e.max()), float(e.mean()))
    scale = float(ref.abs().max())
    print(f"block pair {Cin}->{C} @{H}x{Wd} B={B}: bf16 h max {errs[0][0]:.3e} mean {errs[0][1]:.3e} | f16 h max {errs[1][0]:.3e} mean {errs[1][1]:.3e} "
          f"(|ref| max {scale:.2f})")
    assert errs[1][1] <= errs[0][1], errs                       # the f16 form is the more accurate one on average ...
    assert errs[1][0] <= 2.0 ** -7 * scale and errs[0][0] <= 2.0 ** -6 * scale, (errs, scale)   # ... both inside their rounding budgets


def test_c64w_kernel_matches_the_c64_kernel(tmp_path):
    """Round 6: conv3x3_c64w_kernel (one 512-register wave per SIMD holding the weights of all 64 output channels: 0.5 fragment reads
    per MFMA) runs conv3x3_c64_kernel's MFMA sequence per accumulator — the plain 64 -> 64 convolution must give the SAME BITS with
    the kernel on (PRG_CONV_C64W=1; it is off by default: profiles/r06_c64w_ablations.txt) and off at a shape that selects it (two 8 x 32 tiles per CU), on images with every
    kind of border tile; the ResnetBlock pair (f16 tensor in between, statistics in the epilogue, folded prologue) agrees to the
    last-bit differences of the GroupNorm statistics (a wave totals 64 pixels x 64 channels instead of 128 x 32)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, %r)
from pointreggpt_amd import _lib
lib = _lib.load()
out = {}
g = torch.Generator().manual_seed(11)
for (B, H, Wd) in [(8, 128, 128), (16, 64, 128)]:
    x = torch.randn((B, 64, H, Wd), generator=g).cuda()
    w = np.ascontiguousarray((torch.randn((64, 64, 3, 3), generator=g) / 24.0).numpy())
    bias = np.ascontiguousarray(torch.randn((64,), generator=g).numpy())
    o = torch.empty((B, 64, H, Wd), dtype=torch.float32, device="cuda")
    _lib.check(lib.prg_debug_conv3x3(_lib.ptr(x), w.ctypes.data_as(C.c_void_p), bias.ctypes.data_as(C.c_void_p), _lib.ptr(o), B, 64, 64, H, Wd,
                                     _lib.PRG_BF16, _lib.stream_ptr()))
    out["conv_%%d_%%d" %% (B, H)] = o.cpu().numpy()
B, H = 8, 128
x = torch.randn((B, 64, H, H), generator=g).cuda()
w1 = (torch.randn((64, 64, 3, 3), generator=g) / 24.0); w2 = (torch.randn((64, 64, 3, 3), generator=g) / 24.0)
b1, b2 = torch.randn((64,), generator=g), torch.randn((64,), generator=g)
gamma, beta = 1.0 + 0.3 * torch.randn((64,), generator=g), 0.3 * torch.randn((64,), generator=g)
arrs = [np.ascontiguousarray(t.numpy(), dtype=np.float32) for t in (w1, b1, gamma, beta, w2, b2)]
o = torch.empty((B, 64, H, H), dtype=torch.float32, device="cuda")
_lib.check(lib.prg_debug_block_pair(_lib.ptr(x), *[a.ctypes.data_as(C.c_void_p) for a in arrs], _lib.ptr(o), B, 64, 64, H, H, 8, 1, _lib.stream_ptr()))
out["pair"] = o.cpu().numpy()
np.savez(sys.argv[1], **out)
''' % root
    res = {}
    for on in ("0", "1"):
        path = tmp_path / f"c64w_{on}.npz"
        r = subprocess.run([sys.executable, "-c", code, str(path)], capture_output=True, text=True, timeout=900, env=dict(os.environ, PRG_CONV_C64W=on))
        assert r.returncode == 0, r.stderr[-2000:]
        res[on] = np.load(path)
    for k in res["0"].files:
        assert np.isfinite(res["1"][k]).all(), k
        if k.startswith("conv"):
            assert np.array_equal(res["0"][k], res["1"][k]), (k, float(np.abs(res["0"][k] - res["1"][k]).max()))
        else:
            d = np.abs(res["0"][k].astype(np.float64) - res["1"][k])
            print(f"block pair through c64w vs c64: max |diff| {d.max():.3e} on |y| <= {np.abs(res['0'][k]).max():.2f}, differing elements {float((d > 0).mean()):.4f}")
            assert d.max() <= 2.0 ** -7 * np.abs(res["0"][k]).max() and float((d > 0).mean()) < 0.05


def test_block_pair_h16_overflow_is_visible(hip):
    """An h value beyond the f16 range (65504) must not be clipped silently: the block's output carries NaN / inf."""
    import ctypes as C_
    lib = hip.lib.load()
    B, Cc, H = 4, 64, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn((B, Cc, H, H), generator=g)
    x[1, :, 10:14, 20:24] = 1.2e4                                                   # conv1 sums reach 1.4e5 there (f16 max: 65504)
    w1 = torch.full((Cc, Cc, 3, 3), 0.02)
    w2 = torch.randn((Cc, Cc, 3, 3), generator=g) / 24.0
    z, one = torch.zeros(Cc), torch.ones(Cc)
    hp = lambda t: np.ascontiguousarray(t.numpy(), dtype=np.float32)
    arrs = [hp(t) for t in (w1, z, one, z, w2, z)]
    out = torch.empty((B, Cc, H, H), dtype=torch.float32, device="cuda")
    hip.lib.check(lib.prg_debug_block_pair(hip.lib.ptr(x.cuda().contiguous()), *[a_.ctypes.data_as(C_.c_void_p) for a_ in arrs], hip.lib.ptr(out),
                                           B, Cc, Cc, H, H, 8, 1, hip.lib.stream_ptr()), "prg_debug_block_pair")
    o = out.cpu()
    assert not bool(torch.isfinite(o[1]).all())              # the image with the out-of-range patch
    assert bool(torch.isfinite(o[0]).all()) and bool(torch.isfinite(o[2:]).all())   # its neighbours in the batch are untouched


DOWN_SHAPES = [(64, 64, 64, 64, 64),     # Cout = 64: second channel half of the consumers idle; 8 x 32 tiles, 256 of them
               (32, 64, 128, 64, 64),    # 128 tiles: half the CUs
               (64, 128, 256, 32, 32),   # 16 x 16 tiles, two sub-pixel chunks per source pixel, two channel tiles
               (2, 64, 128, 16, 32)]     # too small for it: the generic implicit-GEMM kernel


@pytest.mark.parametrize("B,Cin,Cout,H,Wd", DOWN_SHAPES)
def test_downsample_conv4x4s2(hip, B, Cin, Cout, H, Wd):
    """Downsample = Conv2d(C, Cout, 4, 2, 1) (sd:596-597): the 256-pixel kernel's 2 x 2-tap mode over the space-to-depth
    view of the shifted input (and the generic kernel for small launches) against a float64 strided convolution of the
    same bf16-rounded operands."""
    import ctypes as C
    g = torch.Generator().manual_seed(Cin * 77 + Cout + H)
    x = torch.randn((B, Cin, H, Wd), generator=g) * torch.exp(torch.randn((1, Cin, 1, 1), generator=g))
    w = torch.randn((Cout, Cin, 4, 4), generator=g) / (4.0 * Cin ** 0.5)
    bias = torch.randn((Cout,), generator=g)
    lib = hip.lib.load()
    out = torch.empty((B, Cout, H // 2, Wd // 2), dtype=torch.float32, device="cuda")
    wh = np.ascontiguousarray(w.numpy(), dtype=np.float32)
    bh = np.ascontiguousarray(bias.numpy(), dtype=np.float32)
    hip.lib.check(lib.prg_debug_conv4x4s2(hip.lib.ptr(x.cuda().contiguous()), wh.ctypes.data_as(C.c_void_p),
                                          bh.ctypes.data_as(C.c_void_p), hip.lib.ptr(out), B, Cin, Cout, H, Wd,
                                          hip.lib.stream_ptr()), "prg_debug_conv4x4s2")
    xb, wb = x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float()
    ref = torch.nn.functional.conv2d(xb.double()[:8], wb.double(), bias.double(), stride=2, padding=1)   # float64 on a slice
    ref32 = torch.nn.functional.conv2d(xb, wb, bias, stride=2, padding=1)                                   # float32 on all
    got = out.cpu()
    tol = 2.0 ** -8 * ref.abs() + 1e-4 * float(ref.abs().max())
    err = (got[:8].double() - ref).abs()
    assert bool((err <= tol).all()), float((err - tol).max())
    tol32 = 2.0 ** -7 * ref32.abs() + 1e-3 * float(ref32.abs().max())
    assert bool(((got - ref32).abs() <= tol32).all())


CONV_SHAPES_MX = CONV_SHAPES + [(8, 64, 128, 64, 64),     # 128 tiles of 8x32x128: the 256-pixel MX kernel, one chunk
                                (8, 128, 256, 32, 64),    # ... two chunks, two output-channel tiles
                                (128, 64, 128, 16, 16)]   # ... 16x16 tiles


@pytest.mark.parametrize("B,Cin,Cout,H,Wd", CONV_SHAPES_MX)
def test_mxfp8_conv_matches_block_scaled_reference(hip, B, Cin, Cout, H, Wd):
    """v_mfma_scale_f32_32x32x64_f8f6f4 path: device-side quantisation of the activations (block maxima, E8M0 scales,
    e4m3 rounding) and host-side quantisation of the weights, against the numpy restatement of OCP MX (oracle/mx.py)."""
    from oracle import mx
    g = torch.Generator().manual_seed(Cin * 77 + Cout + Wd)
    # per-(pixel, block) dynamic range over ~6 binades, one all-zero block, values beyond the e4m3 range of their block
    x = torch.randn((B, Cin, H, Wd), generator=g) * torch.exp2(torch.randint(-3, 4, (B, Cin // 32, H, Wd), generator=g)
                                                               .repeat_interleave(32, dim=1).float())
    x[0, :32, 0, :5] = 0.0
    w = torch.randn((Cout, Cin, 3, 3), generator=g) / (3.0 * Cin ** 0.5)
    bias = torch.randn((Cout,), generator=g)
    got = _debug_conv(hip, x, w, bias, hip.lib.PRG_MXFP8)
    ref = mx.conv3x3_mx_reference(x, w, bias)
    err = (got.double() - ref).abs()
    tol = 2.0 ** -8 * ref.abs() + 1e-4 * float(ref.abs().max())
    assert bool((err <= tol).all()), (float((err - tol).max()), float(ref.abs().max()))
    # and the quantisation is not a no-op: the same conv on unquantised bf16 operands differs visibly
    plain = torch.nn.functional.conv2d(x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), bias.double(), padding=1)
    assert float((plain - ref).abs().max()) > 3 * float(tol.max())


_MX_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
from pointreggpt_amd import weights as W
from pointreggpt_amd.unet import Unet
g = np.load({gold!r})
sd = W.synth_state_dict(W.unet_config(64), 13)
net = Unet(64, dtype="mxfp8").load_state_dict(sd)
D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
tab = np.load({tables!r})
net.set_time_freqs(tab["freqs_dim64"])
y = net(D(g["x"]), D(g["t"]), D(g["pc"])).float().cpu().numpy()
np.save({out!r}, y)
"""


def test_unet_mxfp8_kernel_variants(tmp_path, golden):
    """The MX operand path has three dispatches: the default (256-pixel MX kernel where a launch fills the chip, bf16
    kernels elsewhere), every eligible shape on the 256-pixel MX kernel (PRG_W256_MIN_TILES=1), and every 3x3 conv on MX
    operands (PRG_MX_PURE=1: the simple halo-tile MX kernel for the rest).  All stay inside the reported MX drift bound
    against the reference's fp32 output; the all-MX variants bound the format's worst case."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden", "G13_unet_dim64_128.npz")
    tables = os.path.join(root, "tests", "golden", "G0_host_tables.npz")
    ref = golden("G13_unet_dim64_128")["y"]
    for name, env in {"default": {}, "w256mx_all": {"PRG_W256_MIN_TILES": "1"},
                      "pure": {"PRG_W256_MIN_TILES": "1", "PRG_MX_PURE": "1"}, "pure_simple": {"PRG_MX_PURE": "1", "PRG_CONV_W256MX": "0"}}.items():
        out = str(tmp_path / f"{name}.npy")
        r = subprocess.run([sys.executable, "-c", _MX_SCRIPT.format(root=root, gold=gold, tables=tables, out=out)],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        y = np.load(out)
        e, em = float(np.abs(y - ref).max()), float(np.abs(y - ref).mean())
        print(f"mxfp8 [{name}]: max {e:.3e} mean {em:.3e}")
        assert np.isfinite(y).all() and e <= MXFP8_MAX and em <= MXFP8_MEAN, (name, e, em)


def test_unet_mxfp8_drift_reported(hip, golden):
    """BASELINE configs[4] operand format end to end: dim-64 U-Net at 128x128 with MX-fp8 3x3 convolutions against the
    reference's fp32 output.  fp8 has a 3-bit mantissa: the drift is reported and bounded at <= 2x what is observed."""
    g = golden("G13_unet_dim64_128")
    sd = W.synth_state_dict(W.unet_config(64), 13)
    y = golden_unet(hip, golden, 64, "mxfp8", sd)(D(g["x"]), D(g["t"]), D(g["pc"]))
    yb = golden_unet(hip, golden, 64, "bf16", sd)(D(g["x"]), D(g["t"]), D(g["pc"]))
    e, em = maxerr(y, g["y"]), meanerr(y, g["y"])
    print(f"dim 64 @128 mxfp8: max {e:.3e} mean {em:.3e}   (bf16: max {maxerr(yb, g['y']):.3e} mean {meanerr(yb, g['y']):.3e})")
    assert np.isfinite(y.cpu().numpy()).all()
    assert e <= MXFP8_MAX and em <= MXFP8_MEAN


def test_mxfp8_at_256_configs4_forward(hip, golden):
    """BASELINE configs[4] in its own format: MX-fp8 3x3 convolutions at 256x256.  (a) the G16 fixture (B = 1: the small-launch
    dispatch); (b) a B = 16 batch — the launch shape bench.py's configs4 leg times, where the 256-pixel MX kernel runs —
    whose slot 0 is the fixture's input (reference = G16) and whose slot 15 is checked against the CPU oracle."""
    from oracle import unet as OU
    g = golden("G16_unet_dim64_256")
    sd = W.synth_state_dict(W.unet_config(64), 16)
    net = golden_unet(hip, golden, 64, "mxfp8", sd)
    y1 = net(D(g["x"]), D(g["t"]), D(g["pc"]))
    print(f"mxfp8 @256 B=1: max {maxerr(y1, g['y']):.3e} mean {meanerr(y1, g['y']):.3e}")
    assert maxerr(y1, g["y"]) <= MXFP8_MAX and meanerr(y1, g["y"]) <= MXFP8_MEAN
    gen = torch.Generator().manual_seed(256)
    x = torch.randn((16, 1, 256, 256), generator=gen)
    x[0] = torch.from_numpy(g["x"][0])
    t = torch.randint(0, 1000, (16,), generator=gen)
    t[0] = int(g["t"][0])
    pc = torch.from_numpy(g["pc"]).repeat(16, 1) + torch.randn((16, 4), generator=gen)
    pc[0] = torch.from_numpy(g["pc"][0])
    y = net(x.cuda(), t.cuda(), pc.cuda()).cpu()
    ref15 = OU.unet_forward(sd, x[15:16], t[15:16], pc[15:16])
    e0, m0 = maxerr(y[0:1], g["y"]), meanerr(y[0:1], g["y"])
    e15, m15 = maxerr(y[15:16], ref15.numpy()), meanerr(y[15:16], ref15.numpy())
    print(f"mxfp8 @256 B=16: slot 0 vs G16 max {e0:.3e} mean {m0:.3e}; slot 15 vs oracle max {e15:.3e} mean {m15:.3e}")
    assert np.isfinite(y.numpy()).all()
    assert max(e0, e15) <= MXFP8_MAX and max(m0, m15) <= MXFP8_MEAN


MX_CHAIN256_MAX_M, MX_CHAIN256_MEAN_M = 0.42, 0.08      # metres, in-painted pixels, <= 2x observed (0.206 / 0.038)


def test_mxfp8_ddim_chain_at_256_configs4(hip):
    """configs[4]'s sampler at its resolution: a DDIM chain (1000 -> 6 steps, eta = 1, DDNM replacement) at 256x256, B = 16,
    MX-fp8 operands, calibrated weights, stored noise, against the CPU oracle for two batch slots: known pixels bit-exact,
    in-painted drift reported in metres and bounded."""
    from oracle import diffusion as OD
    from oracle import unet as OU
    from pointreggpt_amd import synthetic
    B, S, steps = 16, 256, 6
    sd = W.synth_state_dict(W.unet_config(64), 256, calibrated=True)
    depth, K, pose = synthetic.synth_batch(256, range(B), S)
    Kd = D(K)
    rpj, hit = hip.G.reproject_tensor(D(depth), Kd, D(pose), clip=(0, 10), depth_unit=10.0, out_scale=0.1)
    _, _, cond = hip.G.apply_mask(torch.ones_like(rpj), rpj, hit, 0.5)
    pc = hip.G.param_vector(Kd)
    gen = torch.Generator().manual_seed(2560)
    noise = torch.randn((steps, B, 1, S, S), generator=gen)
    net = hip.Unet(64, dtype="mxfp8").load_state_dict(sd)
    d = hip.GaussianDiffusion(net, image_size=S, timesteps=1000, sampling_timesteps=steps)
    out = d.sample(param_cond=pc, img_cond=cond, noise=noise.cuda()).cpu()
    sch = OD.schedule(1000)
    den = lambda x, t, c: OU.unet_forward(sd, x, t, c)
    cond_h, pc_h = cond.cpu(), pc.cpu()
    worst_max = worst_mean = 0.0
    for b in (0, 15):
        ref = OD.sample(sch, den, pc_h[b:b + 1], cond_h[b:b + 1], S, OD.stored_noise(noise[:, b:b + 1]), sampling_steps=steps)
        known = OD.cond_mask(cond_h[b:b + 1])
        assert torch.equal(out[b:b + 1][known], ref[known])
        dd = (out[b:b + 1] - ref).abs()[~known] * 10.0
        print(f"mxfp8 DDIM-{steps} @256 slot {b}: in-painted depth vs oracle max {float(dd.max()):.3e} m mean {float(dd.mean()):.3e} m "
              f"(saturated {float(((ref <= 0) | (ref >= 1))[~known].float().mean()):.4f})")
        worst_max, worst_mean = max(worst_max, float(dd.max())), max(worst_mean, float(dd.mean()))
    assert worst_max > 0.0
    assert MX_CHAIN256_MAX_M is not None, "bounds not recorded yet"
    assert worst_max <= MX_CHAIN256_MAX_M and worst_mean <= MX_CHAIN256_MEAN_M


# ------------------------------------------------------------------------------------------------------------------
# CLI: generate_dataset.py + generate_gt.py keep the reference's command line and on-disk layout
# ------------------------------------------------------------------------------------------------------------------
def test_cli_generates_dataset_and_gt(tmp_path):
    import os
    import subprocess
    import sys
    from oracle import geometry as OG
    from pointreggpt_amd import postprocess as PP, synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    common = ["--dataset_name", "ds", "-start", "2", "-stop", "5"]
    cmd = [sys.executable, os.path.join(root, "generate_dataset.py"), "--resume", "synthetic:3", "--synthetic", "7",
           "--image_size", "64", "--sampling_timesteps", "4", "--batch_size", "2", "--dim", "16", "--dtype", "fp32",
           "--mask_threshold", "0.5"] + common
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    for idx in (2, 3, 4):
        d = tmp_path / "ds" / "data" / "scene-{:0>6d}".format(idx)
        for name in ("sample-000000.cloud.ply", "sample-000001.cloud.ply", "camera-intrinsics.txt", "sample-000001.pose.txt",
                     "sample-000000.image.png", "sample-000001.image.png", "sample-000001.depth.png",
                     "reprojected.image.png", "corrected.image.png"):
            assert (d / name).is_file(), name
        # sample 0 is the real (here: synthetic) frame: unproject -> crop box -> 0.025 voxel grid (sd:2479-2500)
        depth, K, _ = synthetic.synth_scene(7, idx, 64)
        assert np.allclose(np.loadtxt(d / "camera-intrinsics.txt"), K)
        ref = PP.voxel_down_sample(PP.crop_aabb(OG.point_cloud(depth * 10, K, (0.5, 10)).astype(np.float32)), 0.025)
        got = PP.read_ply(str(d / "sample-000000.cloud.ply"))
        assert got.shape == ref.shape and np.allclose(got, ref, atol=1e-12)
        gen = PP.read_ply(str(d / "sample-000001.cloud.ply"))
        assert len(gen) > 500 and np.isfinite(gen).all()
        pose_inv = np.loadtxt(d / "sample-000001.pose.txt")
        assert pose_inv.shape == (4, 4) and abs(np.linalg.det(pose_inv[:3, :3]) - 1) < 1e-5
        from PIL import Image
        d16 = np.asarray(Image.open(d / "sample-000001.depth.png"))
        assert d16.dtype == np.uint16 and d16.shape == (64, 64) and d16.max() <= 10000
    # resume: a second run skips finished batches (sd:2371-2381)
    out2 = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0 and "Skip completed scene" in out2.stdout
    gt = subprocess.run([sys.executable, os.path.join(root, "generate_gt.py"), "--disable_tqdm"] + common, cwd=tmp_path,
                        env=env, capture_output=True, text=True, timeout=600)
    assert gt.returncode == 0, gt.stderr[-2000:]
    lines = open(tmp_path / "ds" / "metadata" / "gt.log").read().splitlines()
    assert 1 <= len(lines) <= 3
    for line in lines:
        name, s, t, o1, o2 = line.split("\t")
        assert name.startswith("scene-") and (int(s), int(t)) == (0, 1) and 0 <= float(o1) <= 1 and 0 <= float(o2) <= 1


def _files_of(root):
    import os
    out = {}
    for d, _s, fs in os.walk(root):
        for f in fs:
            out[os.path.relpath(os.path.join(d, f), root)] = open(os.path.join(d, f), "rb").read()
    return out


def test_two_ranks_produce_the_single_rank_files_byte_for_byte(tmp_path):
    """BASELINE configs[3]'s mechanism on one device: `torchrun --nproc-per-node 2 generate_dataset.py` (both ranks on this
    GPU, scene range split in contiguous batch-aligned blocks, no collective) followed by generate_gt.py must leave exactly
    the files of a single-process run of the same range — every PLY / PNG / text file and both gt.log levels, byte for
    byte (sd:2369-2394, 2690; generate_gt.py:177-188)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--resume", "synthetic:5", "--synthetic", "9", "--image_size", "64", "--sampling_timesteps", "6", "--batch_size", "2",
            "--dim", "16", "--dtype", "bf16", "--mask_threshold", "0.5", "--dataset_name", "ds", "-start", "10", "-stop", "18"]
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    r = subprocess.run([sys.executable, os.path.join(root, "generate_dataset.py")] + args, cwd=one, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    port = 29500 + os.getpid() % 2000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "generate_dataset.py")] + args,
                       cwd=two, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "scenes [10, 14)" in r.stdout and "scenes [14, 18)" in r.stdout          # the two shards
    for d in (one, two):
        g = subprocess.run([sys.executable, os.path.join(root, "generate_gt.py"), "--dataset_name", "ds", "-start", "10", "-stop",
                            "18", "--disable_tqdm"], cwd=d, env=env, capture_output=True, text=True, timeout=600)
        assert g.returncode == 0, g.stderr[-2000:]
    f1, f2 = _files_of(one / "ds"), _files_of(two / "ds")
    assert sorted(f1) == sorted(f2) and len(f1) == 8 * 10 + 1            # 9 files + gt.log per scene, metadata/gt.log
    diff = [k for k in f1 if f1[k] != f2[k]]
    assert not diff, diff[:10]
    assert len(f1["metadata/gt.log"].splitlines()) >= 1
    # ... and neither do they depend on the lanes inside a rank (the runs above used the default two): one lane, same bytes
    lane1 = tmp_path / "lane1"
    lane1.mkdir()
    r = subprocess.run([sys.executable, os.path.join(root, "generate_dataset.py"), "--streams", "1"] + args, cwd=lane1, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    f3 = _files_of(lane1 / "ds")
    data = [k for k in f1 if k.startswith("data/") and not k.endswith("gt.log")]
    assert len(data) == 72 and not [k for k in data if f1[k] != f3[k]]


def test_real_data_path_end_to_end(tmp_path):
    """SURVEY 8(f) row 3 through the GPU once: a fake 3DMatch tree (640x480 uint16 depth PNGs, camera-intrinsics.txt,
    <cloud>.info.txt, train_info.pkl), a diffusion checkpoint with `ema_model.`-prefixed keys next to decoy online weights
    and a depth-correction checkpoint, then `generate_dataset.py --resume 1` WITHOUT --synthetic (sd:2307-2324, 2352-2361,
    2397-2500).  Sample 0 must be the oracle's unprojection of the independently decoded frame; the EMA weights (not the
    online ones) must be the ones sampled from (sd:2572); poses come from the seeded numpy stream of random_sample_pose."""
    import os
    import pickle
    import subprocess
    import sys
    from PIL import Image
    from oracle import geometry as OG
    from pointreggpt_amd import geometry as G, postprocess as PP
    from pointreggpt_amd.sharding import batch_pose_seed
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    S, dim = 64, 16
    rng = np.random.default_rng(33)
    data_root = tmp_path / "3dmatch"
    Kraw = np.array([[585.0, 0, 320.0], [0, 585.0, 240.0], [0, 0, 1.0]])
    raws, rels = [], []
    for i, (scene, seq, f0) in enumerate((("sceneA", "seq-01", 12), ("sceneB", "seq-02", 7))):
        yy, xx = np.mgrid[0:480, 0:640]
        raw = (2500 + 2 * (xx - 320) * (0.5 - i) + 1.2 * (yy - 240) + rng.integers(0, 40, size=(480, 640))).astype(np.uint16)   # mm
        raw[rng.random((480, 640)) < 0.03] = 0
        raw[200:230, 300:360] = 12000                      # > 10 m: dropped by the (> 1 -> 0) rule
        (data_root / scene / seq).mkdir(parents=True)
        Image.fromarray(raw).save(data_root / scene / seq / "frame-{:0>6d}.depth.png".format(f0))
        np.savetxt(data_root / scene / "camera-intrinsics.txt", Kraw)
        (tmp_path / "dataset/indoor/data/train" / scene).mkdir(parents=True)
        (tmp_path / "dataset/indoor/data/train" / scene / f"cloud_bin_{i}.info.txt").write_text(f"{scene} {seq} {f0} {f0 + 50}\n")
        raws.append(raw)
        rels.append(f"train/{scene}/cloud_bin_{i}.pth")
    (tmp_path / "dataset/indoor/metadata").mkdir(parents=True)
    with open(tmp_path / "dataset/indoor/metadata/train_info.pkl", "wb") as f:
        pickle.dump({"src": rels, "tgt": rels[::-1], "rot": [np.eye(3)] * 2, "trans": [np.zeros((3, 1))] * 2, "overlap": [0.5, 0.5]}, f)
    ema_sd = W.synth_state_dict(W.unet_config(dim), 41, calibrated=True)
    online_sd = W.synth_state_dict(W.unet_config(dim), 42)          # decoy: must NOT be the weights sampled from
    (tmp_path / "successive_ddnm_diffusion_results").mkdir()
    ckpt = {"step": 1, "model": {"model." + k: v for k, v in online_sd.items()},
            "ema": dict({"ema_model.model." + k: v for k, v in ema_sd.items()}, initted=torch.tensor([True]), step=torch.tensor([1]),
                        **{"ema_model.betas": torch.zeros(1000), "online_model.model.init_conv.bias": online_sd["init_conv.bias"]}),
            "opt": {}, "scaler": None}
    ckpt["model"]["betas"] = torch.zeros(1000)
    torch.save(ckpt, tmp_path / "successive_ddnm_diffusion_results" / "model-1.pt")
    mask_sd = W.synth_state_dict(W.maskunet_config(dim), 43, final_bias=8.0)
    (tmp_path / "depth_correction_results").mkdir()
    torch.save({"epoch": 3, "model": mask_sd, "opt": {}, "metrics": {}}, tmp_path / "depth_correction_results" / "model-best.pt")
    env = dict(os.environ, PYTHONPATH=root)
    cmd = [sys.executable, os.path.join(root, "generate_dataset.py"), "--resume", "1", "--data_root", str(data_root), "--image_size",
           str(S), "--sampling_timesteps", "4", "--batch_size", "2", "--dim", str(dim), "--dtype", "fp32", "--mask_threshold", "0.5",
           "--noise_seed", "77", "--dataset_name", "real", "-start", "0", "-stop", "2"]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # independent decode of the source frames (torchvision's documented arithmetic: short side -> S, nearest source pixel
    # floor((i + 0.5) * scale), centre crop at int(round((n - S) / 2)))
    nw = int(S * 640 / 480)
    xs = np.floor((np.arange(nw) + 0.5) * 640 / nw).astype(int)
    ys = np.floor((np.arange(S) + 0.5) * 480 / S).astype(int)
    left = int(round((nw - S) / 2.0))
    K = OG.intrinsic_transform(Kraw, S, S).astype(np.float32)
    np.random.seed(batch_pose_seed(77, 0, 0))
    poses = OG.random_sample_pose(2).astype(np.float32)
    depths = []
    for i in range(2):
        d = tmp_path / "real" / "data" / "scene-{:0>6d}".format(i)
        depth = raws[i][ys][:, xs][:, left:left + S].astype(np.float32) * np.float32(1e-4)
        depth[depth > 1] = 0
        depths.append(depth)
        assert np.allclose(np.loadtxt(d / "camera-intrinsics.txt"), K)
        cloud0 = OG.point_cloud(depth * 10, K, (0.5, 10.0))
        ref0 = PP.voxel_down_sample(PP.crop_aabb(cloud0.astype(np.float32)), 0.025)
        got0 = PP.read_ply(str(d / "sample-000000.cloud.ply"))
        assert got0.shape == ref0.shape and len(got0) > 300 and np.allclose(got0, ref0, atol=1e-12)
        assert np.allclose(np.loadtxt(d / "sample-000001.pose.txt"), np.linalg.inv(poses[i]), atol=1e-6)
        gen = PP.read_ply(str(d / "sample-000001.cloud.ply"))
        assert len(gen) > 100 and np.isfinite(gen).all()
    # the generated depth equals the oracle's chain on the same inputs with the EMA weights (and not with the decoy ones):
    # rebuild the condition with the product's geometry, then sample with the same Philox keys through the library directly
    from pointreggpt_amd import synthetic
    from pointreggpt_amd.diffusion import GaussianDiffusion
    from pointreggpt_amd.unet import MaskUnet, Unet
    mem = [PP.crop_aabb(OG.point_cloud(depths[i] * 10, K, (0.5, 10.0)).astype(np.float32)).astype(np.float32) for i in range(2)]
    rpj, hit = G.project_clouds(mem, poses, np.stack([K, K]), S, "cuda", depth_scale=0.1)
    mask = MaskUnet(dim, dtype="fp32").load_state_dict(mask_sd)
    _, _, cond = G.apply_mask(mask(rpj), rpj, hit, 0.5)
    seeds = [synthetic.noise_seed(77, i, 0) for i in range(2)]
    imgs = {}
    for tag, sdict in (("ema", ema_sd), ("online", online_sd)):
        net = Unet(dim, dtype="fp32").load_state_dict(sdict)
        diff = GaussianDiffusion(net, image_size=S, timesteps=1000, sampling_timesteps=4)
        img = diff.sample(param_cond=G.param_vector(torch.from_numpy(np.stack([K, K])).cuda()), img_cond=cond, seeds=seeds)
        imgs[tag], _, _ = G.apply_mask(mask(img), img, None, 0.5, want_cond=False)
        diff.close(); net.close()
    for i in range(2):
        d16 = np.asarray(Image.open(tmp_path / "real" / "data" / "scene-{:0>6d}".format(i) / "sample-000001.depth.png")).astype(np.int64)
        e_ema = np.abs(d16 - np.round(imgs["ema"][i, 0].cpu().numpy().astype(np.float64) * 1e4)).max()
        e_onl = np.abs(d16 - np.round(imgs["online"][i, 0].cpu().numpy().astype(np.float64) * 1e4)).max()
        assert e_ema <= 1 and e_onl > 100, (e_ema, e_onl)


def test_bench_contract_line(tmp_path):
    """bench.py's contract with the driver on a small workload: ONE JSON line on stdout with the fixed keys, the roofline object of
    the dominant kernel class measured live, two lanes, the file leg and the configs[4] leg."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8",
                        "--size", "64", "--sampling-steps", "6", "--profile-transitions", "3", "--e2e-batches", "2", "--c4-batch", "2",
                        "--c4-steps", "1", "--no-cpu-baseline", "--no-drift", "--parity-transitions", "4"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 4096, len(lines[0])          # the driver parses a bounded stdout tail (round 5: a 22 KB line was lost)
    c = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "e2e_files", "configs4", "parity_mode", "full"):
        assert k in c, k
    assert c["n_gpus"] == 1 and c["steps"] == 2 and c["warmup"] == 1 and c["higher_is_better"] is True and c["scaling"] == "weak"
    assert c["vs_baseline"] is None and c["dtype"] == "bf16" and c["data"] == "synthetic" and c["unit"] == "pairs/s"
    assert "workload" in c["config"] and "model" not in c["config"] and c["config"]["streams"] == 2
    assert abs(c["value"] - 2 * 8 / (c["ms_per_step"] * 2 / 1e3)) < 1e-6 * c["value"]
    rf = c["roofline"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "launches", "avg_launch_us",
              "executed_frac"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["launches"] > 0
    for k in ("fp32", "f16x3", "f16x3_256"):
        assert c["parity_mode"][k]["pairs_per_s"] > 0 and 0 < c["parity_mode"][k]["frac"] < 1, c["parity_mode"]
    assert c["configs4"]["dtype"] == "mxfp8" and c["configs4"]["peak"] == 5000.0 and 0 <= c["configs4"]["mx_flop_fraction"] <= 1   # (0 at this tiny batch: the MX kernel takes launches that fill the chip)
    assert c["configs4"]["bf16_same_shape"] > 0 and c["e2e_files"]["value"] > 0
    # the sidecar holds everything else (per-kernel table, legs in full, prose)
    j = json.load(open(os.path.join(root, c["full"])))
    for k in ("metric", "value", "config", "roofline", "e2e_files", "configs4", "workload", "parity_mode", "roofline_mem"):
        assert k in j, k
    assert j["value"] == c["value"] and j["roofline"]["frac"] == c["roofline"]["frac"]
    pm = j["parity_mode"]
    assert pm["fp32"]["dtype"] == "fp32" and pm["f16x3"]["dtype"] == "f16x3" and pm["fp32"]["timed_transitions"] == 4
    assert pm["fp32"]["pairs_per_s"] > 0 and pm["f16x3"]["roofline"]["peak"] == 2500.0 / 3 and pm["fp32"]["roofline"]["peak"] == 157.3
    assert len(j["roofline"]["per_kernel"]) > 0
    assert j["e2e_files"]["files_written"] == 9 * 8 * (2 + 2) and j["e2e_files"]["gt_log"]["lines"] >= 0
    assert j["configs4"]["dtype"] == "mxfp8" and j["configs4"]["config"]["image_size"] == 256 and j["configs4"]["roofline"]["peak"] == 5000.0


def test_bench_gpus_flag_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher starts two ranks itself (round-3 VERDICT: the flag used to be parsed and
    ignored).  One device here, so the gloo rehearsal backend lets the ranks share it: exactly one JSON line, n_gpus == 2,
    value = pairs of both ranks / max-over-ranks time.  Without the rehearsal backend the same command must refuse loudly."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "8", "--size", "64",
           "--sampling-steps", "4", "--no-roofline", "--no-e2e-files", "--no-drift", "--no-configs4", "--no-cpu-baseline", "--no-parity-mode"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=900, env=dict(env, PRG_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 4096, len(lines[0])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["scaling"] == "weak"
    assert abs(j["value"] - 2 * 8 / (j["ms_per_step"] / 1e3)) < 1e-6 * j["value"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and "HIP device" in r.stderr


def test_maskunet_stem_on_mfma(hip, golden):
    """Round 5: MaskUnet's 7x7 stem over the three DepthAugment planes runs on the MFMA stem kernel in bf16 mode
    (stem_mfma_kernel<3>: hi / lo split inputs and weights, ~16 significant bits) instead of the direct VALU kernel (541 us per
    call, 3.9e7 LDS bank conflicts).  Its output tensor against the float32 handle's (the direct fmaf chain, bit-identical to
    torch): the only difference allowed is the bf16 rounding of the stored result."""
    g = golden("G15_maskunet_dim64_128")
    sd = W.synth_state_dict(W.maskunet_config(64), 15, final_bias=6.0)
    x = D(g["depth"])
    B = x.shape[0]
    taps = {}
    for dt in ("fp32", "bf16"):
        net = hip.MaskUnet(64, dtype=dt).load_state_dict(sd)
        net.set_taps(True)
        net(x)
        taps[dt] = net.get_tap("init_conv", B).double().cpu()
        net.close()
    ref, got = taps["fp32"], taps["bf16"]
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-4                    # half an ulp of bf16 is 2^-9 relative; the split GEMM itself carries ~2^-16
    print(f"MaskUnet stem on MFMA: max |bf16 - fp32| {float(err.max()):.3e} on |ref| <= {float(ref.abs().max()):.2f}")
    assert bool((err <= tol).all()), float((err - tol).max())


def test_bench_eight_ranks_rehearsal_on_one_device(tmp_path):
    """Round 5: the 8-rank launch the driver runs on an 8-GPU node, rehearsed on the one device with
    the gloo backend at a tiny shape: eight ranks, ONE JSON line, n_gpus == 8, value = 8 ranks' pairs / max-over-ranks time, a
    per-rank table (setup / warm-up / timed seconds, device memory in use, CPU placement) and the device memory of 8 x 2 lanes of
    workspaces reported — so a slow barrier is attributable and the memory of a full node's worth of handles is known to fit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--batch", "4", "--size", "64",
           "--sampling-steps", "3", "--no-roofline", "--no-e2e-files", "--no-drift", "--no-configs4", "--no-cpu-baseline", "--no-parity-mode"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PRG_NO_AFFINITY")}
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=1500, env=dict(env, PRG_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) < 4096, len(lines[0])
    c = json.loads(lines[0])
    assert c["n_gpus"] == 8 and c["scaling"] == "weak" and c["config"]["streams"] == 2
    assert abs(c["value"] - 8 * 4 / (c["ms_per_step"] / 1e3)) < 1e-6 * c["value"]
    j = json.load(open(os.path.join(root, c["full"])))          # the per-rank table lives in the sidecar
    assert j["value"] == c["value"] and j["n_gpus"] == 8
    pr = j["per_rank"]
    assert [p["rank"] for p in pr] == list(range(8))
    assert all(p["setup_s"] > 0 and p["timed_s"] > 0 and p["timed_s"] <= j["ms_per_step"] / 1e3 * 1.001 for p in pr)
    mem = j["device_memory"]
    assert 0 < mem["in_use_bytes"] < mem["total_bytes"]
    print(f"8 ranks x 2 lanes on one device: {mem['in_use_bytes'] / 2**30:.2f} GiB in use of {mem['total_bytes'] / 2**30:.0f}; "
          f"setup {max(p['setup_s'] for p in pr):.1f} s max over ranks; affinity {j['cpu_affinity']}")
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:                                          # every rank pinned to its own disjoint share of the host
        assert all(p["pinned_cpus"] >= 1 for p in pr) and sum(p["pinned_cpus"] for p in pr) <= ncpu


# ------------------------------------------------------------------------------------------------------------------
# round 4: --num_samples > 1 (sd:2525-2680), NaN visibility of the accumulator statistics, 32-bit guard of the c64 kernel
# ------------------------------------------------------------------------------------------------------------------
def test_num_samples_3_against_oracle_sequence(hip, tmp_path):
    """`--num_samples 3` (generate_dataset.py:26-30): three views are generated from a growing scene memory — every view's
    fragment is unprojected into the common frame, the fragments are concatenated, the memory cloud is re-voxelised at 2 mm
    between views, and the final generated cloud is cropped / voxelised in the first view's frame (sd:2525-2680).  The
    sampler's inputs and outputs are recorded through a spy around `GaussianDiffusion.sample`; EVERYTHING AROUND the sampler
    is replayed with the oracle (z-buffer projection of the memory cloud, condition, float64 unprojection, inverse pose,
    dictionary voxel grid) and compared: conditions bit-exact, the final cloud as a point set."""
    from oracle import geometry as OG
    from oracle import postprocess as OP
    from pointreggpt_amd import postprocess as PP, synthetic
    from pointreggpt_amd.generator import Generator
    S, B, NSMP, seed = 64, 2, 3, 11
    net = hip.Unet(16, dtype="fp32").init_synthetic(3)
    mask = hip.MaskUnet(16, dtype="fp32").init_synthetic(4, final_bias=8.0)       # keep-probability ~1 everywhere: no threshold flips
    diff = hip.GaussianDiffusion(net, image_size=S, timesteps=1000, sampling_timesteps=4)
    calls = []
    real_sample = diff.sample

    def spy(**kw):
        out = real_sample(**kw)
        calls.append((kw["img_cond"].detach().cpu().clone(), out.detach().cpu().clone()))
        return out

    diff.sample = spy
    gen = Generator(diff, None, batch_size=B, samples_folder=str(tmp_path / "ds" / "data"), synthetic_seed=seed)
    gen.generate(0, B, NSMP, depth_correction=mask, mask_threshold=0.5, noise_seed=seed)
    torch.cuda.synchronize()
    assert len(calls) == NSMP
    idxs = list(range(B))
    poses = [gen._poses(idxs, s) for s in range(NSMP)]
    for j in idxs:
        d = tmp_path / "ds" / "data" / "scene-{:0>6d}".format(j)
        for s in range(1, NSMP + 1):
            for name in ("sample-{:0>6d}.pose.txt", "sample-{:0>6d}.image.png", "sample-{:0>6d}.depth.png"):
                assert (d / name.format(s)).is_file(), name.format(s)
            assert np.allclose(np.loadtxt(d / "sample-{:0>6d}.pose.txt".format(s)), np.linalg.inv(poses[s - 1][j]), atol=1e-6)
        assert (d / "sample-000001.cloud.ply").is_file() and not (d / "sample-000002.cloud.ply").exists()   # ONE generated cloud (sd:2652-2658)
        depth, K, _ = synthetic.synth_scene(seed, j, S)
        memory = PP.crop_aabb(OG.point_cloud(depth * 10, K, (0.5, 10)).astype(np.float32)).astype(np.float32)
        frags = None
        for s in range(NSMP):
            cond, img = calls[s]
            d_o, hit_o = OG.project_cloud(memory, poses[s][j], K, S)             # z-buffer of the moved memory cloud (sd:2531-2552)
            want = torch.cat([d_o * 0.1, hit_o.float()], dim=1) * 2 - 1
            assert torch.equal(cond[j:j + 1], want), (j, s, float((cond[j:j + 1] - want).abs().max()))
            pc = OG.inverse_pose_apply(OG.point_cloud(img[j, 0].numpy() * 10, K, (0.5, 10)), poses[s][j])   # float64, common frame
            frags = pc if frags is None else np.concatenate([frags, pc], axis=0)
            if s < NSMP - 1:                                                      # memory update at 2 mm (sd:2661-2680)
                memory = OP.voxel_down_sample(np.concatenate([memory.astype(np.float64), pc], axis=0), 0.002).astype(np.float32)
        T = poses[0][j].astype(np.float64)
        moved = frags @ T[:3, :3].T + T[:3, 3]
        ref = OP.voxel_down_sample(PP.crop_aabb(moved), 0.025)
        Ti = np.linalg.inv(T)                                                     # back by inv(T) (sd:2649)
        ref = ref @ Ti[:3, :3].T + Ti[:3, 3]
        got = PP.read_ply(str(d / "sample-000001.cloud.ply"))
        assert got.shape == ref.shape, (got.shape, ref.shape)
        from scipy.spatial import cKDTree                                         # the voxel order is unspecified: compare as point sets
        dist, nn = cKDTree(got).query(ref, k=1)
        assert float(dist.max()) <= 1e-9 and len(np.unique(nn)) == len(ref), float(dist.max())
    diff.close(); net.close(); mask.close()


def test_nonfinite_activations_stay_visible_bf16(hip):
    """the fixed-point GroupNorm accumulators must not turn a NaN / Inf activation into finite, wrong
    statistics.  An image with an Inf pixel yields a non-finite output; its batch neighbour is untouched (bit for bit)."""
    S, B = 64, 2
    net = hip.Unet(64, dtype="bf16").init_synthetic(5, calibrated=True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((B, 1, S, S), generator=g)
    t = torch.tensor([500, 500])
    pc = torch.tensor([[75.0, 76.0, 32.5, 32.0]] * B)
    clean = net(D(x.numpy()), D(t.numpy()), D(pc.numpy())).cpu()
    assert bool(torch.isfinite(clean).all())
    x_bad = x.clone()
    x_bad[0, 0, 10, 10] = float("inf")
    y = net(D(x_bad.numpy()), D(t.numpy()), D(pc.numpy())).cpu()
    assert not bool(torch.isfinite(y[0]).all()), "an Inf input pixel vanished inside the GroupNorm statistics"
    assert torch.equal(y[1], clean[1])
    net.close()


def test_c64_kernel_refuses_inputs_beyond_its_32_bit_offsets(hip):
    """conv3x3_c64 addresses its input through one buffer descriptor with 32-bit byte offsets.  B = 256 at
    256x256 x 64 channels is 2 GiB of bf16: the dispatch must hand that shape to a kernel with 64-bit addressing.  The last
    image (the bytes beyond 2 GiB) is compared with a float64 convolution of the bf16-rounded operands."""
    B, C, H = 256, 64, 256
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((B, C, H, H), generator=g, device="cuda")
    gw = torch.Generator().manual_seed(2)
    w = torch.randn((C, C, 3, 3), generator=gw) / 24.0
    bias = torch.randn((C,), generator=gw)
    import ctypes as C_
    lib = hip.lib.load()
    out = torch.empty((B, C, H, H), dtype=torch.float32, device="cuda")
    wh, bh = np.ascontiguousarray(w.numpy()), np.ascontiguousarray(bias.numpy())
    hip.lib.check(lib.prg_debug_conv3x3(hip.lib.ptr(x), wh.ctypes.data_as(C_.c_void_p), bh.ctypes.data_as(C_.c_void_p), hip.lib.ptr(out),
                                        B, C, C, H, H, hip.lib.PRG_BF16, hip.lib.stream_ptr()), "prg_debug_conv3x3")
    for b in (0, B - 1):
        xb = x[b:b + 1].cpu().to(torch.bfloat16).double()
        ref = torch.nn.functional.conv2d(xb, w.to(torch.bfloat16).double(), bias.double(), padding=1)
        err = (out[b:b + 1].cpu().double() - ref).abs()
        tol = 2.0 ** -8 * ref.abs() + 1e-4 * float(ref.abs().max())
        assert bool((err <= tol).all()), (b, float((err - tol).max()))
