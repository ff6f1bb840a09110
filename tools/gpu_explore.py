"""Exploratory parity report on a GPU box (prints error magnitudes; asserts nothing).  Not part of the product."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointreggpt_amd import geometry as G  # noqa: E402
from pointreggpt_amd import weights as W  # noqa: E402
from pointreggpt_amd.diffusion import GaussianDiffusion  # noqa: E402
from pointreggpt_amd.unet import MaskUnet, Unet  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
dev = "cuda"


def gl(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def err(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    return f"max {np.nanmax(d):.3e} mean {np.nanmean(d):.3e} (ref absmax {np.nanmax(np.abs(b)):.3g})"


def section(fn):
    print(f"\n=== {fn.__name__} ===", flush=True)
    try:
        t = time.time()
        fn()
        torch.cuda.synchronize()
        print(f"   [{time.time() - t:.2f}s]", flush=True)
    except Exception:
        traceback.print_exc()


def geometry():
    g = gl("G4_pc2depth")
    d, m = G.pc2depth_tensor(T(g["pc"]), T(g["valid"]), T(g["K"]), image_size=(64, 64))
    print("pc2depth equal:", np.array_equal(d.cpu().numpy(), g["depth"]), np.array_equal(m.cpu().numpy(), g["mask"]),
          "ndiff", int((d.cpu().numpy() != g["depth"]).sum()))
    d, m = G.pc2depth_tensor(T(g["pc"][:, :5000]), T(g["valid"][:, :5000]), T(g["K"]), image_size=(48, 80))
    print("pc2depth 48x80 equal:", np.array_equal(d.cpu().numpy(), g["depth_48x80"]), np.array_equal(m.cpu().numpy(), g["mask_48x80"]))
    g = gl("G5_G6_reproject_unproject")
    d, m = G.reproject_tensor(T(g["depth"]), T(g["K"]), T(g["pose"]), clip=(0, 10), depth_unit=10.0)
    print("reproject equal:", np.array_equal(d.cpu().numpy(), g["rpj_depth"]), np.array_equal(m.cpu().numpy(), g["rpj_mask"]),
          "ndiff", int((d.cpu().numpy() != g["rpj_depth"]).sum()))
    d, m = G.reproject_tensor(T(g["depth"]), T(g["K"]), T(g["pose"]), clip=(0.5, 10), depth_unit=10.0)
    print("reproject05 equal:", np.array_equal(d.cpu().numpy(), g["rpj05_depth"]), np.array_equal(m.cpu().numpy(), g["rpj05_mask"]))
    pc, ok = G.depth2pc_tensor(T(g["depth"]) * 10, T(g["K"]), clip=(0.5, 10))
    print("depth2pc equal:", np.array_equal(pc.cpu().numpy(), g["pc"], equal_nan=True), np.array_equal(ok.cpu().numpy(), g["pc_valid"]))
    pc, ok = G.depth2pc_tensor(T(g["depth"]) * 10, T(g["K"]), clip=(0, 10), invalid_num=0.0)
    print("depth2pc0 equal:", np.array_equal(pc.cpu().numpy(), g["pc0"]), np.array_equal(ok.cpu().numpy(), g["pc0_valid"]))
    cl = G.point_clouds(T(g["depth"]), T(g["K"]), None)
    cc = G.point_clouds(T(g["depth"]), T(g["K"]), T(g["pose"]))
    for b in range(3):
        print(f"cloud{b}: equal {np.array_equal(cl[b], g[f'cloud{b}'])} common-frame equal {np.array_equal(cc[b], g[f'cloud{b}_common'])}",
              err(cc[b], g[f"cloud{b}_common"]))
    clouds = [g[f"cloud{b}"].astype(np.float32) for b in range(3)]
    d, m = G.project_clouds(clouds, g["pose"], g["K"], 64, dev)
    for b in range(3):
        print(f"project_clouds{b} equal:", np.array_equal(d[b].cpu().numpy(), g[f"gen_depth{b}"]),
              np.array_equal(m[b].cpu().numpy(), g[f"gen_mask{b}"]), "ndiff", int((d[b].cpu().numpy() != g[f"gen_depth{b}"]).sum()))
    g = gl("G11_maskunet")
    print("augment equal:", np.array_equal(G.depth_augment(T(g["depth"])).cpu().numpy(), g["augment"]))
    dd, hh, cond = G.apply_mask(T(g["d16_prob"]), T(g["depth"]), T(g["hit"]), float(g["thr"]))
    print("apply_mask equal:", np.array_equal(dd.cpu().numpy(), g["corrected"]), np.array_equal(hh.cpu().numpy(), g["mask_out"]),
          np.array_equal(cond.cpu().numpy(), g["img_cond"]))


def unet_small():
    g = gl("G7_unet_small_taps")
    for dtype in ("fp32", "bf16"):
        for dim in (8, 16):
            try:
                sd = W.synth_state_dict(W.unet_config(dim), 7)
                net = Unet(dim, dtype=dtype).load_state_dict(sd)
                net.set_taps(True)
                y = net(T(g[f"d{dim}_x"]), T(g[f"d{dim}_t"]), T(g[f"d{dim}_pc"]))
                torch.cuda.synchronize()
                for k in ("init_conv", "down0_block0", "down0_attn", "down0_out", "mid_attn", "up0_out", "final_res"):
                    print(f"  {dtype} dim{dim} tap {k:14s}", err(net.get_tap(k, 2), g[f"d{dim}_tap_{k}"]))
                print(f"  {dtype} dim{dim} OUT", err(y, g[f"d{dim}_y"]))
            except Exception:
                traceback.print_exc()


def unet_full():
    g = gl("G8_unet_dim64")
    for dtype in ("fp32", "bf16"):
        sd = W.synth_state_dict(W.unet_config(64), 8)
        net = Unet(64, dtype=dtype).load_state_dict(sd)
        y = net(T(g["x"]), T(g["t"]), T(g["pc"]))
        print(f"  {dtype} dim64 OUT", err(y, g["y"]))
        if dtype == "bf16":
            sdr = W.round_to_bf16(sd)
            from oracle import unet as OU
            yr = OU.unet_forward(sdr, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["pc"]))
            print("  bf16 vs oracle-with-bf16-rounded-weights", err(y, yr.numpy()))


def maskunet():
    g = gl("G11_maskunet")
    for dtype in ("fp32", "bf16"):
        for dim in (8, 16):
            net = MaskUnet(dim, dtype=dtype).load_state_dict(W.synth_state_dict(W.maskunet_config(dim), 11, final_bias=4.0))
            print(f"  {dtype} dim{dim} prob", err(net(T(g["depth"])), g[f"d{dim}_prob"]))


def sampler():
    g = gl("G9_G10_sampler")
    sd = W.synth_state_dict(W.unet_config(16), 9)
    for dtype in ("fp32", "bf16"):
        net = Unet(16, dtype=dtype).load_state_dict(sd)
        d8 = GaussianDiffusion(net, image_size=32, timesteps=8)
        for graph in (False, True):
            out = d8.sample(param_cond=T(g["pc"]), img_cond=T(g["cond"]), noise=T(g["chain8_noise"]), use_graph=graph)
            print(f"  {dtype} chain8 graph={graph}", err(out, g["chain8_out"]))
        d5 = GaussianDiffusion(net, image_size=32, timesteps=1000, sampling_timesteps=5)
        out = d5.sample(param_cond=T(g["pc"]), img_cond=T(g["cond"]), noise=T(g["ddim5_noise"]))
        print(f"  {dtype} ddim5", err(out, g["ddim5_out"]))
        known = (g["cond"][:, 1:2] + 1) * 0.5 > 0.5
        print("     known pixels exact:", np.array_equal(out.cpu().numpy()[known], g["ddim5_out"][known]))
        out = d5.sample(param_cond=T(g["pc"]), img_cond=None, noise=T(g["ddim5_nocond_noise"]))
        print(f"  {dtype} ddim5 nocond", err(out, g["ddim5_nocond_out"]))
        # philox
        o1 = d5.sample(param_cond=T(g["pc"]), img_cond=T(g["cond"]), seeds=[11, 22])
        o2 = d5.sample(param_cond=T(g["pc"]), img_cond=T(g["cond"]), seeds=[11, 22])
        print("     philox deterministic:", torch.equal(o1, o2), "range", o1.min().item(), o1.max().item())


def end_to_end():
    g = gl("G12_end_to_end_64")
    from pointreggpt_amd import geometry as G
    for dtype in ("fp32", "bf16"):
        unet = Unet(64, dtype=dtype).load_state_dict(W.synth_state_dict(W.unet_config(64), 12))
        mask = MaskUnet(64, dtype=dtype).load_state_dict(W.synth_state_dict(W.maskunet_config(64), 13, final_bias=6.0))
        diff = GaussianDiffusion(unet, image_size=64, timesteps=1000, sampling_timesteps=50)
        K, pose = T(g["K"]), T(g["pose"])
        rpj, hit = G.reproject_tensor(T(g["depth"]), K, pose, clip=(0, 10), depth_unit=10.0, out_scale=0.1)
        print(f"  {dtype} rpj equal", np.array_equal(rpj.cpu().numpy(), g["rpj_depth"]))
        prob1 = mask(rpj)
        print(f"  {dtype} prob1", err(prob1, g["prob1"]))
        # feed the golden condition so the chain comparison is isolated from threshold flips
        t0 = time.time()
        img = diff.sample(param_cond=G.param_vector(K), img_cond=T(g["img_cond"]), noise=T(g["noise"]))
        torch.cuda.synchronize()
        print(f"  {dtype} sampled (50-step ddim)", err(img, g["sampled"]), f"{time.time() - t0:.2f}s")
        prob2 = mask(img)
        print(f"  {dtype} prob2", err(prob2, g["prob2"]))
        out, _, _ = G.apply_mask(T(g["prob2"]), img, None, float(g["thr2"]), want_cond=False)
        cloud = G.point_clouds(out, K, pose)[0]
        print(f"  {dtype} cloud n={len(cloud)} ref n={len(g['cloud'])}",
              err(cloud, g["cloud"]) if len(cloud) == len(g["cloud"]) else "count differs")


def perf():
    """first look at throughput: B=64, 128^2, a few ancestral steps."""
    B, S = 64, 128
    for dtype in ("bf16",):
        unet = Unet(64, dtype=dtype).load_state_dict(W.synth_state_dict(W.unet_config(64), 1))
        diff = GaussianDiffusion(unet, image_size=S, timesteps=1000, sampling_timesteps=20)
        pc = torch.tensor([[151.5, 152.1, 64.5, 64.0]] * B, device=dev)
        cond = torch.zeros((B, 2, S, S), device=dev) - 1
        for graph in (True, False):
            diff.sample(param_cond=pc, img_cond=cond, seeds=list(range(B)), use_graph=graph)
            torch.cuda.synchronize()
            t0 = time.time()
            diff.sample(param_cond=pc, img_cond=cond, seeds=list(range(B)), use_graph=graph)
            torch.cuda.synchronize()
            dt = time.time() - t0
            print(f"  {dtype} B={B} S={S} 20 steps graph={graph}: {dt * 1000 / 20:.2f} ms/step -> "
                  f"{58.976e9 * B / (dt / 20) / 1e12:.1f} TFLOP/s, est {B / (dt / 20 * 1000):.3f} pairs/s @1000 steps")
        diff.sample(param_cond=pc, img_cond=cond, seeds=list(range(B)), profile=True)
        p = diff.last_profile(B)
        print("  profile:", p, f"conv TF/s = {p['conv_flops'] / (p['conv_ms'] * 1e-3) / 1e12:.1f}, conv share = {p['conv_ms'] / p['total_ms']:.2f}")


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    which = sys.argv[1:] or ["geometry", "unet_small", "unet_full", "maskunet", "sampler", "end_to_end", "perf"]
    for name in which:
        section(globals()[name])
