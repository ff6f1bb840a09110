#!/usr/bin/env python
"""generate_dataset.py — same command line and output layout as the reference script (generate_dataset.py:1-62):

    python generate_dataset.py --resume=official [--dataset_name generated_dataset] [-start 0] [-stop 1] [--num_samples 1]

writes ./<dataset_name>/data/scene-XXXXXX/{sample-000000.cloud.ply, sample-000001.cloud.ply, camera-intrinsics.txt,
sample-*.pose.txt, *.png}.  The hot path runs on the MI355X HIP library.  Additive flags (defaults = the reference's
hard-coded literals, generate_dataset.py:32-55):
  --image_size 256  --timesteps 1000  --sampling_timesteps 250  --batch_size 4  --dim 64
  --dtype fp32|f16x3|bf16|mxfp8   f16x3 = float32 storage, split-f16 MFMA contractions: inside 1e-4 m of the reference on the committed reference chains (measured per chain, not guaranteed) at ~4x fp32's speed; fp32 (default: the reference runs with amp=False, generate_dataset.py:54) is the parity mode;
                            bf16 = BASELINE configs[1-3] throughput mode; mxfp8 = configs[4] (3x3 convs on block-scaled fp8 MFMA)
  --data_root /path/to/3DMatch-RGBD/train
  --streams 2         lanes per GPU: batches are dealt round-robin to N host threads / HIP streams with their own network
                      handles, so one lane's launch gaps are filled by the other's kernels (+7 % pairs/s); files are identical
  --device cpu        the prg_cpu_* twins (libprg_cpu.so: plain C++ / OpenMP, float32) instead of the HIP library: BASELINE
                      configs[0]'s plumbing run without a GPU; explicit only, the default never falls back to it
  --synthetic SEED    synthetic scenes instead of 3DMatch frames (no dataset needed)
  --noise_seed N      seed of the diffusion noise (default: fresh entropy, printed; synthetic runs default to SEED)
  --resume synthetic[:SEED]   deterministic synthetic weights instead of ./successive_ddnm_diffusion_results/model-<resume>.pt
Under `torchrun --nproc-per-node N` every rank takes a contiguous block of [-start, -stop) (no collectives).
"""
import argparse
import os

import torch


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--resume", default=None, type=str, help="checkpoint to load", required=True)
    p.add_argument("--dataset_name", default="generated_dataset", type=str, help="")
    p.add_argument("--start_scene_index", "-start", default=0, type=int, help="scenes index to start")
    p.add_argument("--stop_scene_index", "-stop", default=1, type=int, help="scenes index to stop")
    p.add_argument("--num_samples", default=1, type=int, help="sample numbers for each scene")
    p.add_argument("--image_size", default=256, type=int)
    p.add_argument("--timesteps", default=1000, type=int)
    p.add_argument("--sampling_timesteps", default=250, type=int)
    p.add_argument("--batch_size", default=4, type=int)
    p.add_argument("--dim", default=64, type=int)
    p.add_argument("--dtype", default="fp32", choices=["fp32", "f16x3", "bf16", "mxfp8"],
                   help="arithmetic of the two U-Nets: fp32 = parity mode (the reference's amp=False; the only mode LABELLED parity); "
                        "f16x3 = float32 storage, split-f16 MFMA contractions: within 1e-4 m point-XYZ of the reference on every "
                        "calibrated reference chain of tests/golden (64x64 .. 256x256; measured values: tests/test_gpu_f16x3.py, "
                        "bench.py drift_vs_reference) at ~4x fp32's throughput; bf16 / mxfp8 = throughput modes, centimetres away")
    p.add_argument("--streams", default=2, type=int,
                   help="concurrent lanes per GPU (own network handles + HIP stream + host thread each; batches dealt round-robin)")
    p.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                   help="cpu = the prg_cpu_* twins (plain C++ / OpenMP, float32): BASELINE configs[0]'s plumbing run without a "
                        "GPU.  Never chosen implicitly: the default fails loudly when there is no HIP device")
    p.add_argument("--data_root", default="/path/to/3DMatch-RGBD/train", type=str)
    p.add_argument("--synthetic", default=None, type=int, help="seed of the synthetic scene generator")
    p.add_argument("--mask_threshold", default=0.99, type=float)
    p.add_argument("--noise_seed", default=None, type=int,
                   help="seed of the diffusion noise (and, with real data, of the pose stream); default: fresh entropy, "
                        "logged, like the reference's unseeded torch.randn")
    args = p.parse_args()

    from pointreggpt_amd import sharding
    from pointreggpt_amd.generator import Generator
    from pointreggpt_amd.weights import maskunet_state_from_checkpoint
    if args.device == "cpu":
        if args.dtype != "fp32":
            p.error("--device cpu computes in float32: use --dtype fp32")
        from pointreggpt_amd.cpu import GaussianDiffusion, MaskUnet, Unet
        args.streams = 1
    else:
        from pointreggpt_amd.diffusion import GaussianDiffusion
        from pointreggpt_amd.unet import MaskUnet, Unet

    rank, world, local = sharding.rank_world()
    if args.device == "cuda":
        dev_index = local % max(1, torch.cuda.device_count())              # (more ranks than devices: ranks share, used by the tests)
        # before any thread exists (lane threads, the C++ writer pool): this rank's share of the cores of its GPU's NUMA node
        # (PRG_NO_AFFINITY=1 opts out; a single rank is left alone)
        aff = sharding.pin_rank_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), device_index=dev_index)
        if aff.get("pinned"):
            print("[rank {}/{}] pinned to {} CPUs [{}..{}] (NUMA node {}, {})".format(
                rank, world, aff["cpus"], aff["first_cpu"], aff["last_cpu"], aff["numa_node"], aff["source"]))
        torch.cuda.set_device(dev_index)
    start, stop = sharding.shard_range(args.start_scene_index, args.stop_scene_index, rank, world, args.batch_size)

    model = Unet(dim=args.dim, param_cond_dim=4, dim_mults=(1, 2, 4, 8), channels=1, dtype=args.dtype)
    depth_correction = MaskUnet(dim=args.dim, dim_mults=(1, 2, 4, 8), dtype=args.dtype)
    diffusion = GaussianDiffusion(model, image_size=args.image_size, timesteps=args.timesteps,
                                  sampling_timesteps=args.sampling_timesteps, loss_type="l1", objective="pred_x0",
                                  beta_schedule="sigmoid", ddim_sampling_eta=1.0, is_ddnm_sampling=True)
    generator = Generator(diffusion, args.data_root, batch_size=args.batch_size,
                          results_folder="./successive_ddnm_diffusion_results",
                          samples_folder="./{}/data".format(args.dataset_name), synthetic_seed=args.synthetic,
                          device=args.device)

    def load_weights(unet, mask):
        if args.resume.startswith("synthetic"):
            seed = int(args.resume.split(":")[1]) if ":" in args.resume else 0
            unet.init_synthetic(seed)
            mask.init_synthetic(seed + 1, final_bias=8.0)
        else:
            generator.load(args.resume, unet=unet)
            ckpt = torch.load("./depth_correction_results/model-best.pt", map_location="cpu")
            mask.load_state_dict(maskunet_state_from_checkpoint(ckpt, mask.cfg))

    load_weights(model, depth_correction)
    n_batches = (max(0, stop - start) + args.batch_size - 1) // args.batch_size
    lanes = []
    for _ in range(1, max(1, min(args.streams, n_batches))):
        u2 = Unet(dim=args.dim, param_cond_dim=4, dim_mults=(1, 2, 4, 8), channels=1, dtype=args.dtype)
        m2 = MaskUnet(dim=args.dim, dim_mults=(1, 2, 4, 8), dtype=args.dtype)
        load_weights(u2, m2)
        lanes.append((GaussianDiffusion(u2, image_size=args.image_size, timesteps=args.timesteps,
                                        sampling_timesteps=args.sampling_timesteps, loss_type="l1", objective="pred_x0",
                                        beta_schedule="sigmoid", ddim_sampling_eta=1.0, is_ddnm_sampling=True), m2))
    if args.noise_seed is None:
        # ONE seed for the whole job: rank 0 draws it (secrets.randbits) and publishes it to every rank through a c10d store on
        # MASTER_ADDR:MASTER_PORT (sharding.job_seed), so a multi-rank run is reproducible from the one logged value (--noise_seed /
        # PRG_JOB_SEED replay it) and a scene's noise key does not depend on the shard that produced it
        args.noise_seed = sharding.job_seed() if args.synthetic is None else int(args.synthetic)
    print("[rank {}/{}] noise seed: {}  scenes [{}, {})".format(rank, world, args.noise_seed, start, stop))
    if stop > start:
        generator.generate(start_scene_index=start, stop_scene_index=stop, num_samples=args.num_samples,
                           has_refine_step=False, depth_correction=depth_correction,
                           mask_threshold=args.mask_threshold, noise_seed=args.noise_seed, lanes=lanes)
    if args.device == "cuda":
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
