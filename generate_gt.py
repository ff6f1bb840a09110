#!/usr/bin/env python
"""generate_gt.py — same command line and gt.log format as the reference script (generate_gt.py:16-41, 105-188):

    python generate_gt.py --dataset_name=generated_dataset [-start 0] [-stop 1] [--num_samples 2] [--disable_tqdm]

per scene `scene_name\\tsrc_idx\\ttgt_idx\\toverlap_src\\toverlap_tgt` (4 decimals), then ./<dataset>/metadata/gt.log.
"""
import argparse


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--dataset_name", default="generated_dataset", type=str, help="", required=True)
    p.add_argument("--start_scene_index", "-start", default=0, type=int, help="scenes index to start")
    p.add_argument("--stop_scene_index", "-stop", default=1, type=int, help="scenes index to stop")
    p.add_argument("--num_samples", default=2, type=int, help="sample numbers for each scene")
    p.add_argument("--disable_tqdm", action="store_true", help="disable tqdm")
    p.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                   help="cpu = the numpy grid specification of the overlap test instead of the HIP all-pairs kernel "
                        "(BASELINE configs[0]: no GPU); explicit only")
    args = p.parse_args()
    from pointreggpt_amd.generator import gather_gt, generate_gt
    generate_gt(args.dataset_name, args.start_scene_index, args.stop_scene_index, args.num_samples,
                overlap="hip" if args.device == "cuda" else "numpy-spec")
    gather_gt(args.dataset_name, args.start_scene_index, args.stop_scene_index)


if __name__ == "__main__":
    main()
