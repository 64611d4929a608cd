#!/usr/bin/env python3
"""The reference's `python main_h3wb.py general.evaluate=pafuse_model.bin ...` flow on real files (main_h3wb.py:568-760,
194-531): H3WB npz -> loader -> per-camera sequences of the test subjects -> D3DP eval model with the checkpoint ->
flip-TTA DDIM sampling on the HIP path -> J-Best / P-Best / P-Agg / J-Agg and the part-based protocols in mm.

    python examples/evaluate_h3wb.py --data data/train_h3wb.npz --checkpoint checkpoint/pafuse_model.bin \
           ft2d.num_proposals=20 ft2d.sampling_timesteps=10
    torchrun --standalone --nproc-per-node 8 examples/evaluate_h3wb.py ... ft2d.num_proposals=160   # hypothesis-sharded

Trailing `a.b=c` arguments override the config tree like the reference's Hydra command line (pafuse_amd/config.py).
Without `--data` the synthetic H3WB files of tests/golden/h3wb_synth are used (random weights: numbers are meaningless).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import pafuse_amd  # noqa: E402
from pafuse_amd import config, h3wb, harness  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=os.path.join(ROOT, "tests", "golden", "h3wb_synth", "train_h3wb.npz"))
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--config", default="", help="a reference config.yaml (optional)")
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args()
    args = config.load(a.config or None, a.overrides)
    world, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    ge.build()

    dataset = h3wb.Human3WBDataset(a.data)
    keypoints = h3wb.prepare_keypoints(dataset)
    kps_left, kps_right = (list(x) for x in dataset.keypoints_metadata["keypoints_symmetry"])
    joints_left, joints_right = list(dataset.skeleton().joints_left()), list(dataset.skeleton().joints_right())
    action_filter = None if args.data.actions == "*" else args.data.actions.split(",")
    cams, poses_3d, poses_2d = h3wb.fetch(args.data.subjects_test.split(","), keypoints, dataset,
                                          stride=args.experiment.downsample, action_filter=action_filter)

    model = pafuse_amd.D3DP(args, joints_left, joints_right, dataset=dataset, is_train=False,
                            num_proposals=args.ft2d.num_proposals, sampling_timesteps=args.ft2d.sampling_timesteps)
    print("INFO: Trainable parameter count:", sum(p.numel() for p in model.parameters()) / 1e6, "Million")
    if a.checkpoint:
        ckpt = harness.read_checkpoint(a.checkpoint)    # holds a pickled numpy RandomState: weights_only=False
        print("This model was trained for {} epochs".format(ckpt.get("epoch", "?")))
        harness.load_checkpoint(model, ckpt)
    model = model.to(f"cuda:{local}").eval()
    torch.manual_seed(0)            # every rank draws the same full-P noise and keeps its hypothesis slice
    log = print if int(os.environ.get("RANK", "0")) == 0 else (lambda *x: None)
    log("Test time augmentation:", args.model.test_time_augmentation)
    h3wb.evaluate(model, dataset, cams, poses_3d, poses_2d, kps_left, kps_right,
                  batch_size=args.model.batch_size, log=log)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
