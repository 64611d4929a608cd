#!/usr/bin/env python3
"""Train-from-scratch flow of the reference's main_h3wb.py (:760-1040) on this library: H3WB npz -> shuffled,
flip-augmented 27-frame clips -> D3DP in train mode (q_sample targets, three per-part denoisers with DropPath; forward
and backward are HIP kernels) -> mpjpe loss -> AdamW (lr 6e-5, wd 0.1, x0.993 per epoch) -> checkpoints in the
reference's format (evaluate with examples/evaluate_h3wb.py --checkpoint ...).

    python examples/train_h3wb.py --data data/train_h3wb.npz --out checkpoint model.epochs=400
    torchrun --standalone --nproc-per-node 8 examples/train_h3wb.py --data data/train_h3wb.npz --out checkpoint
        (DistributedDataParallel over RCCL: every rank trains on its own shuffle of the clips, gradients averaged)

Without --data the synthetic H3WB files of tests/golden/h3wb_synth are used (smoke run: add data.subjects_train=S1,S5
model.epochs=2 model.batch_size=108, the files hold only those subjects).
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import pafuse_amd  # noqa: E402
from pafuse_amd import config, h3wb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=os.path.join(ROOT, "tests", "golden", "h3wb_synth", "train_h3wb.npz"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "train"))
    ap.add_argument("--config", default="")
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args()
    args = config.load(a.config or None, a.overrides)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    ge.build()
    os.makedirs(a.out, exist_ok=True)
    log = print if rank == 0 else (lambda *x: None)

    dataset = h3wb.Human3WBDataset(a.data)
    keypoints = h3wb.prepare_keypoints(dataset)
    kps_left, kps_right = (list(x) for x in dataset.keypoints_metadata["keypoints_symmetry"])
    joints_left, joints_right = list(dataset.skeleton().joints_left()), list(dataset.skeleton().joints_right())
    action_filter = None if args.data.actions == "*" else args.data.actions.split(",")
    cams, poses_3d, poses_2d = h3wb.fetch(args.data.subjects_train.split(","), keypoints, dataset,
                                          stride=args.experiment.downsample, action_filter=action_filter)
    generator = h3wb.ChunkedClips(args.model.batch_size // args.model.number_of_frames, cams, poses_3d, poses_2d,
                                  args.model.number_of_frames, shuffle=True, random_seed=1234,
                                  augment=args.model.data_augmentation, kps_left=kps_left, kps_right=kps_right,
                                  joints_left=joints_left, joints_right=joints_right,
                                  shard=(rank, world) if world > 1 else None)   # rank's slice of the global batch
    log("INFO: Training on {} frames".format(sum(p.shape[0] for p in poses_2d)))

    model = pafuse_amd.D3DP(args, joints_left, joints_right, dataset=dataset, is_train=True).to(dev).train()
    log("INFO: Trainable parameter count:", sum(p.numel() for p in model.parameters()) / 1e6, "Million")
    model.prepare_for_ddp()               # N > 1: 'f32' products beside RCCL's kernels (decided once, here; see D3DP.prepare_for_ddp)
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local]) if world > 1 else model
    lr = args.model.learning_rate
    optimizer = torch.optim.AdamW(net.parameters(), lr=lr, weight_decay=0.1)
    for epoch in range(args.model.epochs):
        t0 = time.time()
        loss = h3wb.train_epoch(net, optimizer, generator, dataset, dev, wb_loss=args.model.wb_loss, log=log,
                                mse_loss=args.model.mse_loss, weighted_loss=args.model.weighted_loss)
        log("[%d] time %.2f lr %f 3d_train %f" % (epoch + 1, (time.time() - t0) / 60, lr, loss * 1000))
        lr *= args.model.lr_decay
        for group in optimizer.param_groups:
            group["lr"] *= args.model.lr_decay
        if rank == 0 and ((epoch + 1) % args.general.checkpoint_frequency == 0 or epoch + 1 == args.model.epochs):
            print("Saving checkpoint to", h3wb.save_state(net, optimizer, epoch + 1, lr, a.out,
                                                           random_state=generator.random_state()))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
