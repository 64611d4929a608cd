#!/usr/bin/env python3
"""End-to-end use of pafuse_amd the way the reference's evaluate() uses D3DP (main_h3wb.py:194-509), on synthetic
sequences (the H3WB npz files and the released checkpoint are not available offline).

    python examples/evaluate_synthetic.py --proposals 20 --timesteps 10 --sequences 2 --frames 120
    torchrun --standalone --nproc-per-node 8 examples/evaluate_synthetic.py --proposals 160     # hypothesis-sharded

Prints the per-step protocol lines in the reference's log format (values are meaningless with random weights).
"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from pafuse_amd import harness  # noqa: E402
from pafuse_amd import synthetic as gu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proposals", type=int, default=5)
    ap.add_argument("--timesteps", type=int, default=5)
    ap.add_argument("--sequences", type=int, default=2)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--checkpoint", default="", help="a reference pafuse_model.bin (optional)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    ge.build()
    model, _ = ge.make_model(a.proposals, a.timesteps, device=f"cuda:{local}")
    if a.checkpoint:
        harness.load_checkpoint(model, harness.read_checkpoint(a.checkpoint))
        model = model.to(f"cuda:{local}").eval()
    ds = SimpleNamespace(parts_joint_indices=gu.DATASET_PART_JOINTS, root_indices=gu.ROOT_INDICES,
                         parts_connection_indices=dict(gu.CONNECTION_INDICES))
    cam = torch.tensor([2.29, 2.287, 0.025, 0.029, -0.207, 0.247, -0.003, -0.0009, -0.001])   # normalised intrinsics
    torch.manual_seed(0)                      # identical noise on every rank (each keeps its hypothesis slice)
    g = torch.Generator().manual_seed(1)
    total, n, t0 = None, 0, time.time()
    for _ in range(a.sequences):
        seq_2d = torch.rand(a.frames, 134, 2, generator=g) * 2 - 1
        seq_3d = torch.randn(a.frames, 134, 3, generator=g) * 0.25 + torch.tensor([0.0, 0.0, 4.0])
        sums, cnt = harness.evaluate_sequence(model, ds, seq_2d, seq_3d, cam, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT)
        total = sums if total is None else {k: total[k] + v for k, v in sums.items()}
        n += cnt
    torch.cuda.synchronize()
    if int(os.environ.get("RANK", "0")) == 0:
        mm = harness.report(total, n)
        print("Test time augmentation:", True)
        for i in range(a.timesteps):
            print("step %d : Protocol #1 Error (MPJPE) J_Best: %f mm" % (i, mm["j_best"][i]))
            print("step %d : Protocol #1 Error (MPJPE) P_Best: %f mm" % (i, mm["p_best"][i]))
            print("step %d : Protocol #1 Error (MPJPE) P_Agg: %f mm" % (i, mm["p_agg"][i]))
            print("step %d : Protocol #1 Error (MPJPE) J_Agg: %f mm" % (i, mm["j_agg"][i]))
            print("-----------------> Part-Based Evaluation <-----------------")
            print("step %d : Protocol #1 Error (MPJPE) P_Best Part-Based: %f mm" % (i, mm["p_best_pb"][i]))
            print("step %d : Protocol #1 Error (MPJPE) P_Best Part-Based BODY: %f mm" % (i, mm["p_best_pb_body"][i]))
            print("step %d : Protocol #1 Error (MPJPE) P_Best Part-Based FACE: %f mm" % (i, mm["p_best_pb_face"][i]))
            print("step %d : Protocol #1 Error (MPJPE) P_Best Part-Based HANDS: %f mm"
                  % (i, (mm["p_best_pb_right_hand"][i] + mm["p_best_pb_left_hand"][i]) / 2.0))
            print("step %d : Protocol #1 Error (MPJPE) P_Agg Part-Based: %f mm" % (i, mm["p_agg_pb"][i]))
        clips = n // 27
        print(f"{clips} clips x {a.proposals} hypotheses x {a.timesteps} steps in {time.time() - t0:.2f} s "
              f"on {world} GPU(s)")


if __name__ == "__main__":
    main()
