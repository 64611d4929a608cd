#!/usr/bin/env python3
"""In-the-wild 2D->3D lifting, the flow of the reference's in_the_wild/h3wb_diffusion.py:28-145 without the video and
rendering parts: OpenPifPaf whole-body JSON (one line per frame) -> screen-normalised keypoints -> flip-TTA DDIM
sampling on the HIP path with ``input_3d=None`` -> whole-body poses stitched back to the video's frames.

    python examples/infer_in_the_wild.py --keypoints video.openpifpaf.json --width 1920 --height 1080 \
           --data data/train_h3wb.npz --checkpoint checkpoint/pafuse_model.bin --out outputs/video \
           ft2d.num_proposals=20 ft2d.sampling_timesteps=10

Writes <out>/test_3d_output.npy [T,P,frames,134,3] (camera space) and <out>/test_3d_output_postprocess.npy (rotated
with the reference's fixed camera orientation, lowest point at height 0).  Without --keypoints a synthetic detection
file is generated (random weights unless --checkpoint: numbers are meaningless).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import pafuse_amd  # noqa: E402
from pafuse_amd import config, h3wb, harness  # noqa: E402

CAMERA_ROTATION = np.array([0.14070565, -0.15007018, -0.7552408, 0.62232804], dtype=np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keypoints", default="")
    ap.add_argument("--width", type=int, default=1000)
    ap.add_argument("--height", type=int, default=1002)
    ap.add_argument("--data", default=os.path.join(ROOT, "tests", "golden", "h3wb_synth", "train_h3wb.npz"))
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "in_the_wild"))
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("overrides", nargs="*")
    a = ap.parse_args()
    args = config.load(None, a.overrides)
    os.makedirs(a.out, exist_ok=True)
    ge.build()
    if not a.keypoints:
        a.keypoints = os.path.join(a.out, "synthetic.openpifpaf.json")
        rng = np.random.default_rng(0)
        with open(a.keypoints, "w") as f:
            for _ in range(70):
                flat = np.concatenate([rng.uniform(0, a.width, (133, 1)), rng.uniform(0, a.height, (133, 1)),
                                       np.ones((133, 1))], axis=1).reshape(-1)
                f.write(json.dumps({"predictions": [{"keypoints": flat.tolist()}]}) + "\n")

    dataset = h3wb.Human3WBDataset(a.data)
    kps_left, kps_right = (list(x) for x in dataset.keypoints_metadata["keypoints_symmetry"])
    model = pafuse_amd.D3DP(args, kps_left, kps_right, dataset=dataset, is_train=False,
                            num_proposals=args.ft2d.num_proposals, sampling_timesteps=args.ft2d.sampling_timesteps)
    if a.checkpoint:
        harness.load_checkpoint(model, harness.read_checkpoint(a.checkpoint))
    model = model.cuda().eval()

    pixels = harness.load_pifpaf_keypoints(a.keypoints, args.data.num_kps)
    keypoints = h3wb.normalize_screen_coordinates(pixels[..., :2], w=a.width, h=a.height).astype(np.float32)
    pred = harness.infer_sequence(model, dataset, keypoints, kps_left, kps_right, batch_size=a.batch)
    poses = harness.stitch_clips(pred, keypoints.shape[0], model.frames)            # [T,P,frames,134,3]
    np.save(os.path.join(a.out, "test_3d_output.npy"), poses.numpy(), allow_pickle=True)
    world = harness.camera_to_world(poses, CAMERA_ROTATION)
    world[..., 2] -= world[..., 2].min()                                            # no trajectory: rebase the height
    np.save(os.path.join(a.out, "test_3d_output_postprocess.npy"), world.numpy(), allow_pickle=True)
    print(f"{keypoints.shape[0]} frames -> poses {tuple(poses.shape)} in {a.out}")


if __name__ == "__main__":
    main()
