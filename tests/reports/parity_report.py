#!/usr/bin/env python3
"""Print how far the HIP path is from the CPU oracle on seeded inputs (pointwise and MPJPE, fp64 metric math).

Run on the GPU box:  python tests/reports/parity_report.py [P T [B [f32|bf16]]]  ->  one JSON line
(bf16 = the opt-in bf16-operand mode: how far it is from the fp32 oracle).
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import make_model  # noqa: E402
from oracle import d3dp_oracle as orc  # noqa: E402
from tests.golden import golden_util as gu  # noqa: E402
from tests.test_hip_parity import _mpjpe_report  # noqa: E402

P, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 5)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
model, sd = make_model(P, T, seed=77)
model.precision = sys.argv[4] if len(sys.argv) > 4 else "f32"
x2d, x2f = gu.synthetic_inputs_2d(B=B)
noises = gu.synthetic_noises(B=B, P=P, n=T, seed=3)
model.noise_fn = lambda k, shape, device: noises[k]
out = model(x2d.cuda(), None, input_2d_flip=x2f.cuda()).cpu()
t0 = time.time()
ref = orc.ddim_sample(sd, x2d, noises, T, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
cpu_s = time.time() - t0
target = orc.center_pose_parts(gu.synthetic_target_3d(B))
got, want = _mpjpe_report(out, target, x2d), _mpjpe_report(ref, target, x2d)
d = (out - ref).abs()
print(json.dumps({
    "B": B, "P": P, "T": T, "precision": model.precision, "oracle_cpu_s": round(cpu_s, 2),
    "pointwise_max_abs": d.max().item(), "pointwise_mean_abs": d.mean().item(),
    "per_step_max_abs": [d[:, k].max().item() for k in range(T)],
    "clamped_frac": (ref.abs() >= 1.1).float().mean().item(),
    "mpjpe_mm_abs_diff": {k: (got[k] - want[k]).abs().max().item() for k in want},
    "mpjpe_mm_oracle_last_step": {k: want[k][-1].item() for k in want},
}))
