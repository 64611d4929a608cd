#!/usr/bin/env python3
"""Bring-up check of the fused qkv + attention kernel's attention phases (development tool): one denoiser pass per part with the
fused kernel and with qkv GEMM + attn_kernel (fp32 matrix cores), both against an fp64 evaluation (oracle in double)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from oracle import d3dp_oracle as orc
from pafuse_amd import synthetic as gu
DEV = "cuda"
depth = int(os.environ.get("DEPTH", "8"))
model, sd = ge.make_model(2, 2, seed=77, depth=depth)
sd64 = {k: v.double() for k, v in sd.items()}
x2d, _ = gu.synthetic_inputs_2d(B=1)
g = torch.Generator().manual_seed(52)
x3d = torch.randn(1, 2, 27, 134, 3, generator=g).clamp(-1.1, 1.1)
t = torch.tensor([499])
model.precision = "f16x2"
for part, idx in orc.PART_JOINTS.items():
    pre = f"pose_estimator.{part}."
    truth = orc.mixste2_eval(sd64, pre, x2d[..., idx, :].double(), x3d[..., idx, :].double(), t, depth=depth) if depth != 8 else \
        orc.mixste2_eval(sd64, pre, x2d[..., idx, :].double(), x3d[..., idx, :].double(), t)
    m = model.pose_estimator[part]
    row = {"part": part, "depth": depth}
    outs = {}
    for tag, fuse in (("two kernels", False), ("fused", True), ("fused again", True)):
        m.fuse_qkv_attention = fuse
        outs[tag] = m(x2d[..., idx, :].to(DEV), x3d[..., idx, :].to(DEV), t.to(DEV)).cpu()
        e = (outs[tag].double() - truth).abs()
        row[tag] = {"mean": float(e.mean()), "max": float(e.max())}
    row["fused run twice identical"] = bool(torch.equal(outs["fused"], outs["fused again"]))
    print(json.dumps(row), flush=True)
