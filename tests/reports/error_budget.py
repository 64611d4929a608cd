#!/usr/bin/env python3
"""Where the |HIP - oracle| difference comes from: both fp32 implementations against an fp64 evaluation of the
same function (same fp32 weights/inputs, arithmetic in double).  Run on the GPU box; prints JSON lines and, with
`--out FILE`, writes them as one document carrying the kernel-source digest (profiles/r03_error_budget.json: the numbers
behind "both fp32 implementations sit 2.1-2.6e-4 mm from exact arithmetic" in DESIGN.md section 4)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import make_model  # noqa: E402
from oracle import d3dp_oracle as orc  # noqa: E402
from tests.golden import golden_util as gu  # noqa: E402
from pafuse_amd import ops  # noqa: E402

out_path = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--out" else None
rows = []
model, sd = make_model(2, 2, seed=77)
sd64 = {k: v.double() for k, v in sd.items()}
x2d, _ = gu.synthetic_inputs_2d(B=1)
g = torch.Generator().manual_seed(52)
x3d = torch.randn(1, 2, 27, 134, 3, generator=g).clamp(-1.1, 1.1)
for tval in (999, 499, 99):
    t = torch.tensor([tval])
    for part, idx in orc.PART_JOINTS.items():
        pre = f"pose_estimator.{part}."
        C = gu.PART_WIDTH[part]
        te64 = orc.timestep_embedding(sd64, pre, t, C)
        te32 = orc.timestep_embedding(sd, pre, t, C)
        teh = ops.time_embed(model.pose_estimator[part], t.cuda()).cpu()
        o64 = orc.mixste2_eval(sd64, pre, x2d[..., idx, :].double(), x3d[..., idx, :].double(), t)
        o32 = orc.mixste2_eval(sd, pre, x2d[..., idx, :], x3d[..., idx, :], t)
        oh = model.pose_estimator[part](x2d[..., idx, :].cuda(), x3d[..., idx, :].cuda(), t.cuda()).cpu()
        e32, eh = (o32.double() - o64), (oh.double() - o64)
        rows.append({
            "t": tval, "part": part,
            "temb_err_oracle32": (te32.double() - te64).abs().max().item(),
            "temb_err_hip": (teh.double() - te64).abs().max().item(),
            "out_rms": o64.pow(2).mean().sqrt().item(),
            "oracle32_vs_fp64": {"max": e32.abs().max().item(), "mean_abs": e32.abs().mean().item(), "mean": e32.mean().item()},
            "hip_vs_fp64": {"max": eh.abs().max().item(), "mean_abs": eh.abs().mean().item(), "mean": eh.mean().item()},
            "hip_vs_oracle32": {"max": (oh - o32).abs().max().item(), "mean_abs": (oh - o32).abs().mean().item()},
        })
        print(json.dumps(rows[-1]), flush=True)
if out_path:
    from pafuse_amd._lib import kernel_source_digest
    mean = lambda key: sum(r[key]["mean_abs"] for r in rows) / len(rows)
    doc = {"what": "one denoiser pass (P=2, seeded weights 77) per part and timestep: the fp32 CPU oracle and the HIP path (precision "
                   f"{model.precision}) against an fp64 evaluation of the same function; metres",
           "device": torch.cuda.get_device_name(0), "kernel_source_sha256": kernel_source_digest(),
           "summary_mm": {"oracle32_vs_fp64_mean_abs": mean("oracle32_vs_fp64") * 1e3, "hip_vs_fp64_mean_abs": mean("hip_vs_fp64") * 1e3,
                          "hip_vs_oracle32_mean_abs": mean("hip_vs_oracle32") * 1e3},
           "rows": rows}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
