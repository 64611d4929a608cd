#!/usr/bin/env python3
"""Reproducer / detector for the multi-queue hazard of the bf16 matrix instruction (profiles/r02_bf16_mfma_concurrency.md).

Three independent copies of one body-part denoiser (own weights, own workspaces, own outputs) run at the same time on
three HIP streams, 80 times; every output is compared bit for bit with the single-stream result.  Nothing is shared
between the streams but the GPU.

    python tools/mfma_queue_concurrency.py f32|bf16x3|bf16 [depth]

Observed on MI355X (ROCm 7.2), failures out of 240 outputs:  f32 0;  bf16x3 50-80;  bf16 4-50.
Diagnostic builds (PAFUSE_HIP_LIB=...):  -DPAFUSE_MFMA_K8 (legacy v_mfma_f32_32x32x8_bf16_1k instead of
v_mfma_f32_32x32x16_bf16) 0;  -DPAFUSE_MFMA_NOP=3|15 (s_nop before every MFMA) 51 / 70;
-mllvm -amdgpu-mfma-padding-ratio=100  62.  PAFUSE_DEBUG_F32_MASK=1|2 keeps the plain / whole-row layers on the fp32
matrix cores (1: 0 failures at depth 1, 2: 75-87), PAFUSE_DEBUG_LDS_PAD=163840 leaves every plain-GEMM workgroup alone on
its CU (26).  Product code never runs these kernels beside another queue (pafuse_d3dp_sample ignores side streams in the
bf16-MFMA modes); this script exists to re-check that decision on new hardware / ROCm releases.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import make_model  # noqa: E402
from pafuse_amd import synthetic as gu  # noqa: E402

DEV = "cuda"
P = 8


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    models = [make_model(P, 1, seed=51, depth=depth)[0] for _ in range(3)]
    for m in models:
        m.precision = prec
    x2d, _ = gu.synthetic_inputs_2d(B=1)
    g = torch.Generator().manual_seed(1)
    x3 = torch.randn(1, P, 27, 134, 3, generator=g).to(DEV)
    t = torch.tensor([499], device=DEV)
    idx = models[0].parts_joint_indices["body"]
    a2, a3 = x2d[..., idx, :].to(DEV).contiguous(), x3[..., idx, :].contiguous()
    ref = models[0].pose_estimator["body"](a2, a3, t)
    for m in models[1:]:
        assert torch.equal(m.pose_estimator["body"](a2, a3, t), ref)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    bad = 0
    for _ in range(80):
        outs = []
        for m, s in zip(models, streams):
            with torch.cuda.stream(s):
                outs.append(m.pose_estimator["body"](a2, a3, t))
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    print(f"precision {prec} depth {depth} mask {os.environ.get('PAFUSE_DEBUG_F32_MASK')}: {bad} of 240 concurrent "
          f"outputs differ from the single-stream result")


if __name__ == "__main__":
    main()
