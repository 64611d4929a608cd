#!/usr/bin/env python3
"""Bring-up check of the fused MLP kernel (hgemm.hpp hmlp_kernel; development tool, the assertions live in tests/): the
flip loop with the MLP of every block as one kernel against the two-launch form (and, at depth 8 / P=2 / T=2, against the
reference's output: golden G5), per depth."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from pafuse_amd import synthetic as gu  # noqa: E402
from tests.conftest import load_golden  # noqa: E402

DEV = "cuda"
z = load_golden("g5_d3dp.npz")
which = sys.argv[1:] or ["body", "face", "hands"]
for depth, T in ((1, 1), (2, 1), (8, 2)):
    model, sd = ge.make_model(2, T, seed=51, depth=depth)
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=2, n=T, seed=1)
    model.noise_fn = lambda k, shape, device: noises[k]
    model.precision = "f16x2"
    outs = {}
    for tag, fused in (("two launches", {}), ("fused", {p: True for p in which})):
        for name, m in model.denoisers().items():
            m.fuse_mlp = fused.get(name, False)
        outs[tag] = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
        if depth == 8:
            d = (outs[tag] - z["flip_out"]).abs()
            print(json.dumps({"depth": depth, "mlp": tag, "parts": which if fused else [], "vs_reference_max": float(d.max()),
                              "vs_reference_mean": float(d.mean()), "finite": bool(torch.isfinite(outs[tag]).all())}), flush=True)
    d = (outs["fused"] - outs["two launches"]).abs()
    print(json.dumps({"depth": depth, "T": T, "fused_vs_two_launches_max": float(d.max()), "mean": float(d.mean()),
                      "out_rms": float(outs["two launches"].pow(2).mean().sqrt())}), flush=True)
