#!/usr/bin/env python3
"""Unit bring-up of hmlp_kernel through pafuse_mlp_h: against fp64, with diagnostics by column block / row block."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
from pafuse_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(3)
for C_ in [int(a) for a in (sys.argv[1:] or ["224", "256", "384"])]:
    for M in (128, 300, 7344):
        x = torch.randn(M, C_, generator=g)
        xc = (x - x.mean(1, keepdim=True))
        rstd = 1.0 / torch.sqrt(xc.double().pow(2).mean(1) + 1e-6).float()
        W1 = torch.randn(2 * C_, C_, generator=g) * C_ ** -0.5
        b1 = torch.randn(2 * C_, generator=g) * 0.1
        W2 = torch.randn(C_, 2 * C_, generator=g) * (2 * C_) ** -0.5
        b2 = torch.randn(C_, generator=g) * 0.1
        for variant in ("plain", "integers"):
            if variant == "integers":   # exactness: small integers everywhere, GELU of large |v| is v or 0
                xc = torch.randint(-2, 3, (M, C_), generator=g).float()
                W1 = torch.randint(-2, 3, (2 * C_, C_), generator=g).float()
                b1 = torch.randint(-2, 3, (2 * C_,), generator=g).float() * 64
                W2 = torch.randint(-1, 2, (C_, 2 * C_), generator=g).float() * 2.0 ** -10
                b2 = torch.zeros(C_)
                rstd = torch.ones(M)
            y, st = ops.mlp_fused(xc.to(DEV), rstd.to(DEV), W1.to(DEV), b1.to(DEV), W2.to(DEV), b2.to(DEV), in_place=(os.environ.get("INPLACE") == "1"))
            hid = torch.nn.functional.gelu(rstd.double()[:, None] * (xc.double() @ W1.double().t()) + b1.double())
            ref = xc.double() + hid @ W2.double().t() + b2.double()
            mean = ref.mean(1, keepdim=True)
            d = (y.cpu().double() - (ref - mean)).abs()
            row = {"C": C_, "M": M, "variant": variant, "in_place": os.environ.get("INPLACE") == "1", "max": float(d.max()), "mean": float(d.mean()), "ref_rms": float((ref - mean).pow(2).mean().sqrt()),
                   "stats_mean_err": float((st[:, 0].cpu().double() - mean[:, 0]).abs().max()),
                   "worst_row": int(d.max(1).values.argmax()), "rows_above_1e-5": int((d.max(1).values > 1e-5).sum())}
            print(json.dumps(row), flush=True)
