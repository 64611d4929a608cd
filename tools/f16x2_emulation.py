#!/usr/bin/env python3
"""CPU emulation of the candidate fp32-equivalent product schemes of the linear layers, against exact (fp64) arithmetic.
Runs in the build container (no GPU); decides whether the three-product fp16 scheme is worth building.

  fma32   what the f32 matrix cores do: a k-ordered fp32 FMA chain per output
  bf16x3  production today: operands = exact sum of three bf16 slices, six products per k, fp32 accumulate, one rounding
          per 16-deep MFMA
  f16x2   candidate: a = hi + lo 2^-11 with hi = f16(a), lo = f16((a - hi) 2^11); the weight is pre-scaled by a power of two
          (max |W 2^k| < 2^15) and stored as three fp16 slices  w0 = f16(Ws), w1 = f16(Ws - w0), w2 = f16(w0 2^-11);
          products  hi w0 + hi w1 + lo w2  (three MFMAs, ONE accumulator, no scaling of the accumulator inside the loop),
          result 2^-k acc.  Dropped: lo x (Ws - w0) ~ 2^-22 relative.

Unit level: mean / max |error| of the three schemes on the hot path's layer shapes.
`--pass`: one denoiser pass per part through the oracle with its F.linear swapped for each emulation (test infrastructure:
tools/ may not import oracle/ in the product path, this is a report like tests/reports/*)."""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _f16(x):
    return x.to(torch.float16).to(torch.float32)


def split_bf16x3(x):
    s0 = _bf16(x)
    r1 = x - s0
    s1 = _bf16(r1)
    s2 = _bf16(r1 - s1)
    return s0, s1, s2


def weight_scale_exp(W):
    """k with max |W 2^k| in [2^14, 2^15)"""
    m = W.abs().max().item()
    return 0 if m == 0 else 14 - math.floor(math.log2(m))


def split_w_f16(W):
    k = weight_scale_exp(W)
    Ws = W * (2.0 ** k)
    w0 = _f16(Ws)
    w1 = _f16(Ws - w0)
    w2 = _f16(w0 * 2.0 ** -11)
    return w0, w1, w2, k


def split_a_f16(A):
    hi = _f16(A)
    lo = _f16((A - hi) * 2048.0)
    return hi, lo


def chunked(terms, K, depth):
    """fp32 accumulation with one rounding per MFMA: terms = list of (A_slice [M,K], W_slice [N,K]) in issue order per chunk."""
    M, N = terms[0][0].shape[0], terms[0][1].shape[0]
    acc = torch.zeros(M, N, dtype=torch.float32)
    for k0 in range(0, K, depth):
        for a, w in terms:
            part = a[:, k0:k0 + depth].double() @ w[:, k0:k0 + depth].double().T   # exact: products <= 22 bits, 32 terms
            acc = (acc.double() + part).float()
    return acc


def gemm_fma32(A, W):
    """k-ordered fp32 FMA chain (v_mfma_f32_32x32x2_f32 is bitwise this)"""
    acc = torch.zeros(A.shape[0], W.shape[0], dtype=torch.float64)
    for k in range(A.shape[1]):
        acc = (acc + A[:, k:k + 1].double() * W[:, k].double()[None, :]).float().double()   # fma: exact product, one rounding
    return acc.float()


def gemm_bf16x3(A, W, depth=16):
    a0, a1, a2 = split_bf16x3(A)
    w0, w1, w2 = split_bf16x3(W)
    return chunked([(a2, w0), (a0, w2), (a1, w1), (a1, w0), (a0, w1), (a0, w0)], A.shape[1], depth)


def gemm_f16x2(A, W, depth=16):
    hi, lo = split_a_f16(A)
    w0, w1, w2, k = split_w_f16(W)
    return chunked([(lo, w2), (hi, w1), (hi, w0)], A.shape[1], depth) * (2.0 ** -k)


SCHEMES = {"fma32": gemm_fma32, "bf16x3": gemm_bf16x3, "f16x2": gemm_f16x2}


def unit_level(seed=0):
    g = torch.Generator().manual_seed(seed)
    rows = []
    for name, (M, N, K, wstd, astd, amean) in {
        "body qkv (LN output)": (512, 1152, 384, 0.03, 1.0, 0.0),
        "body qkv folded (raw residual, mean 2)": (512, 1152, 384, 0.03, 3.0, 2.0),
        "face fc2 (GELU output, K=448)": (512, 224, 448, 0.03, 0.6, 0.2),
        "hands proj": (512, 256, 256, 0.04, 0.5, 0.0),
        "integers (exactness)": (64, 64, 64, 0, 0, 0),
    }.items():
        if wstd == 0:
            A = torch.randint(-2000, 2000, (M, K), generator=g).float()
            W = torch.randint(-7, 8, (N, K), generator=g).float()
        else:
            A = torch.randn(M, K, generator=g) * astd + amean
            W = torch.randn(N, K, generator=g) * wstd
        exact = A.double() @ W.double().T
        row = {"shape": name, "out_rms": exact.pow(2).mean().sqrt().item()}
        for s, fn in SCHEMES.items():
            e = fn(A, W).double() - exact
            row[s] = {"mean_abs": e.abs().mean().item(), "max": e.abs().max().item()}
        rows.append(row)
        print(json.dumps(row), flush=True)
    return rows


def one_pass(P=2, seed=77):
    """error_budget.py's experiment with the oracle's linear layers swapped for each emulated scheme"""
    import torch.nn.functional as F
    from oracle import d3dp_oracle as orc
    from pafuse_amd import synthetic as gu
    from tests.golden import golden_util  # noqa: F401
    sd = None
    import pafuse_amd  # noqa: F401
    from tests.golden.state_template import d3dp_template   # shapes of the 636-entry state dict without the HIP library
    sd = gu.seeded_state_dict(d3dp_template(), seed=seed)
    sd64 = {k: v.double() for k, v in sd.items()}
    x2d, _ = gu.synthetic_inputs_2d(B=1)
    g = torch.Generator().manual_seed(52)
    x3d = torch.randn(1, P, 27, 134, 3, generator=g).clamp(-1.1, 1.1)
    real_linear = F.linear
    cache = {}

    def emulated(fn):
        def lin(x, w, b=None):
            if x.dtype != torch.float32 or w.shape[1] < 64:   # patch embedding / time MLP input stay as they are (not MFMA layers)
                return real_linear(x, w, b)
            shp = x.shape[:-1]
            y = fn(x.reshape(-1, x.shape[-1]).contiguous(), w)
            if b is not None:
                y = y + b
            return y.reshape(*shp, w.shape[0])
        return lin

    rows = []
    for tval in (999, 499, 99):
        t = torch.tensor([tval])
        for part, idx in orc.PART_JOINTS.items():
            pre = f"pose_estimator.{part}."
            o64 = orc.mixste2_eval(sd64, pre, x2d[..., idx, :].double(), x3d[..., idx, :].double(), t)
            row = {"t": tval, "part": part, "out_rms": o64.pow(2).mean().sqrt().item()}
            o32 = orc.mixste2_eval(sd, pre, x2d[..., idx, :], x3d[..., idx, :], t)
            e = o32.double() - o64
            row["oracle32"] = {"mean_abs": e.abs().mean().item(), "max": e.abs().max().item()}
            for s in ("bf16x3", "f16x2", "f16x2_hres"):
                orc.F.linear = emulated(SCHEMES[s.split("_")[0]])
                real_block, real_ln = orc.transformer_block, orc._layer_norm
                if s.endswith("_hres"):
                    # the residual stream kept ONLY as its two-slice (hi + lo 2^-11) image: every value a whole-row kernel
                    # stores as x is rounded to that grid (after the attention residual; after the block's post-norm)
                    def r2(x):
                        hi, lo = split_a_f16(x)
                        return hi + lo * (2.0 ** -11)

                    def block(sd_, pre_, x, heads, eps=1e-6, drop=None, qk_scale=None):
                        a = orc._self_attention(sd_, pre_ + "attn.", real_ln(sd_, pre_ + "norm1", x, eps), heads, qk_scale)
                        x = r2(x + a)
                        h_ = orc.F.linear(real_ln(sd_, pre_ + "norm2", x, eps), sd_[pre_ + "mlp.fc1.weight"], sd_[pre_ + "mlp.fc1.bias"])
                        m_ = orc.F.linear(torch.nn.functional.gelu(h_), sd_[pre_ + "mlp.fc2.weight"], sd_[pre_ + "mlp.fc2.bias"])
                        return x + m_

                    def ln(sd_, key, x, eps):
                        y = real_ln(sd_, key, x, eps)
                        return r2(y) if key.endswith(("Spatial_norm", "Temporal_norm")) else y
                    orc.transformer_block, orc._layer_norm = block, ln
                try:
                    o = orc.mixste2_eval(sd, pre, x2d[..., idx, :], x3d[..., idx, :], t)
                finally:
                    orc.F.linear = real_linear
                    orc.transformer_block, orc._layer_norm = real_block, real_ln
                e = o.double() - o64
                row[s] = {"mean_abs": e.abs().mean().item(), "max": e.abs().max().item()}
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pass", dest="one_pass", action="store_true")
    ap.add_argument("--out")
    a = ap.parse_args()
    torch.set_num_threads(8)
    doc = {"unit": unit_level()}
    if a.one_pass:
        doc["pass"] = one_pass()
        for s in ("oracle32", "bf16x3", "f16x2", "f16x2_hres"):
            doc.setdefault("pass_mean_abs_mm", {})[s] = 1e3 * float(np.mean([r[s]["mean_abs"] for r in doc["pass"]]))
        print(json.dumps(doc["pass_mean_abs_mm"]))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)
