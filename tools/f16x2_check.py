#!/usr/bin/env python3
"""First-contact check of the f16x2 product scheme on the GPU (development tool; the assertions live in tests/).
Prints one JSON line per experiment: exactness on integer / 22-bit data, fp16-subnormal behaviour of the matrix cores,
error against fp64 per layout beside bf16x3 and the fp32 FMA chain, G4 block goldens, a P=2 / T=2 loop against G5."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from pafuse_amd import ops  # noqa: E402
from pafuse_amd import synthetic as gu  # noqa: E402

DEV = "cuda"


def emit(**kw):
    print(json.dumps(kw), flush=True)


def seeded(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


g = torch.Generator().manual_seed(5)
for layout in (0,):
  for (M, N, K) in ((131, 672, 224), (96, 224, 64), (300, 1152, 384), (513, 128, 32)):
      x = torch.randint(-3, 4, (M, K), generator=g).float()
      w = torch.randint(-3, 4, (N, K), generator=g).float() + torch.arange(N)[:, None].float() % 5
      b = torch.arange(N).float()
      out = ops.linear_split(x.to(DEV), w.to(DEV), b.to(DEV), layout=layout, scheme="f16x2").cpu()
      emit(test="small integers", layout=layout, exact=bool(torch.equal(out, x @ w.t() + b)), maxdiff=float((out - (x @ w.t() + b)).abs().max()))
      # 22-bit activations against one-hot powers of two: hi and lo both in play
      x = torch.randint(2 ** 21, 2 ** 22, (M, K), generator=g).float() * (torch.randint(0, 2, (M, K), generator=g) * 2 - 1) * 2.0 ** -8   # |x| < 65504
      kn = torch.randint(0, K, (N,), generator=g)
      w = torch.zeros(N, K)
      w[torch.arange(N), kn] = 2.0 ** torch.randint(-3, 4, (N,), generator=g).float()
      out = ops.linear_split(x.to(DEV), w.to(DEV), torch.zeros(N, device=DEV), layout=layout, scheme="f16x2").cpu()
      ref = x[:, kn] * w[torch.arange(N), kn]
      emit(test="22-bit activations", layout=layout, exact=bool(torch.equal(out, ref)), max_rel=float(((out - ref) / ref).abs().max()))
      # 22-bit weights against one-hot 0.5 activations
      w = torch.randint(2 ** 21, 2 ** 22, (N, K), generator=g).float()
      x = torch.zeros(M, K)
      km = torch.randint(0, K, (M,), generator=g)
      x[torch.arange(M), km] = 0.5
      out = ops.linear_split(x.to(DEV), w.to(DEV), torch.zeros(N, device=DEV), layout=layout, scheme="f16x2").cpu()
      ref = (w[:, km] * 0.5).t()
      emit(test="22-bit weights", layout=layout, exact=bool(torch.equal(out, ref)), max_rel=float(((out - ref) / ref).abs().max()))
      # tiny activations (fp16 subnormal hi): x = j * 2^-24 .. and 2^-20 scale, weights one-hot 1.0
      x = torch.randint(1, 2 ** 10, (M, K), generator=g).float() * 2.0 ** -26
      w = torch.zeros(N, K)
      w[torch.arange(N), kn] = 1.0
      out = ops.linear_split(x.to(DEV), w.to(DEV), torch.zeros(N, device=DEV), layout=layout, scheme="f16x2").cpu()
      ref = x[:, kn]
      emit(test="tiny activations (hi subnormal in fp16)", layout=layout, exact=bool(torch.equal(out, ref)),
           max_rel=float(((out - ref) / ref).abs().max()))
      # wide dynamic range inside one weight tensor: small weights next to a large one
      w = torch.zeros(N, K)
      w[torch.arange(N), kn] = 2.0 ** torch.randint(-20, 1, (N,), generator=g).float()
      x = torch.randint(2 ** 21, 2 ** 22, (M, K), generator=g).float() * 2.0 ** -22
      out = ops.linear_split(x.to(DEV), w.to(DEV), torch.zeros(N, device=DEV), layout=layout, scheme="f16x2").cpu()
      ref = x[:, kn] * w[torch.arange(N), kn]
      emit(test="weights spanning 2^-20..1 in one tensor", layout=layout, exact=bool(torch.equal(out, ref)),
           max_rel=float(((out - ref) / ref).abs().max()))

# error against fp64, per scheme
for (M, N, K) in [(200, 1152, 384), (300, 672, 224), (77, 384, 768), (64, 768, 256), (129, 448, 224)]:
    x, w, b = seeded((M, K), 1), seeded((N, K), 2, K ** -0.5), seeded((N,), 3, 0.1)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    row = {"test": "fp64 error", "shape": [M, N, K], "bound": 2.5e-7 * K ** 0.5 + 1e-6}
    row["fma32"] = float((ops.linear(x.to(DEV), w.to(DEV), b.to(DEV)).cpu().double() - ref).abs().mean())
    for scheme in ("bf16x3", "f16x2"):
        for layout in ((0, 2) if scheme == "bf16x3" else (0,)):
            o = ops.linear_split(x.to(DEV), w.to(DEV), b.to(DEV), layout=layout, scheme=scheme).cpu().double()
            row[f"{scheme}/{layout}"] = {"mean": float((o - ref).abs().mean()), "max": float((o - ref).abs().max())}
    if scheme == "f16x2":
        og = ops.linear_split(x.to(DEV), w.to(DEV), b.to(DEV), act="gelu", scheme=scheme).cpu().double()
        row["f16x2 gelu max"] = float((og - torch.nn.functional.gelu(ref)).abs().max())
    emit(**row)

# G4 block goldens through pafuse_block_forward (whole-row kernels + attention)
from functools import partial  # noqa: E402
from pafuse_amd.mixste2 import _BlockParams  # noqa: E402
from tests.conftest import load_golden  # noqa: E402
from tests.golden import golden_util as tgu  # noqa: E402
z = load_golden("g4_blocks.npz")
for part, C in tgu.PART_WIDTH.items():
    blk = _BlockParams(C, 2.0, True, partial(torch.nn.LayerNorm, eps=1e-6))
    blk.load_state_dict(tgu.seeded_like(blk.state_dict(), seed=41, prefix=f"g4.{part}."))
    blk = blk.to(DEV)
    for tag in ("s", "t"):
        row = {"test": "G4 block", "part": part, "tag": tag}
        for prec in ("f32", "bf16x3", "f16x2"):
            y = ops.block_forward(blk, z[f"{part}.x{tag}"].to(DEV), precision=prec).cpu()
            row[prec] = float((y - z[f"{part}.y{tag}"]).abs().max())
        emit(**row)

# G5 loop golden P=2, T=2 + mean error against the oracle's output
z = load_golden("g5_d3dp.npz")
model, sd = ge.make_model(2, 2, seed=51)
x2d, x2f = gu.synthetic_inputs_2d(B=1)
noises = gu.synthetic_noises(B=1, P=2, n=2, seed=1)
model.noise_fn = lambda k, shape, device: noises[k]
row = {"test": "G5 flip loop P=2 T=2 vs the reference's output"}
outs = {}
for prec in ("f32", "bf16x3", "f16x2"):
    model.precision = prec
    outs[prec] = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    d = (outs[prec] - z["flip_out"]).abs()
    row[prec] = {"max": float(d.max()), "mean": float(d.mean())}
row["f16x2_vs_bf16x3_max"] = float((outs["f16x2"] - outs["bf16x3"]).abs().max())
emit(**row)
# three streams against one stream, bit for bit (the multi-queue check of round 3, for the new MFMA)
model.precision = "f16x2"
model.n_aux_streams, model._aux_by_device = 0, {}
one = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
emit(test="f16x2: one stream (grouped grids) equals three streams", equal=bool(torch.equal(one, outs["f16x2"])))
