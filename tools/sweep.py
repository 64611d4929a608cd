#!/usr/bin/env python3
"""hypotheses/s of the DDIM loop over the other single-GPU configurations: P in {1, 5, 20, 40} x B in
{1, 8, 75} (main_h3wb.py:306 evaluates all clips of a sequence at once) x {eager, one captured hipGraph}, T = 10, flip-TTA,
in the default product scheme, plus BASELINE configs[1] (P = 5, T = 5) in 'bf16' and in the default.  One process, one JSON
document: `python tools/sweep.py --out profiles/r04_sweep.json` on the GPU box."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from pafuse_amd import synthetic as gu  # noqa: E402

GFLOP_PER_HYP_PASS = 69.384706048
PEAK = {"f32": 157.3, "bf16x3": 2500.0 / 6, "f16x2": 2500.0 / 3, "bf16": 2500.0}


def measure(P, B, T, dtype, graph, budget_s=3.0):
    model, _ = ge.make_model(P, T, seed=51)
    model.precision, model.use_graph = dtype, graph
    x2d, x2f = gu.synthetic_inputs_2d(B=B)
    x2d, x2f = x2d.cuda(), x2f.cuda()
    torch.manual_seed(1)
    model(x2d, None, input_2d_flip=x2f)            # warm-up (builds the weight images; captures the graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model(x2d, None, input_2d_flip=x2f)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    steps = max(2, min(20, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model(x2d, None, input_2d_flip=x2f)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    assert out.shape == (B, T, P, 27, 134, 3) and bool(torch.isfinite(out).all())
    tf = B * P * 2 * T * GFLOP_PER_HYP_PASS / 1e3 / sec
    return {"P": P, "B": B, "T": T, "dtype": dtype, "graph": graph, "steps": steps, "ms_per_loop": round(sec * 1e3, 2),
            "hypotheses_per_s": round(B * P / sec, 2), "loop_tflops": round(tf, 1), "roofline_loop_frac": round(tf / PEAK[dtype], 4),
            "rows_per_library_call": min(2 * B * P, model.max_rows_per_launch // (2 * P) * 2 * P if 2 * P <= model.max_rows_per_launch else 2 * P)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    ge.build()
    rows = []
    default = ge.make_model(1, 1)[0].precision
    cases = [(P, B, 10, default, g) for P in (1, 5, 20, 40) for B in (1, 8, 75) for g in (False, True)]
    cases += [(5, 1, 5, "bf16", False), (5, 1, 5, default, False), (5, 1, 5, "bf16x3", False), (20, 1, 10, "bf16x3", False), (20, 1, 10, "f32", False)]
    if a.quick:
        cases = [c for c in cases if c[1] <= 8 and c[0] in (5, 20)]
    for (P, B, T, dtype, graph) in cases:
        if graph and 2 * B * P > 640:
            continue                                  # one graph per library call: the large batches run as several calls
        r = measure(P, B, T, dtype, graph)
        rows.append(r)
        print(json.dumps(r), flush=True)
    if a.out:
        from pafuse_amd._lib import kernel_source_digest
        with open(a.out, "w") as f:
            json.dump({"what": "hypotheses/s of D3DP.forward (flip-TTA DDIM loop) on one MI355X over P x B x {eager, hipGraph}; three HIP "
                               "streams (the default); default product scheme " + default,
                       "device": torch.cuda.get_device_name(0), "kernel_source_sha256": kernel_source_digest(), "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
