#!/usr/bin/env python3
"""How far the HIP path is from the CPU oracle on seeded inputs: pointwise, and |dMPJPE| (mm, fp64 metric math) for each
of the four protocols (J-Best / P-Best / P-Agg / J-Agg, main_h3wb.py:327-348) at every DDIM step.

Run on the GPU box:
    python tests/reports/parity_report.py [--out FILE.json] CASE [CASE ...]      CASE = P,T[,B[,precision]]
e.g. `5,5 20,10 5,5,1,f32 5,5,1,bf16` (precision defaults to the inference default, bf16x3).  One JSON object per case is printed; with --out they are also written as one JSON file
(`profiles/r02_parity_report.json` is this script's output, and the per-protocol bounds asserted in
tests/test_hip_parity.py::MPJPE_TOL_MM are read off it).  The oracle runs ALL P hypotheses here (the aggregation
protocols reduce over P): P=20, T=10 is about two minutes of host CPU.
"""
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import make_model  # noqa: E402
from oracle import d3dp_oracle as orc  # noqa: E402
from pafuse_amd import synthetic as gu  # noqa: E402
from tests.test_hip_parity import _j_agg_compare, _mpjpe_report  # noqa: E402


_ORACLE = {}        # (P, T, B) -> (oracle output, seconds): the oracle does not depend on the HIP path's product mode


def run_case(P, T, B=1, precision="bf16x3"):
    model, sd = make_model(P, T, seed=77)
    model.precision = precision
    x2d, x2f = gu.synthetic_inputs_2d(B=B)
    noises = gu.synthetic_noises(B=B, P=P, n=T, seed=3)
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.cuda(), None, input_2d_flip=x2f.cuda()).cpu()
    if (P, T, B) not in _ORACLE:
        t0 = time.time()
        ref = orc.ddim_sample(sd, x2d, noises, T, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
        _ORACLE[(P, T, B)] = (ref, time.time() - t0)
    ref, cpu_s = _ORACLE[(P, T, B)]
    target = orc.center_pose_parts(gu.synthetic_target_3d(B))
    got, want = _mpjpe_report(out, target, x2d), _mpjpe_report(ref, target, x2d)
    d = (out - ref).abs()
    jagg_same, jagg_flip_frac, jagg_flip_margin = _j_agg_compare(out, ref, target, x2d)
    # J-Agg picks, per (frame, joint), the hypothesis with the smallest 2-D reprojection error: count the joints whose
    # pick differs between the two runs (a near-tie decided the other way swaps in another hypothesis' 3-D error)
    return {
        "B": B, "P": P, "T": T, "precision": model.precision, "oracle_cpu_s": round(cpu_s, 2),
        "pointwise_max_abs_m": d.max().item(), "pointwise_mean_abs_m": d.mean().item(),
        "per_step_pointwise_max_abs_m": [d[:, k].max().item() for k in range(T)],
        "clamped_frac": (ref.abs() >= 1.1).float().mean().item(),
        "mpjpe_mm_abs_diff_per_step": {k: [abs(v) for v in (got[k] - want[k]).tolist()] for k in want},
        "mpjpe_mm_abs_diff_max": {k: (got[k] - want[k]).abs().max().item() for k in want},
        "mpjpe_mm_oracle_per_step": {k: want[k].tolist() for k in want},
        "j_agg_same_picks_mm_abs_diff_max": jagg_same, "j_agg_fraction_of_joints_with_different_pick": jagg_flip_frac,
        "j_agg_largest_2d_margin_m_among_different_picks": jagg_flip_margin,
        "north_star_1e-4mm_met": {k: bool((got[k] - want[k]).abs().max().item() <= 1e-4) for k in want},
    }


def main(argv):
    out_path = None
    if argv and argv[0] == "--out":
        out_path, argv = argv[1], argv[2:]
    cases = []
    for spec in argv or ["5,5"]:
        f = spec.split(",")
        cases.append((int(f[0]), int(f[1]), int(f[2]) if len(f) > 2 else 1, f[3] if len(f) > 3 else "bf16x3"))
    try:
        sha = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        sha = os.environ.get("PAFUSE_GIT_SHA", "unknown (gpurun snapshot has no .git)")
    results = []
    for c in cases:
        r = run_case(*c)
        print(json.dumps(r), flush=True)
        results.append(r)
    if out_path:
        from pafuse_amd._lib import kernel_source_digest
        doc = {"what": "HIP path vs CPU oracle (oracle/d3dp_oracle.py, pinned to the reference by tests/golden), seeded "
                       "synthetic weights seed 77, inputs seed 1234, noise seed 3; MPJPE in mm with fp64 metric arithmetic",
               "device": torch.cuda.get_device_name(0), "git_sha": sha, "kernel_source_sha256": kernel_source_digest(),
               "cases": results}
        with open(out_path, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
