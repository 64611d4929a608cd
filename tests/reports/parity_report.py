#!/usr/bin/env python3
"""How far the HIP path is from the CPU oracle on seeded inputs: pointwise, and |dMPJPE| (mm, fp64 metric math) for each
of the four protocols (J-Best / P-Best / P-Agg / J-Agg, main_h3wb.py:327-348) at every DDIM step.

Run on the GPU box:
    python tests/reports/parity_report.py --out profiles/r04_parity_report.json [--no-fullsize] [extra P,T,B,precision ...]
It runs exactly the case functions the GPU tests run (tests/test_hip_parity.py::loop_case for LOOP_CASES x {f32, bf16x3},
tests/test_hip_fullsize.py::fullsize_case) plus the metric's own P=20, T=10 loop in both fp32-grade modes and the opt-in
bf16 mode at P=5, T=5; every case is recorded under its name together with the SHA-256 of the oracle's output.  The tests
assert 1.25 x these measurements per case and protocol (tests/test_hip_parity.py::parity_bounds) - valid for the oracle
arithmetic recorded here, which the hash identifies.
"""
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import d3dp_oracle as orc  # noqa: E402
from pafuse_amd import synthetic as gu  # noqa: E402
from tests.test_hip_parity import LOOP_CASES, _j_agg_compare, _mpjpe_report, loop_case, tensor_sha256  # noqa: E402


def measure(name, out, ref, target, x2d, precision, cpu_s=None):
    got, want = _mpjpe_report(out, target, x2d), _mpjpe_report(ref, target, x2d)
    d = (out - ref).abs()
    T = out.shape[1]
    jagg_same, jagg_flip_frac, jagg_flip_margin = _j_agg_compare(out, ref, target, x2d)
    pairs = [(got[k] - want[k]).abs() for k in ("J-Best", "P-Best", "P-Agg")]
    return {
        "name": name, "B": out.shape[0], "P": out.shape[2], "T": T, "precision": precision,
        "oracle_sha256": tensor_sha256(ref), "oracle_cpu_s": None if cpu_s is None else round(cpu_s, 2),
        "pointwise_max_abs_m": d.max().item(), "pointwise_mean_abs_m": d.mean().item(),
        "per_step_pointwise_max_abs_m": [d[:, k].max().item() for k in range(T)],
        "clamped_frac": (ref.abs() >= 1.1).float().mean().item(),
        "mpjpe_mm_abs_diff_per_step": {k: [abs(v) for v in (got[k] - want[k]).tolist()] for k in want},
        "mpjpe_mm_abs_diff_max": {k: (got[k] - want[k]).abs().max().item() for k in want},
        "mpjpe_mm_oracle_per_step": {k: want[k].tolist() for k in want},
        "j_agg_same_picks_mm_abs_diff_max": jagg_same, "j_agg_fraction_of_joints_with_different_pick": jagg_flip_frac,
        "j_agg_largest_2d_margin_m_among_different_picks": jagg_flip_margin,
        "north_star_1e-4mm_met": {k: bool((got[k] - want[k]).abs().max().item() <= 1e-4) for k in want},
        "north_star_1e-4mm_met_fraction_of_step_protocol_pairs": sum(int((v <= 1e-4).sum()) for v in pairs) / sum(v.numel() for v in pairs),
    }


def main(argv):
    out_path, fullsize = None, True
    if argv and argv[0] == "--out":
        out_path, argv = argv[1], argv[2:]
    if argv and argv[0] == "--no-fullsize":
        fullsize, argv = False, argv[1:]
    extra = [(1, 20, 10, "bf16x3_images"), (1, 20, 10, "f32"), (1, 20, 10, "f16x2"), (1, 5, 5, "bf16")]
    for spec in argv:
        f = spec.split(",")
        extra.append((int(f[2]) if len(f) > 2 else 1, int(f[0]), int(f[1]), f[3] if len(f) > 3 else "bf16x3_images"))
    try:
        sha = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        sha = os.environ.get("PAFUSE_GIT_SHA", "unknown (gpurun snapshot has no .git)")
    results = []

    def emit(r):
        print(json.dumps(r), flush=True)
        results.append(r)

    done = set()
    for (B, P, T) in LOOP_CASES:
        for prec in ("f32", "bf16x3_images", "f16x2"):
            t0 = time.time()
            name, out, ref, target, x2d = loop_case(B, P, T, prec)
            emit(measure(name, out, ref, target, x2d, prec, time.time() - t0))
            done.add((B, P, T, prec))
    for (B, P, T, prec) in extra:
        if (B, P, T, prec) in done:
            continue
        t0 = time.time()
        name, out, ref, target, x2d = loop_case(B, P, T, prec)
        emit(measure(name, out, ref, target, x2d, prec, time.time() - t0))
    if fullsize:
        from __graft_entry__ import make_model
        from tests.conftest import load_golden
        from tests.test_hip_fullsize import CHECKED, FULLSIZE_SELECTIONS, T_FULL, fullsize_case, g19_compare
        t0 = time.time()
        fs = fullsize_case()
        out, ref = fs["out"][:, :, list(CHECKED)].cpu(), fs["ref"]
        target = orc.center_pose_parts(gu.synthetic_target_3d(1))
        for name, sel in FULLSIZE_SELECTIONS.items():
            emit(measure(name, out[:, :, sel].contiguous(), ref[:, :, sel].contiguous(), target, fs["x2d"], "bf16x3_images", time.time() - t0))
        # the metric's own configuration against the REFERENCE's run of it (golden G19): all 20 hypotheses, every step
        z = load_golden("g19_metric_config.npz")
        for prec in ("bf16x3_images", "f32", "f16x2"):
            if prec == "bf16x3_images":
                out20 = fs["out"][:, :, :20].cpu()
            else:
                model, _ = make_model(20, T_FULL, seed=51)
                model.precision = prec
                noises = [n[:, :20].contiguous() for n in fs["noises"]]
                model.noise_fn = lambda k, shape, device: noises[k]
                out20 = model(fs["x2d"].cuda(), None, input_2d_flip=fs["x2f"].cuda()).cpu()
            if prec == "f16x2":     # and against the oracle on this box, all 20 hypotheses (tests: test_full_size_f16x2_vs_oracle)
                emit(measure("fullsize_20of20_T10_f16x2", out20, fs["ref"][:, :, :20].contiguous(), target, fs["x2d"], "f16x2"))
            pt, diffs, d, frac, worst = g19_compare(out20, z, fs["x2d"])
            emit({"name": f"g19_P20_T10_{prec}", "B": 1, "P": 20, "T": T_FULL, "precision": prec,
                  "against": "tests/golden/g19_metric_config.npz (the reference's own run of BASELINE configs[2])",
                  "oracle_sha256": None, "pointwise_max_abs_m_on_stored_trajectories": pt,
                  "mpjpe_mm_abs_diff_per_step": {k: v.tolist() for k, v in diffs.items()},
                  "mpjpe_mm_abs_diff_max": {**{k: float(v.max()) for k, v in diffs.items()}, "J-Agg": d},
                  "j_agg_same_picks_mm_abs_diff_max": d, "j_agg_fraction_of_joints_with_different_pick": frac,
                  "j_agg_largest_2d_margin_m_among_different_picks": worst,
                  "north_star_1e-4mm_met_fraction_of_step_protocol_pairs":
                      sum(int((v <= 1e-4).sum()) for v in diffs.values()) / sum(v.numel() for v in diffs.values())})
    if out_path:
        from pafuse_amd._lib import kernel_source_digest
        doc = {"what": "HIP path vs CPU oracle (oracle/d3dp_oracle.py, pinned to the reference by tests/golden) on the GPU tests' own "
                       "seeded cases; MPJPE in mm with fp64 metric arithmetic; the tests assert 1.25 x mpjpe_mm_abs_diff_max per case",
               "device": torch.cuda.get_device_name(0), "git_sha": sha, "kernel_source_sha256": kernel_source_digest(),
               "cpu": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
               "torch": torch.__version__, "cases": results}
        with open(out_path, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
