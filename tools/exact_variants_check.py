"""every parity case of the GPU tests for the exact-width kernel sets, against the frozen bounds (diagnostic)"""
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch
from oracle import d3dp_oracle as orc
from pafuse_amd import synthetic as gu
from tests.conftest import load_golden
from tests.test_hip_parity import loop_case, _mpjpe_report, _j_agg_compare
from tests.test_hip_fullsize import CHECKED, FULLSIZE_SELECTIONS, fullsize_case, g19_compare
bounds = json.load(open("tests/parity_bounds.json"))["cases"]

def row(name, base, diffs, d):
    m = {k: float(v.max()) for k, v in diffs.items()}
    m["J-Agg"] = d
    lit = bounds[base]["bound_mm"]
    f32 = bounds.get(base.replace("_bf16x3", "_f32"), {}).get("bound_mm")
    print(json.dumps({"case": name, "measured_e-4": {k: round(v * 1e4, 3) for k, v in m.items()},
                      "literal_ok": {k: m[k] <= lit[k] for k in m}, "literal_e-4": {k: round(lit[k] * 1e4, 3) for k in m},
                      "max_of_modes_ok": None if f32 is None else {k: m[k] <= max(lit[k], f32[k]) for k in m}}), flush=True)

for prec in sys.argv[1:] or ["bf16x3"]:
    for (B, P, T) in ((2, 3, 2), (1, 5, 5), (1, 20, 10)):
        name, out, ref, target, x2d = loop_case(B, P, T, prec)
        got, want = _mpjpe_report(out, target, x2d), _mpjpe_report(ref, target, x2d)
        diffs = {k: (got[k] - want[k]).abs() for k in ("J-Best", "P-Best", "P-Agg")}
        d, frac, worst = _j_agg_compare(out, ref, target, x2d)
        row(name, f"loop_B{B}_P{P}_T{T}_bf16x3", diffs, d)
    fs = fullsize_case(prec)
    out, ref = fs["out"][:, :, list(CHECKED)].cpu(), fs["ref"]
    target = orc.center_pose_parts(gu.synthetic_target_3d(1))
    for name, sel in FULLSIZE_SELECTIONS.items():
        o, r = out[:, :, sel].contiguous(), ref[:, :, sel].contiguous()
        got, want = _mpjpe_report(o, target, fs["x2d"]), _mpjpe_report(r, target, fs["x2d"])
        diffs = {k: (got[k] - want[k]).abs() for k in ("J-Best", "P-Best", "P-Agg")}
        d, frac, worst = _j_agg_compare(o, r, target, fs["x2d"])
        row(name + "@" + prec, name, diffs, d)
    z = load_golden("g19_metric_config.npz")
    pt, diffs, d, frac, worst = g19_compare(fs["out"][:, :, :20].cpu(), z, fs["x2d"])
    row("g19@" + prec + f" picks_diff {frac:.1e}", "g19_P20_T10_bf16x3", diffs, d)
