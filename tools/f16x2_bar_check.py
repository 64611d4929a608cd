"""VERDICT r4 item 1, the alternative bar for an f16x2 headline: on loop_B2_P3_T2 the north star's 1e-4 mm met on every (step,
protocol) pair where the f32 mode meets it, zero J-Agg pick differences on G19, the frozen 'bf16x3_images' bounds held.  Measures the
f16x2 variants (residual stream as H image only / fp32 rows kept) on the GPU box and prints one JSON line per variant."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_hip_parity as tp
from tests.conftest import load_golden
from tests.golden import golden_util as gu
from oracle import d3dp_oracle as orc

DEV = "cuda"


def run_case(B, P, T, precision, keep):
    from __graft_entry__ import make_model
    model, sd = make_model(P, T, seed=77)
    model.precision = precision
    for m in model.denoisers().values():
        m.keep_f32_residual = keep
    x2d, x2f = gu.synthetic_inputs_2d(B=B, seed=1234)
    noises = gu.synthetic_noises(B=B, P=P, n=T, seed=3)
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    key = (B, P, T)
    if key not in tp._LOOP_ORACLE:
        tp._LOOP_ORACLE[key] = orc.ddim_sample(sd, x2d, noises, T, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    ref = tp._LOOP_ORACLE[key]
    target = orc.center_pose_parts(gu.synthetic_target_3d(B))
    got, want = tp._mpjpe_report(out, target, x2d), tp._mpjpe_report(ref, target, x2d)
    diffs = {k: (got[k] - want[k]).abs() for k in ("J-Best", "P-Best", "P-Agg")}
    d, frac, worst = tp._j_agg_compare(out, ref, target, x2d)
    return {"case": f"loop_B{B}_P{P}_T{T}", "precision": precision, "keep_f32_residual": keep,
            "max_mm": {k: float(v.max()) for k, v in diffs.items()}, "j_agg_same_picks_mm": d, "j_agg_pick_diff_frac": frac,
            "pairs_le_1e-4": int(sum(int((v <= 1e-4).sum()) for v in diffs.values())), "pairs": int(sum(v.numel() for v in diffs.values()))}


def g19(precision, keep):
    from __graft_entry__ import make_model
    from tests.test_hip_fullsize import g19_compare, T_FULL
    z = load_golden("g19_metric_config.npz")
    model, _ = make_model(20, T_FULL, seed=51)
    model.precision = precision
    for m in model.denoisers().values():
        m.keep_f32_residual = keep
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = [n[:, :20].contiguous() for n in gu.synthetic_noises(B=1, P=160, n=T_FULL, seed=160)]
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    pt, diffs, d, frac, worst = g19_compare(out, z, x2d)
    return {"case": "g19", "precision": precision, "keep_f32_residual": keep, "pointwise_max": pt,
            "max_mm": {k: float(v.max()) for k, v in diffs.items()}, "j_agg_same_picks_mm": d, "j_agg_pick_diff_frac": frac}


if __name__ == "__main__":
    for precision, keep in (("f32", False), ("bf16x3_images", False), ("f16x2", False), ("f16x2", True)):
        for B, P, T in ((2, 3, 2), (1, 5, 5)):
            print(json.dumps(run_case(B, P, T, precision, keep)), flush=True)
        print(json.dumps(g19(precision, keep)), flush=True)
