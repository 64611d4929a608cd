"""GPU parity at the metric's own sizes (BASELINE configs[2] P=20/T=10 and configs[3] P=160/T=10 in eight P=20
shards on one GPU) and the direct, bit-exact tests of the index-carrying stages (pafuse_embed / pafuse_ddim_finalize).

Full-size checks use what the domain offers: a hypothesis' DDIM trajectory depends on nothing but its
own noise draws, so (a) any sharding of the hypothesis axis must reproduce the unsharded run bit for bit, and (b) the
CPU oracle, run on a few hypotheses only, is an exact check of those hypotheses inside the big run.

MPJPE bounds are per case and per protocol, 1.25 x the measurements committed in profiles/r03_parity_report.json
(tests/reports/parity_report.py runs this module's case function on MI355X); what is asserted, and why north_star's
1e-4 mm is out of reach for two fp32 implementations, is stated in tests/test_hip_parity.py next to parity_bounds.
"""
import ctypes as C

import pytest
import torch

from oracle import d3dp_oracle as orc
from tests.conftest import load_golden
from tests.golden import golden_util as gu
from tests.test_hip_parity import _assert_mpjpe_parity

pytestmark = pytest.mark.gpu
DEV = "cuda"
T_FULL = 10
CHECKED = tuple(range(20)) + (83, 159)   # hypotheses the oracle re-computes live: ALL 20 of the metric's P=20 run (= the first
#                                          20 of the P=160 run, same noise) + two from the far shards.  The same 20 are also held
#                                          against the reference's own output (golden G19, test_g19_metric_config_vs_reference)


def fullsize_case(precision="bf16x3"):
    """ONE P=160, T=10, B=1 run (configs[3]'s hypothesis count on one GPU) + the oracle on 22 of its hypotheses, all ten
    steps (about a minute of host CPU at 16 threads).  Shared by the tests below and tests/reports/parity_report.py."""
    from __graft_entry__ import make_model
    model, sd = make_model(160, T_FULL, seed=51)
    model.precision = precision       # the committed full-size cases are bf16x3's; f16x2 has its own tests below
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=160, n=T_FULL, seed=160)
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    sub = [n[:, list(CHECKED)] for n in noises]
    ref = orc.ddim_sample(sd, x2d, sub, T_FULL, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    return dict(model=model, sd=sd, x2d=x2d, x2f=x2f, noises=noises, out=out, ref=ref)


FULLSIZE_SELECTIONS = {"fullsize_20of20_T10_bf16x3": slice(0, 20), "fullsize_22of160_T10_bf16x3": slice(0, 22)}


@pytest.fixture(scope="module")
def full160():
    return fullsize_case()


def test_p160_single_run_shape_and_finite(full160):
    out = full160["out"]
    assert out.shape == (1, T_FULL, 160, 27, 134, 3) and bool(torch.isfinite(out).all())


def test_p160_in_eight_p20_shards_equals_single_run(full160):
    """configs[3]: P=160 sharded 20 per rank.  Each shard is run the way a rank runs it (full-P noise drawn, own
    slice kept: pafuse_amd.parallel.shard_range) and must equal its slice of the single P=160 run bit for bit."""
    from pafuse_amd.parallel import shard_range
    m, x2d, x2f = full160["model"], full160["x2d"].to(DEV), full160["x2f"].to(DEV)
    try:
        for rank in range(8):
            lo, hi = shard_range(160, rank, 8)
            assert hi - lo == 20
            m.proposal_shard = (lo, hi)
            part = m(x2d, None, input_2d_flip=x2f)
            assert torch.equal(part, full160["out"][:, :, lo:hi]), rank
    finally:
        m.proposal_shard = None


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_p20_t10_equals_its_halves_and_the_p160_prefix(full160, precision):
    """configs[2], the metric's own configuration (B=1, P=20, T=10): the run equals the concatenation of its
    proposal_shard halves, and - same noise, same product scheme - the first 20 hypotheses of the P=160 run."""
    from __graft_entry__ import make_model
    model, _ = make_model(20, T_FULL, seed=51)
    model.precision = precision
    noises = [n[:, :20].contiguous() for n in full160["noises"]]
    model.noise_fn = lambda k, shape, device: noises[k]
    x2d, x2f = full160["x2d"].to(DEV), full160["x2f"].to(DEV)
    full = model(x2d, None, input_2d_flip=x2f)
    assert full.shape == (1, T_FULL, 20, 27, 134, 3)
    halves = []
    for lo, hi in ((0, 10), (10, 20)):
        model.proposal_shard = (lo, hi)
        halves.append(model(x2d, None, input_2d_flip=x2f))
    model.proposal_shard = None
    assert torch.equal(full, torch.cat(halves, dim=2))
    if precision == "bf16x3":
        assert torch.equal(full, full160["out"][:, :, :20])
    # and as one captured hipGraph (the loop bench.py --graph times)
    model.use_graph = True
    assert torch.equal(model(x2d, None, input_2d_flip=x2f), full)


def test_full_size_trajectories_vs_oracle(full160):
    """the oracle on ALL 20 hypotheses of the metric's P=20 run and {83, 159} of the P=160 run, all ten steps:
    pointwise 1e-5, and the four MPJPE protocols (over the 20, and over all 22) within the per-case bounds."""
    out = full160["out"][:, :, list(CHECKED)].cpu()
    ref = full160["ref"]
    assert out.shape == ref.shape == (1, T_FULL, len(CHECKED), 27, 134, 3)
    d = (out - ref).abs()
    assert float(d.max()) <= 1e-5, [float(d[:, k].max()) for k in range(T_FULL)]
    target = orc.center_pose_parts(gu.synthetic_target_3d(1))
    for case, sel in FULLSIZE_SELECTIONS.items():      # the P=20 run's hypotheses, then all 22
        _assert_mpjpe_parity(out[:, :, sel].contiguous(), ref[:, :, sel].contiguous(), target, full160["x2d"], case)


@pytest.mark.parametrize("precision", ["f16x2"])
def test_full_size_opt_in_modes_vs_oracle(full160, precision):
    """the metric's configuration (P=20, T=10) with the opt-in product modes - f16x2, and bf16x3 on the image pipeline - against
    the oracle on all 20 hypotheses, all ten steps: pointwise 1e-5 and the four MPJPE protocols within the frozen bounds of the
    case (tests/parity_bounds.json)."""
    from __graft_entry__ import make_model
    model, _ = make_model(20, T_FULL, seed=51)
    model.precision = precision
    noises = [n[:, :20].contiguous() for n in full160["noises"]]
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(full160["x2d"].to(DEV), None, input_2d_flip=full160["x2f"].to(DEV)).cpu()
    ref = full160["ref"][:, :, :20].contiguous()
    assert float((out - ref).abs().max()) <= 1e-5
    target = orc.center_pose_parts(gu.synthetic_target_3d(1))
    _assert_mpjpe_parity(out, ref, target, full160["x2d"], f"fullsize_20of20_T10_{precision}")


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_metric_config_vs_fp64_truth(full160, precision):
    """hypotheses {0, 7, 19} of the metric's configuration (P=20, T=10) through all ten steps against the
    oracle evaluated in fp64 - per protocol the HIP path is not further from exact arithmetic than the reference's own fp32
    arithmetic is (x 1.1, + 2e-5 mm), in both fp32-equivalent product schemes."""
    from __graft_entry__ import make_model
    from tests.test_hip_parity import assert_not_further_from_fp64, fp64_truth
    sub = [0, 7, 19]
    noises = [n[:, sub].contiguous() for n in full160["noises"]]
    if "truth3" not in full160:
        full160["truth3"] = fp64_truth(full160["sd"], full160["x2d"], full160["x2f"], noises, T_FULL)
    if precision == "bf16x3":
        out = full160["out"][:, :, sub].cpu()
    else:
        model, _ = make_model(3, T_FULL, seed=51)
        model.precision = precision
        model.noise_fn = lambda k, shape, device: noises[k]
        out = model(full160["x2d"].to(DEV), None, input_2d_flip=full160["x2f"].to(DEV)).cpu()
    ref32 = full160["ref"][:, :, sub].contiguous()
    target = orc.center_pose_parts(gu.synthetic_target_3d(1))
    assert_not_further_from_fp64(f"metric_config_h0_7_19_{precision}", out, ref32, full160["truth3"], target, full160["x2d"])


def g19_compare(out20, z, x2d):
    """HIP predictions [1,10,20,27,134,3] of the metric's configuration against golden G19 (the REFERENCE's own run of it):
    (pointwise max |d| on the stored trajectories, {protocol: |dMPJPE| per step in mm} for J-Best / P-Best / P-Agg,
    J-Agg |d| on the joints where both pick the same hypothesis, fraction of different picks, largest 2-D margin there)."""
    from tests.test_hip_parity import _j_agg_parts, _mpjpe_report
    target = orc.center_pose_parts(gu.synthetic_target_3d(1))
    sub = [int(i) for i in z["sub"]]
    pt = float((out20[:, :, sub] - z["out_sub"]).abs().max())
    got = _mpjpe_report(out20, target, x2d)
    diffs = {k: (got[k] - z["mpjpe_mm"][i]).abs() for i, k in enumerate(("J-Best", "P-Best", "P-Agg"))}
    pick, e3, margin = _j_agg_parts(out20, target, x2d)
    same = pick == z["jagg_pick"].long()
    n = same.sum(dim=(0, 2, 3)).clamp(min=1)
    d = float((((e3 - z["jagg_e3"]) * same).sum(dim=(0, 2, 3)) / n).abs().max()) * 1000
    flipped = ~same
    worst = float(torch.maximum(margin, z["jagg_margin"])[flipped].max()) if bool(flipped.any()) else 0.0
    return pt, diffs, d, float(flipped.double().mean()), worst


@pytest.mark.parametrize("precision", ["bf16x3", "f32", "f16x2"])
def test_g19_metric_config_vs_reference(full160, precision):
    """BASELINE configs[2] - the configuration the metric is quoted on (B=1, P=20, T=10, flip-TTA) - against the output
    of the REFERENCE itself on the same weights, inputs and noise (golden G19, made by tests/golden/make_golden.py from
    /root/reference): three whole trajectories pointwise, and all four MPJPE protocols over ALL 20 hypotheses at every
    step, within 1.25 x the committed measurement of this very comparison.  The golden was computed on the build
    container's CPU; the reference's fp32 arithmetic itself differs between that host and the GPU box's (up to 1.6e-3 mm at
    two timesteps, profiles/r03_host_variation.json), which is what these bounds contain - the oracle run on THIS box
    agrees with the HIP path to 3.7e-4 mm (test_full_size_trajectories_vs_oracle, test_loop_vs_oracle_mpjpe)."""
    from tests.test_hip_parity import PARITY_LINES, parity_bounds
    z = load_golden("g19_metric_config.npz")
    assert gu.sha256_of(full160["sd"]) == z["sha"].numpy().tobytes()
    if precision == "bf16x3":
        out20 = full160["out"][:, :, :20].cpu()        # (bit-equal to a P=20 run: test_p20_t10_equals_its_halves...)
    else:
        from __graft_entry__ import make_model
        model, _ = make_model(20, T_FULL, seed=51)
        model.precision = precision
        noises = [n[:, :20].contiguous() for n in full160["noises"]]
        model.noise_fn = lambda k, shape, device: noises[k]
        out20 = model(full160["x2d"].to(DEV), None, input_2d_flip=full160["x2f"].to(DEV)).cpu()
    pt, diffs, d, frac, worst = g19_compare(out20, z, full160["x2d"])
    assert pt <= 2e-5, pt       # (measured 8.5e-6: 4/5 of it is the two hosts' difference)
    case = f"g19_P20_T10_{precision}"
    bounds, _ = parity_bounds(case, None)              # a golden comparison: nothing about this box' CPU enters
    met = sum(int((v <= 1e-4).sum()) for v in diffs.values())
    PARITY_LINES.append(f"{case} (vs the reference's own run): |dMPJPE| max mm " +
                        ", ".join(f"{k} {float(v.max()):.2e} (<= {bounds[k]:.2e})" for k, v in diffs.items()) +
                        f", J-Agg same picks {d:.2e} (<= {bounds['J-Agg']:.2e}), different picks {frac:.1e} of joints; "
                        f"north_star 1e-4 mm met at {met} of {sum(v.numel() for v in diffs.values())} (step, protocol) pairs")
    for k, v in diffs.items():
        assert v.max() <= bounds[k], (k, float(v.max()), bounds[k])
    assert d <= bounds["J-Agg"] and frac <= 2e-3 and worst <= 1e-4, (d, bounds["J-Agg"], frac, worst)


# ------------------------------------------------------------------------------ index stages, bit for bit (golden G6)
def _selector_embed(J, C=64):
    """patch embedding that copies the five inputs into channels 0..4 (everything else zero): the embedding's
    pre-norm output then IS the gathered / flipped input, exactly."""
    pw = torch.zeros(C, 5)
    pw[torch.arange(5), torch.arange(5)] = 1.0
    return (pw.to(DEV), torch.zeros(C, device=DEV), torch.zeros(J, C, device=DEV), torch.ones(C, device=DEV),
            torch.zeros(C, device=DEV))


def test_embed_gather_and_flip_bit_exact_g6():
    """pafuse_embed on golden G6's integer-valued pose tensor (reference output of the flip / split index ops):
    part gather (split_data, diffusionpose.py:328-335), L/R swap + x negation of the flipped copy (:195-198) and the
    2-D broadcast over hypotheses must be bit-exact."""
    from pafuse_amd import ops
    z = load_golden("g6_index_ops.npz")
    x = z["x"]                                                     # [2,3,4,134,3] integer-valued
    B, P, F, J3, _ = x.shape
    perm = orc.flip_permutation(gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT).to(torch.int32)
    g = torch.Generator().manual_seed(6)
    x2d = torch.randint(-9, 10, (B, F, J3, 2), generator=g).float()
    x2f = x2d[:, :, perm.long()].clone()
    x2f[..., 0] *= -1
    temb = torch.zeros(B, 64, device=DEV)
    for part, idx in orc.PART_JOINTS.items():
        J = len(idx)
        pw, pb, pos, nw, nb = _selector_embed(J)
        joints = torch.tensor(idx, dtype=torch.int32, device=DEV)
        xe, _ = ops.embed(x.to(DEV), x2d.to(DEV), pw, pb, pos, temb, nw, nb, joints=joints, perm=perm.to(DEV),
                          x2d_flip=x2f.to(DEV))
        xe = xe.cpu()                                              # [2(flip), B, P, F, J, 64]
        assert torch.equal(xe[0, ..., 2:5], z[f"split_{part}"])                          # plain half: the part's joints
        assert torch.equal(xe[1, ..., 2:5], z["flipped"][..., idx, :])                   # flipped half: reference flip
        assert torch.equal(xe[0, ..., 0:2], x2d[:, None, :, idx].expand(B, P, F, J, 2))  # 2-D input over P
        assert torch.equal(xe[1, ..., 0:2], x2f[:, None, :, idx].expand(B, P, F, J, 2))
        assert not xe[..., 5:].any()
    # no part list = the whole skeleton (the MixSTE2 unit entry), no flip
    pw, pb, pos, nw, nb = _selector_embed(J3)
    xe, _ = ops.embed(x.to(DEV), x2d.to(DEV), pw, pb, pos, temb, nw, nb)
    assert torch.equal(xe.cpu()[0, ..., 2:5], x)


def test_embed_clamp_and_scale_exact():
    """clamp(+-1.1*scale) / scale of the noised pose (diffusionpose.py:193-194) with scale = 2 on values that are
    exact in fp32 either way."""
    from pafuse_amd import ops
    x = torch.tensor([-8.0, -2.25, -2.0, -0.5, 0.0, 0.75, 2.0, 2.5, 64.0]).repeat(3)[:24].reshape(1, 1, 1, 8, 3)
    x2d = torch.zeros(1, 1, 8, 2)
    pw, pb, pos, nw, nb = _selector_embed(8)
    xe, _ = ops.embed(x.to(DEV), x2d.to(DEV), pw, pb, pos, torch.zeros(1, 64, device=DEV), nw, nb, do_clamp=True, scale=2.0)
    assert torch.equal(xe.cpu()[0, ..., 2:5], torch.clamp(x, min=-1.1 * 2.0, max=1.1 * 2.0) / 2.0)
    # a scale that is not an fp32 number: the bound is float(1.1 * 0.3) (fp64 product), the divisor float(0.3)
    xe, _ = ops.embed(x.to(DEV), x2d.to(DEV), pw, pb, pos, torch.zeros(1, 64, device=DEV), nw, nb, do_clamp=True, scale=0.3)
    assert torch.equal(xe.cpu()[0, ..., 2:5], torch.clamp(x, min=-1.1 * 0.3, max=1.1 * 0.3) / 0.3)


def test_finalize_concat_and_unflip_bit_exact_g6():
    """pafuse_ddim_finalize fed with golden G6's part tensors as the "denoiser outputs" (plain half = split parts,
    flipped half = the reference's flipped tensor, split): concat (diffusionpose.py:171), un-flip (:211-213) and the
    TTA mean (:214-215) must give back the original tensor exactly: the "predictions" are x / 64 (exact, and inside the
    +-1.1 clamp), scale = 64, so x_start = ((x/64 + x/64) / 2) * 64 = x."""
    from pafuse_amd import _lib, ops
    z = load_golden("g6_index_ops.npz")
    x = z["x"]
    B, P, F, J, _ = x.shape
    perm = orc.flip_permutation(gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT).to(torch.int32)
    joint_part = torch.empty(J, dtype=torch.int32)
    joint_local = torch.empty(J, dtype=torch.int32)
    preds = []
    for i, (part, idx) in enumerate(orc.PART_JOINTS.items()):
        joint_part[idx], joint_local[idx] = i, torch.arange(len(idx), dtype=torch.int32)
        preds.append((torch.stack([z[f"split_{part}"], z["flipped"][..., idx, :]]) / 64.0).contiguous().to(DEV))
    st = _lib.DDIMStep()
    st.time, st.last = 0, 1
    img = torch.zeros(B, P, F, J, 3, device=DEV)
    out, img = ops.ddim_finalize(preds, joint_part.to(DEV), joint_local.to(DEV), img, st, flip_perm=perm.to(DEV),
                                 scale=64.0, T=3, step=1)
    assert float(x.abs().max()) / 64.0 < 1.1
    assert torch.equal(out[:, 1].cpu(), x) and torch.equal(img.cpu(), x)
    assert not out[:, 0].any() and not out[:, 2].any()            # only the step's slot is written
    # no flip: plain concat of the first halves
    out, _ = ops.ddim_finalize([p[:1].contiguous() for p in preds], joint_part.to(DEV), joint_local.to(DEV),
                               torch.zeros(B, P, F, J, 3, device=DEV), st, scale=64.0)
    assert torch.equal(out[:, 0].cpu(), z["cat"])


def test_finalize_update_matches_oracle_arithmetic():
    """a non-final step through the unit entry point: eps in fp64 and the img update, against the oracle's formulas on
    random tensors (both samplers' scalar demotion rules)."""
    from pafuse_amd import _lib, ops
    from __graft_entry__ import make_model
    model, sd = make_model(2, 5, device="cpu")
    steps = model.ddim_steps()
    g = torch.Generator().manual_seed(8)
    B, P, F, J = 1, 2, 3, 134
    img0 = torch.randn(B, P, F, J, 3, generator=g)
    noise = torch.randn(B, P, F, J, 3, generator=g)
    pred = torch.randn(B, P, F, J, 3, generator=g) * 0.5
    jp = torch.zeros(J, dtype=torch.int32, device=DEV)
    jl = torch.arange(J, dtype=torch.int32, device=DEV)
    time, time_next = model.time_pairs()[1]
    x0 = torch.clamp(pred * 1.0, -1.1, 1.1)
    eps64 = orc._eps_from_x0(sd, img0, torch.full((B,), time, dtype=torch.long), x0)          # fp64
    sqrt_an, c, sigma = orc.ddim_coefficients(sd["alphas_cumprod"], time, time_next, 1.0)
    want = (x0 * sqrt_an + c * eps64 + sigma * noise).float()                 # ddim_sample (no flip): fp64 chain
    img = img0.clone().to(DEV)
    out, img = ops.ddim_finalize([pred[None].contiguous().to(DEV)], jp, jl, img, steps[1], noise=noise.to(DEV), T=5, step=1)
    assert torch.equal(out[:, 1].cpu(), x0)
    assert torch.allclose(img.cpu(), want, rtol=0, atol=2e-6), (img.cpu() - want).abs().max()


def test_unit_entry_points_refuse_bad_operands():
    from pafuse_amd import _lib, ops
    lib = _lib.load()
    z = torch.zeros(4, device=DEV)
    p = z.data_ptr()
    assert lib.pafuse_embed(p, p, None, None, None, p, p, p, p, p, p, 1e-6, 1, 1, 1, 4, 8, 64, 1, 0, 1.0, p, p, None) == -1
    assert b"joint list" in lib.pafuse_last_error()
    assert lib.pafuse_embed(p, p, None, None, None, p, p, p, p, p, p, 1e-6, 1, 1, 1, 4, 4, 66, 1, 0, 1.0, p, p, None) == -2
    st = _lib.DDIMStep()
    st.last = 1
    arr, cnt = (C.c_void_p * 1)(p), (C.c_int32 * 1)(3)
    assert lib.pafuse_ddim_finalize(arr, cnt, 1, p, p, None, p, None, p, 1, 1, 1, 4, 1, 0, 0, 1.0, C.byref(st), None) == -2
    assert b"cover" in lib.pafuse_last_error()
    with pytest.raises(ValueError):
        ops.embed(torch.zeros(1, 1, 1, 4, 3, device=DEV), torch.zeros(1, 1, 5, 2, device=DEV), *_selector_embed(4)[:3],
                  torch.zeros(1, 64, device=DEV), *_selector_embed(4)[3:])


# ------------------------------------------------------------------------- general.part_based_model = False (G17)
@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_g17_single_model_variant_golden(precision):
    """One MixSTE2 over all 134 keypoints at width 288 (common/diffusionpose.py:150-153): 134-token spatial sequences
    (the 144-key attention tiles), whole-row kernels of width 288; against the reference's own outputs."""
    from __graft_entry__ import make_model
    z = load_golden("g17_single_model.npz")
    model, sd = make_model(2, 2, seed=171, part_based=False)
    assert gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    model.precision = precision
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=2, n=2, seed=17)
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    assert out.shape == z["flip_out"].shape
    assert torch.allclose(out, z["flip_out"], rtol=0, atol=1e-5), (out - z["flip_out"]).abs().max()
    # no-flip sampler, P = 1, T = 1
    m1, _ = make_model(1, 1, seed=171, part_based=False, flip=False)
    m1.precision = precision
    n1 = gu.synthetic_noises(B=1, P=1, n=1, seed=18)
    m1.noise_fn = lambda k, shape, device: n1[k]
    o1 = m1(x2d.to(DEV), None).cpu()
    assert torch.allclose(o1, z["noflip_out"], rtol=0, atol=1e-5), (o1 - z["noflip_out"]).abs().max()


def test_single_model_variant_vs_oracle_and_hypothesis_independence():
    from __graft_entry__ import make_model
    model, sd = make_model(4, 3, seed=172, part_based=False)
    x2d, x2f = gu.synthetic_inputs_2d(B=2)
    noises = gu.synthetic_noises(B=2, P=4, n=3, seed=19)
    model.noise_fn = lambda k, shape, device: noises[k]
    out = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    ref = orc.ddim_sample(sd, x2d, noises, 3, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f,
                          part_joints=orc.SINGLE_MODEL)
    assert (out.cpu() - ref).abs().max() <= 1e-5
    try:
        model.proposal_shard = (1, 3)                     # a rank's slice of the hypothesis axis: same bits
        part = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    finally:
        model.proposal_shard = None
    assert torch.equal(part, out[:, :, 1:3])


def test_single_model_variant_training_is_refused_loudly():
    """attn_backward_kernel keeps one sequence in LDS (<= 80 tokens): the 134-joint model is inference only and says so."""
    from __graft_entry__ import make_model
    from pafuse_amd._lib import PafuseError
    model, _ = make_model(1, 1, seed=173, part_based=False, is_train=True)
    x2d, _ = gu.synthetic_inputs_2d(B=1)
    with pytest.raises(PafuseError, match="inference only"):
        model(x2d.to(DEV), torch.zeros(1, 27, 134, 3, device=DEV))


def test_grouped_launches_equal_part_by_part_launches():
    """The single-stream schedule (no aux streams) puts the same layer of the three parts into shared grids
    (grouped_*_kernel).  A tile's products and sums do not depend on the grid it runs in: the same loop with every layer
    launched part by part (pafuse_d3dp_config.part_by_part_launches, a per-call option: the library keeps no process-wide
    schedule state) is the same function, BIT FOR BIT.  The per-part whole-row launches write their rows through per-wave LDS
    slabs (epilogue_rows_h) while the shared grid keeps the direct epilogue (three tile shapes in one kernel: the slab form
    spilled there): the same arithmetic in the same order, and since round 6 every multiply-add of both is an explicit fmaf, so
    the compiler cannot contract the two code shapes differently (round 5 had left that to chance: 4e-6 apart)."""
    from __graft_entry__ import make_model
    model, sd = make_model(20, 2, seed=52)
    model.precision = "bf16x3"       # (the shared grids are the single-stream schedule of the round-3 kernels)
    model.n_aux_streams = 0
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=20, n=2, seed=7)
    model.noise_fn = lambda k, shape, device: noises[k]
    grouped = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    model.part_by_part_launches = True
    part_by_part = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV)).cpu()
    assert torch.equal(grouped, part_by_part), float((grouped - part_by_part).abs().max())
    sub = [0, 7, 19]
    ref = orc.ddim_sample(sd, x2d, [n[:, sub] for n in noises], 2, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    assert float((grouped[:, :, sub] - ref).abs().max()) <= 1e-5 and float((part_by_part[:, :, sub] - ref).abs().max()) <= 1e-5


@pytest.mark.parametrize("precision", ["bf16x3", "f32", "f16x2"])
def test_side_streams_return_the_single_stream_bits(precision):
    """The default schedule runs the three parts on three HIP streams (queues).  In round 2 that was wrong now and then
    in the bf16-MFMA modes; the cause - packed-fp32 VALU instructions beside v_mfma_f32_32x32x16_bf16 waves of another
    queue, profiles/r03_bf16_mfma_concurrency.md - is compiled out (pafuse_amd/build_flags.py).  The metric's configuration
    (P=20, T=10), twenty times on three and on six streams: every run equals the single-stream run bit for bit."""
    import ctypes
    from __graft_entry__ import make_model
    from pafuse_amd import _lib
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=20, n=T_FULL, seed=21)
    x2d, x2f = x2d.to(DEV), x2f.to(DEV)
    outs = {}
    for aux in (0, 2, 5):
        model, _ = make_model(20, T_FULL, seed=52)
        model.precision, model.n_aux_streams = precision, aux
        # (the single-stream reference runs its DEFAULT schedule - the shared whole-row grids - since round 6: their direct epilogue
        # and the per-part launches' slab epilogue return the same bits again, test_grouped_launches_equal_part_by_part_launches)
        model.noise_fn = lambda k, shape, device: noises[k]
        lanes = _lib.check(_lib.load().pafuse_d3dp_lanes(ctypes.byref(model.config_struct(True)), 1, 20, aux))
        assert lanes == aux + 1, (aux, lanes)
        runs = [model(x2d, None, input_2d_flip=x2f) for _ in range(20 if aux else 1)]
        outs[aux] = runs
    single = outs[0][0]
    for aux in (2, 5):
        differ = sum(int(not torch.equal(o, single)) for o in outs[aux])
        assert differ == 0, f"{differ} of {len(outs[aux])} runs on {aux + 1} streams differ from the single-stream run"


def test_bench_two_rank_rehearsal_on_one_gpu(tmp_path):
    """bench.py's N > 1 code path (rank census, hypothesis sharding, the all-gather and its timing, max-over-ranks
    clock) run as two real ranks, started the way the driver starts a scaling run (`python bench.py --gpus 2`: the parent spawns
    torch.distributed.run before it touches the GPU and relays rank 0's line) - sharing this box's single GPU over gloo, which is a
    rehearsal of the code path, never a performance number (the line says so).  The gathered predictions of the last
    step must EQUAL, bit for bit, what ONE process computes for all P hypotheses from the same seed: every rank draws
    the full-P noise and keeps its slice, the gather puts the slices back in hypothesis order."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from __graft_entry__ import make_model
    from tests.conftest import ROOT
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dump = str(tmp_path / "gathered.pt")
    # the driver's own spelling: `python bench.py --gpus N ...` - bench.py starts its ranks itself (a child torch.distributed.run)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--single-device", "--steps", "1", "--warmup", "1", "--proposals", "2", "--timesteps", "2",
           "--no-cpu-baseline", "--no-roofline", "--dump-output", dump]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                              # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["P_local_per_rank"] == [2, 2]
    assert line["config"]["P_total"] == 4 and line["allgather_ms"] is not None and line["allgather_ms"] > 0
    assert line["gather_copy_ms"] is not None and line["gather_copy_ms"] <= line["allgather_ms"] * 1.5 + 1.0
    assert line["scaling"] == "weak" and line["value"] > 0 and line["config"]["single_device_rehearsal"] is True
    # the same job in ONE process: same weights (seed 51), same generator seed, the second forward (warm-up + 1 step)
    model, _ = make_model(4, 2, seed=51)
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    torch.manual_seed(1234)
    for _ in range(2):
        single = model(x2d.to(DEV), None, input_2d_flip=x2f.to(DEV))
    gathered = torch.load(dump)
    assert gathered.shape == single.shape == (1, 2, 4, 27, 134, 3)
    assert torch.equal(gathered, single.cpu())
