"""The oracle (oracle/d3dp_oracle.py) against every golden vector captured from the real reference.

These pins are what allows the GPU parity tests to use the oracle as the checker at other sizes/seeds.
"""
import numpy as np
import pytest
import torch

from oracle import d3dp_oracle as orc
from tests.golden import golden_util as gu
from tests.conftest import load_golden


def _sub(z, prefix):
    return {k[len(prefix):]: v for k, v in z.items() if k.startswith(prefix)}


def test_g1_tiny_mixste_bit_level():
    z = load_golden("g1_tiny_mixste.npz")
    sd = _sub(z, "sd.")
    out = orc.mixste2_eval(sd, "", z["x2d"], z["x3d"], z["t"], depth=2, heads=8)
    assert out.shape == z["out"].shape
    assert torch.allclose(out, z["out"], rtol=0, atol=2e-6), (out - z["out"]).abs().max()


def test_g2_schedule_exact():
    z = load_golden("g2_schedule.npz")
    bufs = orc.schedule_buffers(1000)
    ref = _sub(z, "buf.")
    assert set(ref) == set(bufs) and len(bufs) == 12
    for k, v in bufs.items():
        assert v.dtype == torch.float64
        assert torch.allclose(v, ref[k], rtol=1e-13, atol=0), k        # bit-exact on the fixture's host
    for T in (1, 2, 5, 10, 20, 50):
        pairs = orc.ddim_time_pairs(1000, T)
        assert pairs == [tuple(p) for p in z[f"pairs.{T}"].tolist()]
        coefs = [[float(x) for x in orc.ddim_coefficients(bufs["alphas_cumprod"], a, b)]
                 for a, b in pairs if b >= 0]
        assert np.allclose(np.asarray(coefs).reshape(-1, 3), z[f"coefs.{T}"].numpy(), rtol=1e-12, atol=0)
    assert orc.ddim_time_pairs(1000, 10)[0] == (999, 899) and orc.ddim_time_pairs(1000, 10)[-1] == (99, -1)


def test_g3_time_mlp():
    z = load_golden("g3_time_mlp.npz")
    for part, C in gu.PART_WIDTH.items():
        shapes = {"1.weight": (2 * C, C), "1.bias": (2 * C,), "3.weight": (C, 2 * C), "3.bias": (C,)}
        sd = {k: gu.seeded_tensor(f"g3.{part}.time_mlp.{k}", s, 31) for k, s in shapes.items()}
        assert gu.sha256_of(sd) == z[f"{part}.sha"].numpy().tobytes()
        arg = z["t"][:, None] * orc.sinusoid_frequencies(C)[None, :]
        assert torch.equal(torch.cat((arg.sin(), arg.cos()), -1), z[f"{part}.sin"])
        out = orc.timestep_embedding({"time_mlp." + k: v for k, v in sd.items()}, "", z["t"], C)
        assert torch.allclose(out, z[f"{part}.out"], rtol=0, atol=1e-6)


def test_g4_blocks():
    z = load_golden("g4_blocks.npz")
    for part, C in gu.PART_WIDTH.items():
        shapes = {"norm1.weight": (C,), "norm1.bias": (C,), "attn.qkv.weight": (3 * C, C), "attn.qkv.bias": (3 * C,),
                  "attn.proj.weight": (C, C), "attn.proj.bias": (C,), "norm2.weight": (C,), "norm2.bias": (C,),
                  "mlp.fc1.weight": (2 * C, C), "mlp.fc1.bias": (2 * C,), "mlp.fc2.weight": (C, 2 * C),
                  "mlp.fc2.bias": (C,)}
        sd = {k: gu.seeded_tensor(f"g4.{part}.{k}", s, 41) for k, s in shapes.items()}
        assert gu.sha256_of(sd) == z[f"{part}.sha"].numpy().tobytes()
        for tag in ("s", "t"):
            y = orc.transformer_block(sd, "", z[f"{part}.x{tag}"], heads=8)
            assert torch.allclose(y, z[f"{part}.y{tag}"], rtol=0, atol=5e-6), (part, tag)


def _g5_state_dict(z):
    from tests.golden.state_template import d3dp_template
    sd = gu.seeded_state_dict(d3dp_template(), seed=51)
    assert gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    return sd


def test_g5_d3dp_loops():
    z = load_golden("g5_d3dp.npz")
    sd = _g5_state_dict(z)
    assert len(sd) == 636
    x2d, x2d_flip = gu.synthetic_inputs_2d(B=1)
    assert torch.equal(x2d, z["flip_x2d"]) and torch.equal(x2d_flip, z["flip_x2d_flip"])
    # single denoiser pass per part
    t = torch.tensor([499])
    for part, idx in orc.PART_JOINTS.items():
        out = orc.mixste2_eval(sd, f"pose_estimator.{part}.", x2d[..., idx, :], z["part_x3d"][..., idx, :], t)
        assert torch.allclose(out, z[f"part.{part}"], rtol=0, atol=2e-5), part
    # flip-TTA loop P=2, T=2
    noises = gu.synthetic_noises(B=1, P=2, n=2, seed=1)
    out = orc.ddim_sample(sd, x2d, noises, 2, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2d_flip)
    assert out.shape == (1, 2, 2, 27, 134, 3) and out.dtype == torch.float32
    assert torch.allclose(out, z["flip_out"], rtol=0, atol=2e-5), (out - z["flip_out"]).abs().max()
    # P=1, T=1 both samplers (BASELINE config 1: CPU plumbing)
    n1 = gu.synthetic_noises(B=1, P=1, n=1, seed=2)
    o_nf = orc.ddim_sample(sd, x2d, n1, 1)
    assert torch.allclose(o_nf, z["noflip_out"], rtol=0, atol=2e-5)
    o_f = orc.ddim_sample(sd, x2d, n1, 1, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2d_flip)
    assert torch.allclose(o_f, z["flip11_out"], rtol=0, atol=2e-5)


def test_g6_index_ops_bit_exact():
    z = load_golden("g6_index_ops.npz")
    x = z["x"]
    perm = orc.flip_permutation(gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT)
    flipped = x[:, :, :, perm].clone()
    flipped[..., 0] *= -1
    assert torch.equal(flipped, z["flipped"])
    assert torch.equal(perm[perm], torch.arange(134))                     # an involution
    parts = [x[..., idx, :] for idx in orc.PART_JOINTS.values()]
    assert torch.equal(parts[0], z["split_body"]) and torch.equal(parts[1], z["split_face"])
    assert torch.equal(parts[2], z["split_hands"])
    assert torch.equal(torch.cat(parts, dim=-2), z["cat"]) and torch.equal(z["cat"], x)
    assert torch.equal(orc.center_pose_parts(z["pose"].clone()), z["centred"])
    wb_in = z["pose"].clone()
    assert torch.equal(orc.wb_pose_from_parts(wb_in), z["wb_out"])
    assert torch.equal(wb_in, z["wb_in_after"])                           # the reference's in-place side effect
    assert torch.all(z["wb_out"][..., 0, :] == 0)


def test_g7_metrics():
    z = load_golden("g7_metrics.npz")
    pred, target = z["pred"], z["target"]
    B, T, P, F = pred.shape[:4]
    absolute = (pred + z["traj"][:, None, None]).reshape(B * T * P * F, 134, 3)
    reproj = orc.project_to_2d(absolute, z["cam"].repeat(B * T * P * F, 1)).reshape(B, T, P, F, 134, 2)
    assert torch.allclose(reproj, z["reproj"], rtol=0, atol=1e-6)
    assert torch.allclose(orc.j_best(pred, target), z["j_best"], rtol=0, atol=1e-7)
    assert torch.allclose(orc.p_best(pred, target), z["p_best"], rtol=0, atol=1e-7)
    assert torch.allclose(orc.p_agg(pred, target), z["p_agg"], rtol=0, atol=1e-7)
    assert torch.allclose(orc.j_agg(pred, target, z["reproj"], z["target_2d"]), z["j_agg"], rtol=0, atol=1e-7)


def test_g9_evaluate_accumulators():
    z = load_golden("g9_evaluate.npz")
    got = orc.evaluate_accumulators(z["pred_parts"], z["gt_parts"], z["x2d"], z["traj"], z["cam"])
    keys = [k for k in z if k not in ("pred_parts", "gt_parts", "x2d", "traj", "cam")]
    assert len(keys) == 14 and set(keys) == set(got)
    for k in keys:
        assert torch.allclose(got[k], z[k], rtol=0, atol=1e-7), (k, got[k], z[k])


def test_g11_scale():
    from tests.golden.state_template import d3dp_template
    z = load_golden("g11_scale.npz")
    sd = gu.seeded_state_dict(d3dp_template(), seed=111)
    assert gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = [n * 1.5 for n in gu.synthetic_noises(B=1, P=2, n=2, seed=12)]
    out = orc.ddim_sample(sd, x2d, noises, 2, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f, scale=2.0)
    assert torch.allclose(out, z["out"], rtol=0, atol=2e-5), (out - z["out"]).abs().max()


def test_g17_single_model_variant():
    """general.part_based_model = False (one MixSTE2 over the 134 keypoints, width cs = 288): the module's state dict has
    the reference's keys and shapes (the seeded weights hash to the fixture's digest), and the oracle reproduces the
    reference's flip (P=2, T=2) and no-flip (P=1, T=1) loops."""
    from __graft_entry__ import make_model
    z = load_golden("g17_single_model.npz")
    model, sd = make_model(2, 2, seed=171, device="cpu", part_based=False)
    assert len(sd) == int(z["n_keys"]) and gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    assert all(k.startswith("pose_estimator.") and k.split(".")[1] not in ("body", "face", "hands")
               for k, v in sd.items() if v.dtype == torch.float32)
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    out = orc.ddim_sample(sd, x2d, gu.synthetic_noises(B=1, P=2, n=2, seed=17), 2, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT,
                          inputs_2d_flip=x2f, part_joints=orc.SINGLE_MODEL)
    assert torch.allclose(out, z["flip_out"], rtol=0, atol=2e-5), (out - z["flip_out"]).abs().max()
    o1 = orc.ddim_sample(sd, x2d, gu.synthetic_noises(B=1, P=1, n=1, seed=18), 1, part_joints=orc.SINGLE_MODEL)
    assert torch.allclose(o1, z["noflip_out"], rtol=0, atol=2e-5), (o1 - z["noflip_out"]).abs().max()


G18_KW = dict(num_frame=9, num_joints=17, in_chans=5, embed_dim_ratio=64, depth=2, num_heads=8, mlp_ratio=3.,
              qkv_bias=False, qk_scale=0.3, drop_rate=0.1, attn_drop_rate=0.2, is_train=False)


def test_g18_mixste2_constructor_options():
    """mlp_ratio=3, qkv_bias=False, qk_scale=0.3, dropout rates (common/mixste.py:141-144; identity in eval): the module
    mirror has the reference's state-dict layout (no qkv bias keys, [3C] hidden) and the oracle reproduces the
    reference's eval forward."""
    import pafuse_amd
    z = load_golden("g18_mixste_options.npz")
    m = pafuse_amd.MixSTE2(**G18_KW)
    sd = gu.seeded_state_dict(m.state_dict(), seed=181)
    assert len(sd) == int(z["n_keys"]) and gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    assert not any(k.endswith("qkv.bias") for k in sd) and sd["STEblocks.0.mlp.fc1.weight"].shape == (192, 64)
    out = orc.mixste2_eval(sd, "", z["x2d"], z["x3d"], z["t"], depth=2, heads=8, qk_scale=0.3)
    assert torch.equal(out, z["out"]) or torch.allclose(out, z["out"], rtol=0, atol=2e-6), (out - z["out"]).abs().max()
    with pytest.raises(NotImplementedError):            # hidden width 80 is no multiple of 32
        pafuse_amd.MixSTE2(embed_dim_ratio=64, mlp_ratio=1.25)


# ------------------------------------------------------------------------------------------------ training (n2)
def drops_from_tape(tape, rates):
    """[(attn, mlp)] per block in execution order from the factors in the order the reference drew them."""
    tape, out = list(tape), []
    for r in rates:
        for _ in range(2):                              # STE_i then TTE_i share dpr[i]
            out.append((tape.pop(0), tape.pop(0)) if r > 0 else (None, None))
    assert not tape
    return out


def grad_stats(g):
    flat = g.reshape(-1).double()
    head = torch.zeros(8, dtype=torch.float64)
    head[:min(8, flat.numel())] = flat[:8]
    return torch.cat([flat.sum()[None], flat.norm()[None], head])


def test_g12_train_mode_forward_and_gradients():
    """train-mode MixSTE2 with DropPath: forward and every parameter gradient of the oracle (torch autograd over the
    functional restatement) against the reference's."""
    z = load_golden("g12_train_tiny.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sub(z, "sd.").items()}
    rates = orc.drop_path_rates(0.5, 2)
    assert rates == [0.0, 0.5]
    drops = drops_from_tape([z[f"drop.{i}"] for i in range(int(z["n_drop"]))], rates)
    out = orc.mixste2_train(sd, "", z["x2d"], z["x3d"], z["t"], depth=2, heads=8, drop=drops)
    assert torch.allclose(out, z["out"], rtol=0, atol=2e-6), (out - z["out"]).abs().max()
    out.backward(z["dout"])
    ref = _sub(z, "grad.")
    assert set(ref) == set(sd)
    for k, g in ref.items():
        tol = 1e-5 * float(g.abs().max()) + 1e-7
        assert torch.allclose(sd[k].grad, g, rtol=1e-4, atol=tol), (k, (sd[k].grad - g).abs().max())
    # the factors themselves follow timm's rule: 0 or 1/keep
    for i in range(int(z["n_drop"])):
        assert set(z[f"drop.{i}"].tolist()) <= {0.0, 2.0}
    torch.manual_seed(5)
    drawn = orc.draw_drop_path(0.5, 2, B=4, Fr=3, J=5, like=torch.zeros(1))
    assert [None if a is None else tuple(a.shape) for a, _ in drawn] == [None, None, (12,), (20,)]


def test_g13_d3dp_train_forward_loss_and_gradient_statistics():
    z = load_golden("g13_d3dp_train.npz")
    from tests.golden.state_template import d3dp_template
    sd = gu.seeded_state_dict(d3dp_template(depth=1), seed=131)
    assert gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    x2d, _ = gu.synthetic_inputs_2d(B=2)
    target = gu.synthetic_target_3d(B=2)
    t = z["t"].reshape(-1)
    x_poses = orc.q_sample_targets(sd, target, t, z["noise"], scale=1.0)
    assert torch.equal(x_poses, z["x_poses"])
    assert int(z["n_drop"]) == 0 and orc.drop_path_rates(0.1, 1) == [0.0]     # depth 1: linspace(0, 0.1, 1) = [0]
    leaves = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 else v) for k, v in sd.items()}
    pred = orc.train_forward(leaves, x2d, x_poses, t, depth=1, heads=8)
    assert torch.allclose(pred, z["pred"], rtol=0, atol=2e-6), (pred - z["pred"]).abs().max()
    loss = orc.mpjpe(pred, target)
    assert torch.allclose(loss, z["loss"], rtol=1e-6, atol=0)
    loss.backward()
    for k, ref in _sub(z, "gstat.").items():
        got = grad_stats(leaves[k].grad)
        scale = float(ref[1]) + 1e-12                     # the gradient's L2 norm
        assert abs(float(got[1] - ref[1])) <= 1e-4 * scale, (k, got[1], ref[1])
        assert torch.allclose(got[2:], ref[2:], rtol=1e-3, atol=1e-5 * scale), k


def test_g19_oracle_equals_the_reference_on_the_metric_configuration():
    """golden G19 = the reference's own run of BASELINE configs[2] (P=20, T=10, flip-TTA): the oracle re-computes the three
    stored hypotheses through all ten steps (a hypothesis' trajectory depends on nothing but its own noise draws) and
    must reproduce them (bit for bit here; atol slack for another host BLAS)."""
    from tests.golden.state_template import d3dp_template
    z = load_golden("g19_metric_config.npz")
    sd = gu.seeded_state_dict(d3dp_template(), seed=51)
    assert gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    sub = [int(i) for i in z["sub"]]
    noises = [n[:, sub].contiguous() for n in gu.synthetic_noises(B=1, P=160, n=10, seed=160)]
    out = orc.ddim_sample(sd, x2d, noises, 10, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f)
    assert out.shape == z["out_sub"].shape == (1, 10, 3, 27, 134, 3)
    assert torch.allclose(out, z["out_sub"], rtol=0, atol=2e-6), (out - z["out_sub"]).abs().max()
