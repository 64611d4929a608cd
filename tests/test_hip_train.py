"""GPU parity tests of the training path: train-mode MixSTE2 / D3DP forward and the
gradient of every parameter, HIP kernels through the C ABI against the reference's autograd (golden G12, G13) and the
oracle's autograd at other sizes.

Tolerances: forward as in test_hip_parity (1e-5 pointwise); gradients are sums over 1e3-1e5 rows of fp32 products in
a different (but fixed) order than ATen's, compared relative to the tensor's largest entry: 2e-4 * max|g|.
"""
import pytest
import torch

from oracle import d3dp_oracle as orc
from tests.conftest import load_golden
from tests.golden import golden_util as gu
from tests.test_oracle_golden import _sub, drops_from_tape, grad_stats

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, want, what, rel=2e-4):
    tol = rel * float(want.abs().max()) + 1e-7
    err = float((got.cpu() - want).abs().max())
    assert err <= tol, (what, err, tol)


def _tiny(z, drop_rate=0.5):
    import pafuse_amd
    m = pafuse_amd.MixSTE2(num_frame=3, num_joints=5, in_chans=5, embed_dim_ratio=64, depth=2, num_heads=8,
                           mlp_ratio=2.0, qkv_bias=True, qk_scale=None, drop_path_rate=drop_rate, is_train=True)
    m.load_state_dict(_sub(z, "sd."))
    return m.to(DEV).train()


def test_g12_train_tiny_golden():
    """forward (with the DropPath factors the reference drew) and all 112 parameter gradients vs the reference."""
    z = load_golden("g12_train_tiny.npz")
    m = _tiny(z)
    drops = drops_from_tape([z[f"drop.{i}"] for i in range(int(z["n_drop"]))], orc.drop_path_rates(0.5, 2))
    m.drop_fn = lambda block, branch, nseq, rate: drops[block][branch]
    out = m(z["x2d"].to(DEV), z["x3d"].to(DEV), z["t"].to(DEV))
    assert torch.allclose(out.detach().cpu(), z["out"], rtol=0, atol=1e-5), (out.cpu() - z["out"]).abs().max()
    out.backward(z["dout"].to(DEV))
    ref = _sub(z, "grad.")
    assert set(ref) == {n for n, _ in m.named_parameters()}
    for n, p in m.named_parameters():
        _close(p.grad, ref[n], n)


def test_drop_path_draws_follow_the_reference_order():
    """without a hook the module draws its own factors: nothing for rate-0 blocks, values in {0, 1/keep}, one per
    sequence of the block's batch axis; eval() switches DropPath off."""
    z = load_golden("g12_train_tiny.npz")
    m = _tiny(z)
    torch.manual_seed(3)
    d = m.drop_path_factors(4, torch.device(DEV))
    assert d.shape == (4, 2, 4 * 5)
    assert torch.all(d[:2] == 1)                                          # dpr[0] = 0: nn.Identity
    assert set(d[2, :, :12].unique().tolist()) <= {0.0, 2.0} and set(d[3].unique().tolist()) <= {0.0, 2.0}
    assert torch.all(d[2, :, 12:] == 1)                                   # spatial blocks have B*F = 12 sequences
    m.eval()
    assert m.drop_path_factors(4, torch.device(DEV)) is None


def test_g13_d3dp_train_golden():
    """D3DP.forward in train mode at the real widths: q_sample bit-exact, prediction, loss, gradient statistics."""
    from __graft_entry__ import make_model
    z = load_golden("g13_d3dp_train.npz")
    model, sd = make_model(1, 1, seed=131, is_train=True, depth=1)
    assert gu.sha256_of(sd) == z["sha"].numpy().tobytes()
    x2d, _ = gu.synthetic_inputs_2d(B=2)
    target = gu.synthetic_target_3d(B=2)
    model.train_draw_fn = lambda i: (z["t"][i], z["noise"][i])
    x_poses, noise, t = model.prepare_targets(target.to(DEV))
    assert torch.equal(x_poses.cpu(), z["x_poses"]) and torch.equal(t.cpu(), z["t"])
    pred = model(x2d.to(DEV), target.to(DEV))
    assert torch.allclose(pred.detach().cpu(), z["pred"], rtol=0, atol=1e-5), (pred.cpu() - z["pred"]).abs().max()
    loss = orc.mpjpe(pred, target.to(DEV))                               # the caller's loss (main_h3wb.py:851)
    assert torch.allclose(loss.detach().cpu(), z["loss"], rtol=1e-5, atol=0)
    loss.backward()
    ref = _sub(z, "gstat.")
    params = dict(model.named_parameters())
    assert set(ref) == set(params)
    for k, want in ref.items():
        got = grad_stats(params[k].grad.cpu())
        scale = float(want[1]) + 1e-12
        assert abs(float(got[1] - want[1])) <= 2e-4 * scale, (k, got[1], want[1])
        assert torch.allclose(got[2:], want[2:], rtol=2e-3, atol=2e-4 * scale), k


@pytest.mark.parametrize("part,B,depth,rate,rel,precision",
                         [("body", 3, 2, 0.3, 2e-4, "f32"), ("face", 2, 2, 0.3, 2e-4, "f32"), ("hands", 2, 2, 0.3, 2e-4, "f32"),
                          ("body", 37, 8, 0.1, 5e-4, "f32"), ("face", 37, 8, 0.1, 5e-4, "f32"), ("hands", 37, 8, 0.1, 5e-4, "f32"),
                          # split-precision products in the plain GEMMs of training (qkv, fc1, every dX): the same bounds
                          ("body", 3, 2, 0.3, 2e-4, "bf16x3"), ("face", 2, 2, 0.3, 2e-4, "bf16x3"),
                          ("hands", 2, 2, 0.3, 2e-4, "bf16x3"), ("body", 37, 8, 0.1, 5e-4, "bf16x3"),
                          ("face", 37, 8, 0.1, 5e-4, "bf16x3"), ("hands", 37, 8, 0.1, 5e-4, "bf16x3")])
def test_train_gradients_vs_oracle_real_widths(part, B, depth, rate, rel, precision):
    """one part at its real width, DropPath active with seeded factors, against torch autograd over the oracle on the
    CPU: depth 2 at small batches for every part, and BASELINE configs[4]'s own size for EVERY part's denoiser - depth 8,
    B = 37 clips (1024 // 27, main_h3wb.py:781), drop_path_rate 0.1 (diffusionpose.py:147): 23 976 / 67 932 / 41 958
    tokens, every one of the 208 parameter gradients (sums over up to 68 k rows: 5e-4 of the tensor's largest entry)."""
    import pafuse_amd
    J, C = len(gu.PART_JOINTS[part]), gu.PART_WIDTH[part]
    m = pafuse_amd.MixSTE2(num_frame=27, num_joints=J, in_chans=5, embed_dim_ratio=C, depth=depth, num_heads=8,
                           drop_path_rate=rate, is_train=True)
    sd = {k: gu.seeded_tensor(k, v.shape, 77) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    m.operand_bf16 = {"f32": 0, "bf16x3": 2}[precision]
    g = torch.Generator().manual_seed(78)
    drops = []
    for r in orc.drop_path_rates(rate, depth):
        for nseq in (B * 27, B * J):
            drops.append(tuple((torch.rand(nseq, generator=g) < 1 - r).float() / (1 - r) if r > 0 else None
                               for _ in range(2)))
    m.drop_fn = lambda block, branch, nseq, rate: drops[block][branch]
    x2d = torch.rand(B, 27, J, 2, generator=g) * 2 - 1
    x3d = torch.randn(B, 27, J, 3, generator=g)
    t = torch.tensor([999, 250, 3][:B]) if B <= 3 else torch.randint(0, 1000, (B,), generator=g)
    dout = torch.randn(B, 27, J, 3, generator=g)
    out = m(x2d.to(DEV), x3d.to(DEV), t.to(DEV))
    out.backward(dout.to(DEV))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = orc.mixste2_train(leaves, "", x2d, x3d, t, depth=depth, heads=8, drop=drops)
    assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=0, atol=1e-5), (out.cpu() - ref).abs().max()
    ref.backward(dout)
    assert len(leaves) == len(list(m.named_parameters())) == 16 + 24 * depth
    for n, p in m.named_parameters():
        _close(p.grad, leaves[n].grad, n, rel)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_d3dp_train_step_full_size_vs_oracle(precision):
    """BASELINE configs[4]'s step at its own size: the three-part D3DP in train mode, depth 8, B = 37 clips, DropPath 0.1
    with seeded factors, per-sample (t, noise) draws -> q_sample -> the parts' train-mode denoisers -> mpjpe loss -> backward
    (common/diffusionpose.py:346-388, main_h3wb.py:781,850-871), against torch autograd over the oracle on the CPU: the noised
    poses bit for bit, the prediction pointwise, the loss, and every one of the 624 parameter gradients.  (The oracle side
    runs part by part - the parts meet only in the loss' mean over joints - so that its autograd graph stays in memory.)"""
    from __graft_entry__ import make_model
    B, depth, rate = 37, 8, 0.1
    model, sd = make_model(1, 1, seed=132, is_train=True, depth=depth)
    model.precision = precision
    x2d, _ = gu.synthetic_inputs_2d(B=B)
    target = gu.synthetic_target_3d(B=B)
    g = torch.Generator().manual_seed(133)
    draws = [(torch.randint(0, 1000, (1,), generator=g), torch.randn(27, 134, 3, generator=g)) for _ in range(B)]
    model.train_draw_fn = lambda i: draws[i]
    drops = {}
    for part, m in model.pose_estimator.items():
        J = m.num_joints
        assert abs(m.drop_path_rate - rate) < 1e-12 and m.block_depth == depth
        d = []
        for r in orc.drop_path_rates(rate, depth):
            for nseq in (B * 27, B * J):
                d.append(tuple((torch.rand(nseq, generator=g) < 1 - r).float() / (1 - r) if r > 0 else None for _ in range(2)))
        drops[part] = d
        m.drop_fn = lambda block, branch, nseq, rate_, d=d: d[block][branch]
    pred = model(x2d.to(DEV), target.to(DEV))
    loss = orc.mpjpe(pred, target.to(DEV))
    loss.backward()
    # ---- the oracle
    t = torch.stack([d[0] for d in draws]).squeeze(-1)
    noise = torch.stack([d[1] for d in draws])
    x_poses = orc.q_sample_targets(sd, target, t, noise)
    x_hip, _, _ = model.prepare_targets(target.to(DEV))
    assert torch.equal(x_hip.cpu(), x_poses)
    with torch.no_grad():
        ref = orc.train_forward(sd, x2d, x_poses, t, depth=depth, drops=[drops[p] for p in orc.PART_JOINTS])
    assert torch.allclose(pred.detach().cpu(), ref, rtol=0, atol=1e-5), (pred.detach().cpu() - ref).abs().max()
    ref_leaf = ref.clone().requires_grad_(True)
    ref_loss = orc.mpjpe(ref_leaf, target)
    assert torch.allclose(loss.detach().cpu(), ref_loss.detach(), rtol=1e-5, atol=0), (float(loss), float(ref_loss))
    ref_loss.backward()
    params = dict(model.named_parameters())
    checked = 0
    for part, idx in orc.PART_JOINTS.items():
        pre = f"pose_estimator.{part}."
        leaves = {k: (v.clone().requires_grad_(True) if k.startswith(pre) else v) for k, v in sd.items()}
        out = orc.mixste2_train(leaves, pre, x2d[..., idx, :], x_poses[..., idx, :], t, depth, 8, drop=drops[part])
        out.backward(ref_leaf.grad[..., idx, :])
        for k, v in leaves.items():
            if k.startswith(pre):
                _close(params[k].grad, v.grad, k, 5e-4)
                checked += 1
        del leaves, out
    assert checked == len(params) == 3 * (16 + 24 * depth)


def test_train_backward_is_bit_reproducible():
    """run to run, and with the weight-gradient GEMMs forked onto a side stream"""
    z = load_golden("g12_train_tiny.npz")
    grads = []
    for side in (False, False, True):
        m = _tiny(z, drop_rate=0.0)
        m.use_side_stream = side
        out = m(z["x2d"].to(DEV), z["x3d"].to(DEV), z["t"].to(DEV))
        out.backward(z["dout"].to(DEV))
        grads.append({n: p.grad.clone() for n, p in m.named_parameters()})
    torch.cuda.synchronize()
    assert all(torch.equal(grads[0][n], grads[1][n]) and torch.equal(grads[0][n], grads[2][n]) for n in grads[0])


def test_optimizer_step_reduces_the_loss():
    """a few AdamW steps (lr 6e-5 * 10, the caller's optimiser: main_h3wb.py:761) on one batch lower the mpjpe."""
    from __graft_entry__ import make_model
    model, _ = make_model(1, 1, seed=141, is_train=True, depth=1)
    x2d, _ = gu.synthetic_inputs_2d(B=2)
    target = gu.synthetic_target_3d(B=2).to(DEV)
    g = torch.Generator().manual_seed(5)
    draws = [(torch.tensor([500]), torch.randn(27, 134, 3, generator=g)) for _ in range(2)]
    model.train_draw_fn = lambda i: draws[i]
    opt = torch.optim.AdamW(model.parameters(), lr=6e-4, weight_decay=0.1)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = orc.mpjpe(model(x2d.to(DEV), target), target)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses


def test_ddp_wrapper_produces_the_same_gradients():
    """torch DistributedDataParallel over RCCL (world size 1 on this box) around the train-mode module: the gradient
    hooks fire for every parameter of the custom autograd node and the averaged gradients equal the plain ones."""
    import os
    import torch.distributed as dist
    from __graft_entry__ import make_model
    model, _ = make_model(1, 1, seed=151, is_train=True, depth=1)
    x2d, _ = gu.synthetic_inputs_2d(B=2)
    target = gu.synthetic_target_3d(B=2).to(DEV)
    g = torch.Generator().manual_seed(6)
    draws = [(torch.tensor([300]), torch.randn(27, 134, 3, generator=g)) for _ in range(2)]
    model.train_draw_fn = lambda i: draws[i]
    orc.mpjpe(model(x2d.to(DEV), target), target).backward()
    plain = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
        orc.mpjpe(ddp(x2d.to(DEV), target), target).backward()
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            assert p.grad is not None and torch.equal(p.grad, plain[n]), n
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("train_dtype", ["bf16x3", "f32"])
def test_train_two_rank_rehearsal_on_one_gpu(tmp_path, train_dtype):
    """The one parallel mechanism of the reference's training (main_h3wb.py:699-705,850-871: replicas + gradient averaging) as it
    runs here: torch DistributedDataParallel around pafuse_amd.D3DP, TWO real ranks started the way the driver starts a scaling run
    (`python bench.py --train --gpus 2`: the parent spawns torch.distributed.run before it touches the GPU), sharing this box's one
    GPU over gloo (a rehearsal of the code path: prepare_for_ddp, the autograd node's hooks under a multi-rank process group, the
    bucketed all-reduce).  The gradients rank 0 holds after ONE backward - DDP's average over the two ranks, each on its own
    disjoint batch - must equal a single-process step on the CONCATENATED batch (same weights, same per-sample draws, DropPath
    off): within 5e-4 of every tensor's largest entry.  Both requested precisions: 'bf16x3' (kept under the process group since
    round 6) and 'f32'."""
    import json
    import os
    import subprocess
    import sys
    from __graft_entry__ import make_model
    from tests.conftest import ROOT
    B = 3
    dump = str(tmp_path / "grads.pt")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--train", "--gpus", "2", "--backend", "gloo", "--single-device",
           "--batch", str(B), "--train-dtype", train_dtype, "--no-cpu-baseline", "--dump-grads", dump]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2, res.stdout
    got = torch.load(dump)
    assert got["world"] == 2 and got["B_per_rank"] == B
    # the same step in ONE process on the concatenated batch (rank r's clips: seeds 1234 + r / 1235 + r, draws 7000 + r B + i)
    model, _ = make_model(1, 1, seed=51, is_train=True)
    model.precision = got["precision"]          # what prepare_for_ddp left in effect on the ranks
    assert got["precision"] == train_dtype, got["precision"]
    for m in model.denoisers().values():
        m.drop_path_rate = 0.0

    def draw(i):
        g = torch.Generator().manual_seed(7000 + i)
        return torch.randint(0, 1000, (1,), generator=g), torch.randn(27, 134, 3, generator=g)
    model.train_draw_fn = draw
    x2d = torch.cat([gu.synthetic_inputs_2d(B=B, seed=1234 + r)[0] for r in range(2)]).to(DEV)
    target = torch.cat([gu.synthetic_target_3d(B=B, seed=1235 + r) for r in range(2)]).to(DEV)
    pred = model(x2d, target)
    torch.mean(torch.norm(pred - target, dim=-1)).backward()
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    assert set(names) == set(got["grads"]) and len(names) == 624, (len(names), len(got["grads"]))
    worst = 0.0
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        a, b = got["grads"][n].double(), p.grad.detach().cpu().double()
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        worst = max(worst, err / max(scale, 1e-30))
        assert err <= 5e-4 * scale + 1e-12, (n, err, scale)
    print(f"two-rank averaged gradients vs one process on the concatenated batch ({train_dtype}): worst max|d| / max|g| = {worst:.2e}")


def test_training_errors_are_loud():
    """unsupported width, CPU tensors and a short activation buffer are refused - nothing falls back"""
    import ctypes as C
    import pafuse_amd
    from pafuse_amd import _lib
    m = pafuse_amd.MixSTE2(3, 5, 5, 16, 2, 8, drop_path_rate=0.0, is_train=True).to(DEV)
    args = (torch.zeros(1, 3, 5, 2, device=DEV), torch.zeros(1, 3, 5, 3, device=DEV),
            torch.zeros(1, dtype=torch.long, device=DEV))
    with pytest.raises(_lib.PafuseError):
        m(*args)
    z = load_golden("g12_train_tiny.npz")
    m = _tiny(z)
    with pytest.raises(_lib.PafuseError):
        m(z["x2d"], z["x3d"], z["t"])                                    # CPU tensors
    lib, w = _lib.load(), m.weights_struct()
    need = lib.pafuse_mixste2_train_bytes(C.byref(w), 4)
    assert need > 0
    short = torch.empty(need // 2, dtype=torch.uint8, device=DEV)
    out = torch.empty(4, 3, 5, 3, device=DEV)
    rc = lib.pafuse_mixste2_train_forward(C.byref(w), z["x2d"].to(DEV).data_ptr(), z["x3d"].to(DEV).data_ptr(),
                                          z["t"].to(DEV).data_ptr(), 4, None, out.data_ptr(), short.data_ptr(),
                                          need // 2, torch.cuda.current_stream().cuda_stream)
    assert rc == -3 and b"too small" in lib.pafuse_last_error()


def test_train_epoch_on_synthetic_h3wb_files_and_checkpoint_round_trip(tmp_path):
    """files -> ChunkedClips -> train_epoch (HIP forward/backward, AdamW) -> reference-format checkpoint -> an eval-mode
    D3DP loads it (same 636 keys) and samples."""
    import os
    import pafuse_amd
    from pafuse_amd import config, h3wb, harness
    from tests.conftest import ROOT
    ds = h3wb.Human3WBDataset(os.path.join(ROOT, "tests", "golden", "h3wb_synth", "train_h3wb.npz"))
    keypoints = h3wb.prepare_keypoints(ds)
    kl, kr = (list(x) for x in ds.keypoints_metadata["keypoints_symmetry"])
    jl, jr = list(ds.skeleton().joints_left()), list(ds.skeleton().joints_right())
    cams, p3, p2 = h3wb.fetch(["S1", "S5"], keypoints, ds)
    gen = h3wb.ChunkedClips(4, cams, p3, p2, 27, augment=True, kps_left=kl, kps_right=kr, joints_left=jl, joints_right=jr)
    args = config.load(overrides=["model.dep=1"])
    torch.manual_seed(0)
    model = pafuse_amd.D3DP(args, jl, jr, dataset=ds, is_train=True).to(DEV).train()
    opt = torch.optim.AdamW(model.parameters(), lr=6e-4, weight_decay=0.1)
    losses = [h3wb.train_epoch(model, opt, gen, ds, torch.device(DEV)) for _ in range(3)]
    assert all(l == l and l > 0 for l in losses) and losses[-1] < losses[0], losses
    fname = h3wb.save_state(model, opt, 3, 6e-4, str(tmp_path), random_state=gen.random_state())
    ckpt = harness.read_checkpoint(fname)
    assert all(k.startswith("module.") for k in ckpt["model_pos"])                     # the reference's layout
    assert set(ckpt) == {"optimizer", "epoch", "lr", "model_pos", "random_state"} and len(ckpt["model_pos"]) == 12 + 3 * 40
    ev = pafuse_amd.D3DP(args, jl, jr, dataset=ds, is_train=False, num_proposals=2, sampling_timesteps=1)
    harness.load_checkpoint(ev, ckpt)
    ev = ev.to(DEV).eval()
    x2d = harness.cut_clips(torch.as_tensor(p2[0], dtype=torch.float32)).to(DEV)      # 8 frames -> one padded clip
    with pytest.raises(ValueError):                                                    # wrong frame count: refused
        ev(x2d[:, :8], None, input_2d_flip=x2d[:, :8])
    out = ev(x2d, None, input_2d_flip=harness.flip_2d(x2d, kl, kr))
    assert out.shape == (1, 1, 2, 27, 134, 3) and bool(torch.isfinite(out).all())


@pytest.mark.parametrize("F,J,C,depth,B", [(2, 1, 64, 1, 1), (5, 7, 128, 1, 2), (9, 17, 64, 2, 3), (27, 80, 64, 1, 1)])
def test_train_gradients_vs_oracle_odd_shapes(F, J, C, depth, B):
    """edge geometries (one joint, sequences of 80, widths 64/128, several batch sizes) against the oracle's autograd"""
    import pafuse_amd
    m = pafuse_amd.MixSTE2(num_frame=F, num_joints=J, in_chans=5, embed_dim_ratio=C, depth=depth, num_heads=8,
                           drop_path_rate=0.0, is_train=True)
    sd = {k: gu.seeded_tensor(k, v.shape, 91) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    g = torch.Generator().manual_seed(92)
    x2d = torch.rand(B, F, J, 2, generator=g) * 2 - 1
    x3d = torch.randn(B, F, J, 3, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    dout = torch.randn(B, F, J, 3, generator=g)
    out = m(x2d.to(DEV), x3d.to(DEV), t.to(DEV))
    out.backward(dout.to(DEV))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = orc.mixste2_train(leaves, "", x2d, x3d, t, depth=depth, heads=8)
    assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=0, atol=1e-5), (out.cpu() - ref).abs().max()
    ref.backward(dout)
    for n, p in m.named_parameters():
        _close(p.grad, leaves[n].grad, n)


# ---------------------------------------------------------------------------------------------------------------------------
# The backward kernels on their own (unit entries pafuse_attention_backward / pafuse_linear_weight_grad): against fp64
# ---------------------------------------------------------------------------------------------------------------------------
def _attention_fp64(qkv, heads, rows_of):
    """softmax(q k^T d^-1/2) v in fp64 with torch autograd; rows_of: list of index tensors, the rows of each sequence."""
    M, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    o = torch.zeros(M, C, dtype=torch.float64)
    for rows in rows_of:
        x = qkv[rows].reshape(len(rows), 3, heads, d)
        q, k, v = x[:, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)      # [heads, L, d]
        p = torch.softmax(q @ k.transpose(1, 2) * d ** -0.5, dim=-1)
        o = o.index_put((rows,), (p @ v).transpose(0, 1).reshape(len(rows), C))
    return o


# (L, d): the four instantiations the step launches (spatial 24 / 68 / 42 joints, temporal 27 frames at d = 48 / 28 / 32), the
# padded corners (L = 80 = a full tile, 17, 5; d = 8, 16) and one shape of the FMA fallback (d = 12 is not a multiple of 16... it
# is a multiple of 4: the matrix kernel; L = 81 exceeds its tiles)
@pytest.mark.parametrize("L,d,temporal", [(24, 48, False), (27, 48, True), (68, 28, False), (27, 28, True), (42, 32, False),
                                          (27, 32, True), (80, 8, False), (17, 16, True), (5, 4, False), (33, 12, False),
                                          (81, 8, False)])
def test_attention_backward_kernel_against_fp64(L, d, temporal):
    from pafuse_amd import ops
    heads, other = 8, 3                     # `other` = the axis the sequences do not run over (frames or joints), 2 clips
    C = heads * d
    g = torch.Generator().manual_seed(1000 * L + d)
    nseq, M = 2 * other, 2 * other * L
    qkv = torch.randn(M, 3 * C, generator=g) * 1.5
    d_o = torch.randn(M, C, generator=g)
    if temporal:     # rows (b, f, j) with f the sequence axis: sequence s = (b, j) holds rows b L other + t other + j
        rows_of = [torch.arange(L) * other + (s // other) * L * other + s % other for s in range(nseq)]
        kw = dict(group=other, group_stride=L * other, seq_stride=1, tok_stride=other)
    else:
        rows_of = [torch.arange(L) + s * L for s in range(nseq)]
        kw = {}
    q64 = qkv.double().requires_grad_(True)
    _attention_fp64(q64, heads, rows_of).backward(d_o.double())
    got = ops.attention_backward(qkv.to(DEV), d_o.to(DEV), heads, nseq, L, **kw).cpu().double()
    err = (got - q64.grad).abs()
    scale = float(q64.grad.abs().max())
    print(f"attention backward L={L} d={d}: max {float(err.max()):.2e} mean {float(err.mean()):.2e} of scale {scale:.2e}")
    assert float(err.max()) <= 4e-6 * scale + 1e-7 and float(err.mean()) <= 3e-7 * scale, (float(err.max()), float(err.mean()), scale)


# (N, K): one shape per tile form of the split kernel - 256 x 256 (hands), 384 x 128 (body), 224 x 224 (face), 192 x 192 (a width
# that only that tile divides), the 128 x 128 fallback with ragged edges - at a contraction that is not a multiple of anything
@pytest.mark.parametrize("N,K,M", [(768, 256, 4696), (1152, 384, 4696), (672, 224, 4696), (192, 576, 4696), (200, 72, 4696),
                                   (768, 256, 389)])          # (below 4096 rows: the 128 x 128 kernel whatever the width)
@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_weight_gradient_kernels_exact_on_integers_and_against_fp64(N, K, M, precision):
    from pafuse_amd import ops
    g = torch.Generator().manual_seed(N + K + M)
    # small integers: every product and every partial sum is exact in fp32, whatever the order - any slip of a fragment
    # layout, a tile edge or a split boundary shows as a wrong integer
    dy = torch.randint(-3, 4, (M, N), generator=g).float()
    x = torch.randint(-3, 4, (M, K), generator=g).float()
    dw, db = ops.linear_weight_grad(dy.to(DEV), x.to(DEV), precision)
    assert torch.equal(dw.cpu().double(), dy.double().t() @ x.double())
    assert torch.equal(db.cpu().double(), dy.double().sum(0))
    # 24-bit integers against powers of two: all three slices of an operand are needed
    xi = torch.randint(-2 ** 23, 2 ** 23, (M, K), generator=g).float()
    dp = torch.zeros(M, N)
    dp[torch.arange(M), torch.randint(0, N, (M,), generator=g)] = 1.0      # one 1 per row: every sum has few terms... of one column
    dp = dp * (2.0 ** torch.randint(-3, 4, (M, 1), generator=g).float())
    if precision == "bf16x3":
        dw2, _ = ops.linear_weight_grad(dp.to(DEV), xi.to(DEV), precision, bias=False)
        want = dp.double().t() @ xi.double()
        assert float((dw2.cpu().double() - want).abs().max()) <= 2.0 ** -20 * float(want.abs().max())
    # random data against fp64
    dy, x = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g) * 2 + 0.5
    dw, db = ops.linear_weight_grad(dy.to(DEV), x.to(DEV), precision)
    want = dy.double().t() @ x.double()
    err = (dw.cpu().double() - want).abs()
    assert float(err.max()) <= 3e-6 * float(want.abs().max()) and float(err.mean()) <= 3e-7 * float(want.abs().max())
    _close(db, dy.double().sum(0).float(), "db", rel=3e-6)
