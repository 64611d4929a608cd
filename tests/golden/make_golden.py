#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (valeoai/PAFUSE, /root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  Nothing of the
reference is copied: it is imported through three shims

  1. stub ``timm`` modules (only DropPath is used, and only in training),
  2. a SimpleNamespace config tree instead of hydra/omegaconf and a tiny dataset object exposing the
     part-index tables of common/h3wb_dataset.py:49-61,198-213,
  3. ``model.device = 'cpu'`` plus a no-op ``Tensor.cuda`` (common/diffusionpose.py:71,288).

Usage:  python tests/golden/make_golden.py            (re-writes every fixture, deterministic)

Real-width fixtures do not store the 35 M weights: they are re-generated from a seed by
``golden_util.seeded_state_dict`` and a SHA-256 of the result is stored, so the GPU box can prove it
re-generated identical tensors.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("PAFUSE_REFERENCE", "/root/reference")

from tests.golden import golden_util as gu  # noqa: E402


DROP_TAPE = []      # DropPath factors in the order the reference's blocks drew them


# ------------------------------------------------------------------------------------------------- shims
def install_shims():
    class DropPath(torch.nn.Module):
        """timm.models.layers.DropPath is absent from this image (and unpinned by the reference): its published
        algorithm (timm.layers.drop.drop_path, scale_by_keep=True).  The factors it draws are recorded in DROP_TAPE."""

        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            if self.p == 0.0 or not self.training:
                return x
            keep = 1 - self.p
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0:
                mask.div_(keep)
            DROP_TAPE.append(mask.reshape(-1).clone())
            return x * mask

    def _mk(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    _mk("timm")
    _mk("timm.data", IMAGENET_DEFAULT_MEAN=(0, 0, 0), IMAGENET_DEFAULT_STD=(1, 1, 1))
    _mk("timm.models")
    _mk("timm.models.helpers", load_pretrained=lambda *a, **k: None)
    _mk("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: (x, x), trunc_normal_=lambda *a, **k: None)
    _mk("timm.models.registry", register_model=lambda f: f)
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def make_args():
    return SimpleNamespace(
        general=SimpleNamespace(part_based_model=True),
        data=SimpleNamespace(num_kps=134, merge_hands=True),
        model=SimpleNamespace(number_of_frames=27, test_time_augmentation=True, diff_model="MixSTE2",
                              input_size=5, dep=8, cs=288),
        ft2d=SimpleNamespace(timestep=1000, scale=1.0),
    )


class FakeDataset:
    """What D3DP / the pose utilities read from Human3WBDataset (common/diffusionpose.py:73-75)."""

    def __init__(self):
        self.metadata = {}
        self.root_indices = dict(gu.ROOT_INDICES)
        self.parts_joint_indices = {k: list(v) for k, v in gu.DATASET_PART_JOINTS.items()}
        self.parts_connection_indices = dict(gu.CONNECTION_INDICES)


def build_reference_d3dp(P, T, flip=True):
    from common.diffusionpose import D3DP
    args = make_args()
    args.model.test_time_augmentation = flip
    m = D3DP(args, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, FakeDataset(), is_train=False,
             num_proposals=P, sampling_timesteps=T)
    m.device = "cpu"
    m.eval()
    return m


class NoiseTape:
    """Patch torch.randn / randn_like so the reference consumes a prepared list of noise tensors."""

    def __init__(self, noises):
        self.noises = list(noises)
        self.k = 0

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like

        def take(*a, **k):
            out = self.noises[self.k]
            self.k += 1
            return out.clone()

        torch.randn = take
        torch.randn_like = take
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# -------------------------------------------------------------------------------------------------- G1
def g1_tiny_mixste():
    """tiny MixSTE2 (F=3, J=5, C=16, depth 2): full state dict + inputs + output."""
    from common.mixste import MixSTE2
    torch.manual_seed(11)
    m = MixSTE2(num_frame=3, num_joints=5, in_chans=5, embed_dim_ratio=16, depth=2, num_heads=8,
                mlp_ratio=2.0, qkv_bias=True, qk_scale=None, drop_path_rate=0.0, is_train=False).eval()
    with torch.no_grad():
        m.Spatial_pos_embed.normal_(0, 0.2)
        m.Temporal_pos_embed.normal_(0, 0.2)
        for k, p in m.named_parameters():        # make LayerNorm affine parameters non-trivial
            if "norm" in k or k.startswith("head.0"):
                p.add_(torch.randn_like(p) * 0.1)
    g = torch.Generator().manual_seed(12)
    x2d = torch.rand(2, 3, 5, 2, generator=g) * 2 - 1
    x3d = torch.randn(2, 4, 3, 5, 3, generator=g)
    t = torch.tensor([999, 3])
    with torch.no_grad():
        out = m(x2d, x3d, t)
    arrays = {"sd." + k: v for k, v in m.state_dict().items()}
    save("g1_tiny_mixste.npz", x2d=x2d, x3d=x3d, t=t, out=out, **arrays)


# -------------------------------------------------------------------------------------------------- G2
def g2_schedule():
    m = build_reference_d3dp(1, 1)
    arrays = {"buf." + k: v for k, v in m.state_dict().items() if not k.startswith("pose_estimator")}
    for T in (1, 2, 5, 10, 20, 50):
        times = torch.linspace(-1, 999, steps=T + 1)
        times = list(reversed(times.int().tolist()))
        pairs = list(zip(times[:-1], times[1:]))
        arrays[f"pairs.{T}"] = np.asarray(pairs, dtype=np.int64)
        coefs = []
        for time, time_next in pairs:
            if time_next < 0:
                continue
            alpha = m.alphas_cumprod[time]
            alpha_next = m.alphas_cumprod[time_next]
            sigma = 1.0 * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = (1 - alpha_next - sigma ** 2).sqrt()
            coefs.append([alpha_next.sqrt().item(), c.item(), sigma.item()])
        arrays[f"coefs.{T}"] = np.asarray(coefs, dtype=np.float64).reshape(-1, 3)
    save("g2_schedule.npz", **arrays)


# -------------------------------------------------------------------------------------------------- G3
def g3_time_mlp():
    """sinusoid + time_mlp at the three real widths; weights re-generated by key (SHA-256 stored)."""
    from common.mixste import MixSTE2
    arrays = {}
    ts = torch.tensor([999, 899, 799, 699, 599, 499, 399, 299, 199, 99, 0])
    for part, C in gu.PART_WIDTH.items():
        m = MixSTE2(num_frame=2, num_joints=2, in_chans=5, embed_dim_ratio=C, depth=1, num_heads=8,
                    drop_path_rate=0.0, is_train=False).eval()
        sd = gu.seeded_like(m.time_mlp.state_dict(), seed=31, prefix=f"g3.{part}.time_mlp.")
        m.time_mlp.load_state_dict(sd)
        arrays[f"{part}.sha"] = np.frombuffer(gu.sha256_of(sd), dtype=np.uint8)
        with torch.no_grad():
            arrays[f"{part}.sin"] = m.time_mlp[0](ts)
            arrays[f"{part}.out"] = m.time_mlp(ts)
    save("g3_time_mlp.npz", t=ts, **arrays)


# -------------------------------------------------------------------------------------------------- G4
def g4_blocks():
    """one real-width Block per part on a few spatial and temporal sequences; weights re-generated by key."""
    from common.mixste import Block
    from functools import partial
    arrays = {}
    for part, C in gu.PART_WIDTH.items():
        J = len(gu.PART_JOINTS[part])
        blk = Block(dim=C, num_heads=8, mlp_ratio=2.0, qkv_bias=True, qk_scale=None,
                    norm_layer=partial(torch.nn.LayerNorm, eps=1e-6)).eval()
        sd = gu.seeded_like(blk.state_dict(), seed=41, prefix=f"g4.{part}.")
        blk.load_state_dict(sd)
        arrays[f"{part}.sha"] = np.frombuffer(gu.sha256_of(sd), dtype=np.uint8)
        g = torch.Generator().manual_seed(42)
        xs = torch.randn(3, J, C, generator=g)
        xt = torch.randn(4, 27, C, generator=g)
        with torch.no_grad():
            arrays[f"{part}.xs"], arrays[f"{part}.ys"] = xs, blk(xs)
            arrays[f"{part}.xt"], arrays[f"{part}.yt"] = xt, blk(xt)
    save("g4_blocks.npz", **arrays)


# -------------------------------------------------------------------------------------------------- G5
def g5_d3dp_loops():
    """full D3DP at real dims: flip P=2,T=2 (every x_start) and no-flip P=1,T=1; one per-part MixSTE2 pass."""
    arrays = {}
    m = build_reference_d3dp(2, 2, flip=True)
    sd = gu.seeded_state_dict(m.state_dict(), seed=51)
    m.load_state_dict(sd)
    arrays["sha"] = np.frombuffer(gu.sha256_of(sd), dtype=np.uint8)
    x2d, x2d_flip = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=2, n=2, seed=1)
    with NoiseTape(noises) as tape, torch.no_grad():
        out = m(x2d, None, input_2d_flip=x2d_flip)
        assert tape.k == 2
    arrays.update(flip_x2d=x2d, flip_x2d_flip=x2d_flip, flip_out=out)

    # a single denoiser pass per part at a mid-schedule timestep (x_3d of plausible scale)
    g = torch.Generator().manual_seed(52)
    x3d = torch.randn(1, 2, 27, 134, 3, generator=g).clamp(-1.1, 1.1)
    t = torch.tensor([499])
    with torch.no_grad():
        for part, idx in m.parts_joint_indices.items():
            arrays[f"part.{part}"] = m.pose_estimator[part](x2d[..., idx, :], x3d[..., idx, :], t)
    arrays["part_x3d"] = x3d

    m1 = build_reference_d3dp(1, 1, flip=False)
    m1.load_state_dict(sd)
    noises1 = gu.synthetic_noises(B=1, P=1, n=1, seed=2)
    with NoiseTape(noises1), torch.no_grad():
        arrays["noflip_out"] = m1(x2d, None)
    m1f = build_reference_d3dp(1, 1, flip=True)
    m1f.load_state_dict(sd)
    with NoiseTape(noises1), torch.no_grad():
        arrays["flip11_out"] = m1f(x2d, None, input_2d_flip=x2d_flip)
    save("g5_d3dp.npz", **arrays)


# -------------------------------------------------------------------------------------------------- G6
def g6_index_ops():
    """integer-valued tensors through the flip permutation, part split/concat and the pose utilities."""
    from common.utils import center_pose_parts, wb_pose_from_parts
    ds = FakeDataset()
    m = build_reference_d3dp(1, 1)
    g = torch.Generator().manual_seed(61)
    x = torch.randint(-50, 50, (2, 3, 4, 134, 3), generator=g).float()
    lr = gu.SYN_JOINTS_LEFT + gu.SYN_JOINTS_RIGHT
    rl = gu.SYN_JOINTS_RIGHT + gu.SYN_JOINTS_LEFT
    flipped = x.clone()
    flipped[:, :, :, :, 0] *= -1
    flipped[:, :, :, lr] = flipped[:, :, :, rl]
    d2, d3 = m.split_data(x[:, 0, :, :, :2], x)
    cat = torch.cat([d3[p] for p in m.pose_estimator.keys()], dim=-2)
    pose = torch.randint(-50, 50, (2, 4, 134, 3), generator=g).float()
    centred = center_pose_parts(pose.clone(), ds)
    wb_in = pose.clone()
    wb_out = wb_pose_from_parts(wb_in, ds)
    save("g6_index_ops.npz", x=x, flipped=flipped, cat=cat,
         split_body=d3["body"], split_face=d3["face"], split_hands=d3["hands"],
         pose=pose, centred=centred, wb_out=wb_out, wb_in_after=wb_in)


# -------------------------------------------------------------------------------------------------- G7
def g7_metrics():
    from common.loss import mpjpe_diffusion, mpjpe_diffusion_all_min, mpjpe_diffusion_reproj
    from common.camera import project_to_2d
    g = torch.Generator().manual_seed(71)
    B, T, P = 2, 3, 4
    pred = torch.randn(B, T, P, 5, 134, 3, generator=g) * 0.3
    target = torch.randn(B, 5, 134, 3, generator=g) * 0.3
    target_2d = torch.rand(B, 5, 134, 2, generator=g) * 2 - 1
    traj = torch.randn(B, 5, 1, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 4.0])
    cam = torch.tensor([[2.29, 2.287, 0.025, 0.029, -0.207, 0.247, -0.003, -0.0009, -0.001]])
    absolute = (pred + traj[:, None, None]).reshape(B * T * P * 5, 134, 3)
    reproj = project_to_2d(absolute, cam.repeat(B * T * P * 5, 1)).reshape(B, T, P, 5, 134, 2)
    save("g7_metrics.npz", pred=pred, target=target, target_2d=target_2d, traj=traj, cam=cam, reproj=reproj,
         j_best=mpjpe_diffusion_all_min(pred, target),
         p_best=mpjpe_diffusion(pred.clone(), target.clone())[0],
         p_agg=mpjpe_diffusion_all_min(pred, target, mean_pos=True),
         j_agg=mpjpe_diffusion_reproj(pred, target, reproj, target_2d))


# ------------------------------------------------------------------------------------------------- G11
def g11_scale():
    """flip loop with ft2d.scale = 2.0 (clamp bounds, /scale, *scale paths), P=2, T=2."""
    from common.diffusionpose import D3DP
    args = make_args()
    args.ft2d.scale = 2.0
    m = D3DP(args, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, FakeDataset(), is_train=False, num_proposals=2,
             sampling_timesteps=2)
    m.device = "cpu"
    m.eval()
    sd = gu.seeded_state_dict(m.state_dict(), seed=111)
    m.load_state_dict(sd)
    x2d, x2d_flip = gu.synthetic_inputs_2d(B=1)
    noises = [n * 1.5 for n in gu.synthetic_noises(B=1, P=2, n=2, seed=12)]
    with NoiseTape(noises), torch.no_grad():
        out = m(x2d, None, input_2d_flip=x2d_flip)
    save("g11_scale.npz", sha=np.frombuffer(gu.sha256_of(sd), dtype=np.uint8), out=out)


# ------------------------------------------------------------------------------------------------- G17
def g17_single_model():
    """general.part_based_model = False: ONE MixSTE2 over the 134 keypoints at width model.cs = 288
    (common/diffusionpose.py:150-153), flip loop P=2, T=2 and the no-flip loop P=1, T=1."""
    from common.diffusionpose import D3DP

    def build(P, T, flip):
        args = make_args()
        args.general.part_based_model = False
        args.model.test_time_augmentation = flip
        m = D3DP(args, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, FakeDataset(), is_train=False, num_proposals=P,
                 sampling_timesteps=T)
        m.device = "cpu"
        return m.eval()

    m = build(2, 2, True)
    sd = gu.seeded_state_dict(m.state_dict(), seed=171)
    m.load_state_dict(sd)
    x2d, x2d_flip = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=2, n=2, seed=17)
    with NoiseTape(noises) as tape, torch.no_grad():
        out = m(x2d, None, input_2d_flip=x2d_flip)
        assert tape.k == 2
    m1 = build(1, 1, False)
    m1.load_state_dict(sd)
    noises1 = gu.synthetic_noises(B=1, P=1, n=1, seed=18)
    with NoiseTape(noises1), torch.no_grad():
        out1 = m1(x2d, None)
    save("g17_single_model.npz", sha=np.frombuffer(gu.sha256_of(sd), dtype=np.uint8), flip_out=out, noflip_out=out1,
         n_keys=np.asarray(len(sd)))


# ------------------------------------------------------------------------------------------------- G18
G18_KW = dict(num_frame=9, num_joints=17, in_chans=5, embed_dim_ratio=64, depth=2, num_heads=8, mlp_ratio=3.,
              qkv_bias=False, qk_scale=0.3, drop_rate=0.1, attn_drop_rate=0.2, is_train=False)


def g18_mixste_options():
    """MixSTE2 constructor options PAFUSE never sets (common/mixste.py:141-144): mlp_ratio=3, qkv_bias=False,
    qk_scale=0.3 and dropout rates (identity in eval) on a small eval-mode model."""
    from common.mixste import MixSTE2
    m = MixSTE2(**G18_KW).eval()
    sd = gu.seeded_state_dict(m.state_dict(), seed=181)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(182)
    x2d = torch.rand(2, 9, 17, 2, generator=g) * 2 - 1
    x3d = torch.randn(2, 3, 9, 17, 3, generator=g).clamp(-1.1, 1.1)
    t = torch.tensor([17, 803])
    with torch.no_grad():
        out = m(x2d, x3d, t)
    save("g18_mixste_options.npz", sha=np.frombuffer(gu.sha256_of(sd), dtype=np.uint8), x2d=x2d, x3d=x3d, t=t, out=out,
         n_keys=np.asarray(len(sd)))


# ------------------------------------------------------------------------------------------------- G12
def _grad_stats(g):
    """compact pin of one gradient tensor: sum, L2 norm, first 8 entries"""
    flat = g.reshape(-1).double()
    head = torch.zeros(8, dtype=torch.float64)
    head[:min(8, flat.numel())] = flat[:8]
    return torch.cat([flat.sum()[None], flat.norm()[None], head])


def g12_train_tiny():
    """train-mode MixSTE2 (F=3, J=5, C=64, depth 2, DropPath 0.5): forward, DropPath factors as drawn, and the
    gradient of every parameter for a seeded output gradient."""
    from common.mixste import MixSTE2
    torch.manual_seed(121)
    m = MixSTE2(num_frame=3, num_joints=5, in_chans=5, embed_dim_ratio=64, depth=2, num_heads=8,
                mlp_ratio=2.0, qkv_bias=True, qk_scale=None, drop_path_rate=0.5, is_train=True).train()
    with torch.no_grad():
        m.Spatial_pos_embed.normal_(0, 0.2)
        m.Temporal_pos_embed.normal_(0, 0.2)
        for k, p in m.named_parameters():
            if "norm" in k or k.startswith("head.0"):
                p.add_(torch.randn_like(p) * 0.1)
    g = torch.Generator().manual_seed(122)
    x2d = torch.rand(4, 3, 5, 2, generator=g) * 2 - 1
    x3d = torch.randn(4, 3, 5, 3, generator=g)
    t = torch.tensor([999, 3, 500, 41])
    dout = torch.randn(4, 3, 5, 3, generator=g)
    DROP_TAPE.clear()
    torch.manual_seed(123)
    out = m(x2d, x3d, t)
    out.backward(dout)
    arrays = {"sd." + k: v for k, v in m.state_dict().items()}
    arrays.update({"grad." + k: p.grad for k, p in m.named_parameters()})
    arrays.update({f"drop.{i}": d for i, d in enumerate(DROP_TAPE)})
    save("g12_train_tiny.npz", x2d=x2d, x3d=x3d, t=t, dout=dout, out=out.detach(), n_drop=torch.tensor(len(DROP_TAPE)),
         **arrays)


# ------------------------------------------------------------------------------------------------- G13
def g13_d3dp_train():
    """D3DP.forward in train mode at the real widths (depth 1 to keep the fixture small), B=2: the per-sample
    (t, noise) draws, the noised poses, the prediction, the mpjpe loss and compact statistics of every gradient."""
    from common.diffusionpose import D3DP
    from common.loss import mpjpe
    args = make_args()
    args.model.dep = 1
    m = D3DP(args, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, FakeDataset(), is_train=True)
    m.device = "cpu"
    m.train()
    sd = gu.seeded_state_dict(m.state_dict(), seed=131)
    m.load_state_dict(sd)
    x2d, _ = gu.synthetic_inputs_2d(B=2)
    target = gu.synthetic_target_3d(B=2)
    g = torch.Generator().manual_seed(132)
    ts = [torch.tensor([977]), torch.tensor([12])]
    noises = [torch.randn(27, 134, 3, generator=g) for _ in range(2)]
    real_randint, real_randn = torch.randint, torch.randn
    it_t, it_n = iter(ts), iter(noises)
    torch.randint = lambda *a, **k: next(it_t).clone()          # prepare_diffusion_concat asks for device='cuda'
    torch.randn = lambda *a, **k: next(it_n).clone()
    try:
        x_poses, _, t = m.prepare_targets(target)
    finally:
        torch.randint, torch.randn = real_randint, real_randn
    it_t, it_n = iter(ts), iter(noises)
    torch.randint = lambda *a, **k: next(it_t).clone()
    torch.randn = lambda *a, **k: next(it_n).clone()
    DROP_TAPE.clear()
    try:
        pred = m(x2d, target)
    finally:
        torch.randint, torch.randn = real_randint, real_randn
    loss = mpjpe(pred, target)
    loss.backward()
    stats = {"gstat." + k: _grad_stats(p.grad) for k, p in m.named_parameters()}
    save("g13_d3dp_train.npz", sha=np.frombuffer(gu.sha256_of(sd), dtype=np.uint8), t=torch.stack(ts),
         noise=torch.stack(noises), x_poses=x_poses.float(), pred=pred.detach(), loss=loss.detach(),
         n_drop=torch.tensor(len(DROP_TAPE)), **stats)


# -------------------------------------------------------------------------------------------------- G9
def g9_evaluate_accumulators():
    """the 14 per-step error vectors evaluate() accumulates (main_h3wb.py:327-362) on random part-centred inputs."""
    from common.utils import center_pose_parts, wb_pose_from_parts
    from common.loss import mpjpe_diffusion, mpjpe_diffusion_all_min, mpjpe_diffusion_reproj
    from common.camera import project_to_2d
    ds = FakeDataset()
    ds.parts_connection_indices = dict(gu.CONNECTION_INDICES)
    g = torch.Generator().manual_seed(91)
    B, T, P, Fr = 2, 3, 5, 27
    pred_parts = center_pose_parts(torch.randn(B, T, P, Fr, 134, 3, generator=g) * 0.3, ds)
    gt_parts = center_pose_parts(torch.randn(B, Fr, 134, 3, generator=g) * 0.3, ds)
    x2d = torch.rand(B, Fr, 134, 2, generator=g) * 2 - 1
    traj = torch.randn(B, Fr, 1, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 4.0])
    cam = torch.tensor([[2.29, 2.287, 0.025, 0.029, -0.207, 0.247, -0.003, -0.0009, -0.001]])
    pred = wb_pose_from_parts(pred_parts.clone(), ds)
    gt = wb_pose_from_parts(gt_parts.clone(), ds)
    absolute = (pred + traj.unsqueeze(1).unsqueeze(1)).reshape(B * T * P * Fr, 134, 3)
    reproj = project_to_2d(absolute, cam.repeat(B * T * P * Fr, 1)).reshape(B, T, P, Fr, 134, 2)
    out = {"j_best": mpjpe_diffusion_all_min(pred, gt), "p_best": mpjpe_diffusion(pred, gt)[0],
           "p_agg": mpjpe_diffusion_all_min(pred, gt, mean_pos=True),
           "j_agg": mpjpe_diffusion_reproj(pred, gt, reproj, x2d)}
    e, parts = mpjpe_diffusion(pred.clone(), gt.clone(), part_based=True, dataset=ds)
    out["p_best_pb"] = e
    for k, v in parts.items():
        out["p_best_pb_" + k] = v
    e, parts = mpjpe_diffusion_all_min(pred.clone(), gt.clone(), mean_pos=True, part_based=True, dataset=ds)
    out["p_agg_pb"] = e
    for k, v in parts.items():
        out["p_agg_pb_" + k] = v
    assert len(out) == 14
    save("g9_evaluate.npz", pred_parts=pred_parts, gt_parts=gt_parts, x2d=x2d, traj=traj, cam=cam, **out)


# ------------------------------------------------------------------------------------------------- G10
def g10_clip_cutting():
    """eval_data_prepare (main_h3wb.py:122-154) on sequences shorter than / equal to / not a multiple of a clip.
    main_h3wb.py is a script with heavy imports, so only that function's source is exec'd from it."""
    import ast
    src = open(os.path.join(REF, "main_h3wb.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "eval_data_prepare")
    from einops import rearrange
    ns = {"torch": torch, "rearrange": rearrange}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "eval_data_prepare", "exec"), ns)
    g = torch.Generator().manual_seed(101)
    arrays = {}
    for n in (10, 27, 54, 60):
        x2 = torch.randn(1, n, 6, 2, generator=g)
        x3 = torch.randn(1, n, 6, 3, generator=g)
        c2, c3 = ns["eval_data_prepare"](27, x2, x3)
        arrays.update({f"x2.{n}": x2, f"x3.{n}": x3, f"c2.{n}": c2, f"c3.{n}": c3})
    save("g10_clips.npz", **arrays)


# ------------------------------------------------------------------------------------------------- G14
def write_synthetic_h3wb(dirname, seed=141):
    """A tiny dataset with the H3WB npz schema the reference loader expects (common/h3wb_dataset.py:18-26,124-133):
    2 training subjects + the S8 test file, one or two actions, 4 cameras, 133 keypoints, a few frames."""
    rng = np.random.default_rng(seed)
    cams = ["54138969", "55011271", "58860488", "60457274"]
    meta = {"body": list(range(0, 17)), "left_foot": [17, 18, 19], "right_foot": [20, 21, 22],
            "face": list(range(23, 91)), "left_hand": list(range(91, 112)), "right_hand": list(range(112, 133)),
            # 0-based sides; keypoint 0 listed on both sides exercises the duplicate filter (h3wb_dataset.py:30-38)
            "left_side": [0] + [j for j in (list(range(1, 16, 2)) + [17, 18, 19] + list(range(91, 112)))],
            "right_side": [0] + [j for j in (list(range(2, 17, 2)) + [20, 21, 22] + list(range(112, 133)))]}

    def action(n):
        rec = {"global_3d": rng.normal(0, 500, (n, 133, 3)).astype(np.float32), "frame_id": np.arange(n)}
        for c in cams:
            rec[c] = {"sample_id": np.arange(n),
                      "camera_3d": (rng.normal(0, 400, (n, 133, 3)) + [0, 0, 4000]).astype(np.float32),
                      "pose_2d": rng.uniform(0, 1000, (n, 133, 2)).astype(np.float32)}
        return rec

    train = {"S1": {"Directions": action(8), "Walking 1": action(5)}, "S5": {"Directions": action(7)}}
    test = {"S8": {"Directions": action(6)}}
    for s in ("S1", "S5", "S8"):
        meta[s] = {c: {"id": c, "subject": s} for c in cams}
    os.makedirs(dirname, exist_ok=True)
    np.savez_compressed(os.path.join(dirname, "train_h3wb.npz"), metadata=np.array(meta, dtype=object),
                        train_data=np.array(train, dtype=object))
    np.savez_compressed(os.path.join(dirname, "task1_test_3d.npz"), data=np.array(test, dtype=object))


def g14_h3wb_loader():
    """The reference's Human3WBDataset + the data preparation and fetch() of main_h3wb.py:57-119,621-648 on the synthetic
    H3WB files of tests/golden/h3wb_synth/ (written here too)."""
    import ast
    from common.h3wb_dataset import Human3WBDataset
    from common.camera import normalize_screen_coordinates
    d = os.path.join(HERE, "h3wb_synth")
    write_synthetic_h3wb(d)
    ds = Human3WBDataset(os.path.join(d, "train_h3wb.npz"))
    for subject in ds.subjects():                                   # main_h3wb.py:621-648
        for act in ds[subject].keys():
            anim = ds[subject][act]
            anim["positions_3d"] = [p / 1000. for p in anim["positions_3d"]]
    keypoints = {}
    for subject in ds.subjects():
        keypoints[subject] = {}
        for act in ds[subject].keys():
            keypoints[subject][act] = []
            for cam_idx, kps in enumerate(ds[subject][act]["pose_2d"]):
                cam = ds.cameras()[subject][cam_idx]
                kps[..., :2] = normalize_screen_coordinates(kps[..., :2], w=cam["res_w"], h=cam["res_h"])
                keypoints[subject][act].append(kps)
    src = open(os.path.join(REF, "main_h3wb.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "fetch")
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "fetch", "exec"), ns)
    arrays = {"joints_left": torch.tensor(ds.skeleton().joints_left()),
              "joints_right": torch.tensor(ds.skeleton().joints_right()),
              "kps_left": torch.tensor(ds.keypoints_metadata["keypoints_symmetry"][0]),
              "parents": torch.tensor(np.asarray(ds.skeleton().parents())),
              "num_kps": torch.tensor(ds.num_kps)}
    for part, idx in ds.parts_joint_indices.items():
        arrays["part." + part] = torch.tensor(idx)
    for subject in ("S1", "S8"):
        for i, cam in enumerate(ds.cameras()[subject]):
            arrays[f"cam.{subject}.{i}.intrinsic"] = torch.from_numpy(cam["intrinsic"])
            arrays[f"cam.{subject}.{i}.translation"] = torch.from_numpy(cam["translation"])
    arrays["positions.S1.Walking 1"] = torch.from_numpy(ds["S1"]["Walking 1"]["positions"])
    for tag, (subjects, stride, filt) in {"test": (["S8"], 1, None), "train2": (["S1", "S5"], 2, ["Dir"])}.items():
        cams, p3, p2 = ns["fetch"](subjects, keypoints, ds, stride, filt)
        arrays[f"fetch.{tag}.n"] = torch.tensor(len(p2))
        for i in range(len(p2)):
            arrays[f"fetch.{tag}.{i}.cam"] = torch.from_numpy(cams[i])
            arrays[f"fetch.{tag}.{i}.p3"] = torch.from_numpy(p3[i])
            arrays[f"fetch.{tag}.{i}.p2"] = torch.from_numpy(p2[i])
    save("g14_h3wb_loader.npz", **arrays)


# ------------------------------------------------------------------------------------------------- G15
def g15_camera_to_world():
    """camera_to_world (common/camera.py:27-28) with the in-the-wild script's fixed camera rotation."""
    from common.camera import camera_to_world
    g = torch.Generator().manual_seed(151)
    X = torch.randn(2, 3, 7, 134, 3, generator=g).numpy()
    rot = np.array([0.14070565, -0.15007018, -0.7552408, 0.62232804], dtype=np.float32)
    save("g15_camera_to_world.npz", X=X, rot=rot, out=camera_to_world(X, R=rot, t=0))


# ------------------------------------------------------------------------------------------------- G16
def g16_chunked_generator():
    """ChunkedGenerator_Seq (common/generators.py:5-172) on the synthetic H3WB training subjects: shuffled,
    flip-augmented 27-frame clips, batch size 3 - the first two batches of the first two epochs."""
    from common.generators import ChunkedGenerator_Seq
    from pafuse_amd import h3wb                                  # the loader is pinned by G14
    ds = h3wb.Human3WBDataset(os.path.join(HERE, "h3wb_synth", "train_h3wb.npz"))
    keypoints = h3wb.prepare_keypoints(ds)
    kl, kr = ds.keypoints_metadata["keypoints_symmetry"]
    cams, p3, p2 = h3wb.fetch(["S1", "S5"], keypoints, ds)
    gen = ChunkedGenerator_Seq(3, cams, p3, p2, 27, pad=13, causal_shift=0, shuffle=True, augment=True,
                               kps_left=kl, kps_right=kr, joints_left=list(ds.skeleton().joints_left()),
                               joints_right=list(ds.skeleton().joints_right()))
    arrays = {"num_batches": torch.tensor(gen.batch_num()), "num_pairs": torch.tensor(len(gen.pairs))}
    for epoch in range(2):
        for b, (cam, b3, b2) in enumerate(gen.next_epoch()):
            if b < 2:
                arrays[f"e{epoch}.b{b}.cam"] = torch.from_numpy(cam.copy())
                arrays[f"e{epoch}.b{b}.p3"] = torch.from_numpy(b3.copy())
                arrays[f"e{epoch}.b{b}.p2"] = torch.from_numpy(b2.copy())
    save("g16_chunked_generator.npz", **arrays)


# -------------------------------------------------------------------------------------------------- G8
# ------------------------------------------------------------------------------------------------- G19
G19_SUB = (0, 7, 19)     # hypotheses whose whole trajectory is stored


def g19_metric_config_p20_t10():
    """The metric's own configuration run by the REFERENCE: D3DP.forward, flip-TTA, B=1, P=20, T=10 at the real dims
    (BASELINE configs[2]; about three minutes of CPU), on the seeded weights / inputs / noise of the full-size GPU tests
    (tests/test_hip_fullsize.py: weights seed 51, noise seed 160, the first 20 hypotheses of the P=160 draw).  Stored: the
    trajectories of three hypotheses (pointwise check) and, over ALL 20 hypotheses, what the four MPJPE protocols reduce
    to per DDIM step - J-Best / P-Best / P-Agg / J-Agg in mm (fp64 metric arithmetic of the oracle's metric functions,
    themselves pinned by G7 / G9) plus the J-Agg pick, picked 3-D error and 2-D margin of every (step, frame, joint), so
    that a run can be compared with the reference on the joints where both pick the same hypothesis."""
    from oracle import d3dp_oracle as orc
    from tests.test_hip_parity import PROTOCOLS, _j_agg_parts, _mpjpe_report
    m = build_reference_d3dp(20, 10, flip=True)
    sd = gu.seeded_state_dict(m.state_dict(), seed=51)
    m.load_state_dict(sd)
    x2d, x2d_flip = gu.synthetic_inputs_2d(B=1)
    noises = [n[:, :20].contiguous() for n in gu.synthetic_noises(B=1, P=160, n=10, seed=160)]
    with NoiseTape(noises) as tape, torch.no_grad():
        out = m(x2d, None, input_2d_flip=x2d_flip)
        assert tape.k == 10
    assert out.shape == (1, 10, 20, 27, 134, 3)
    target = orc.center_pose_parts(gu.synthetic_target_3d(1))
    rep = _mpjpe_report(out, target, x2d)
    pick, e3, margin = _j_agg_parts(out, target, x2d)
    save("g19_metric_config.npz", sha=np.frombuffer(gu.sha256_of(sd), dtype=np.uint8),
         out_sub=out[:, :, list(G19_SUB)], sub=np.asarray(G19_SUB),
         mpjpe_mm=torch.stack([rep[k] for k in PROTOCOLS]),            # [4, T] fp64
         jagg_pick=pick.to(torch.int8), jagg_e3=e3, jagg_margin=margin)


def g8_default_init():
    """SHA-256 of the reference's default-initialised MixSTE2 under a fixed seed (pins parameter creation order)."""
    from common.mixste import MixSTE2
    torch.manual_seed(123)
    m = MixSTE2(num_frame=27, num_joints=42, in_chans=5, embed_dim_ratio=256, depth=2, num_heads=8,
                drop_path_rate=0.0, is_train=False)
    save("g8_init.npz", sha=np.frombuffer(gu.sha256_of(m.state_dict()), dtype=np.uint8))


if __name__ == "__main__":
    install_shims()
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19"]
    table = dict(g1=g1_tiny_mixste, g2=g2_schedule, g3=g3_time_mlp, g4=g4_blocks, g5=g5_d3dp_loops,
                 g6=g6_index_ops, g7=g7_metrics, g8=g8_default_init, g9=g9_evaluate_accumulators,
                 g10=g10_clip_cutting, g11=g11_scale, g12=g12_train_tiny, g13=g13_d3dp_train,
                 g14=g14_h3wb_loader, g15=g15_camera_to_world,
                 g16=g16_chunked_generator, g17=g17_single_model, g18=g18_mixste_options, g19=g19_metric_config_p20_t10)
    for w in which:
        table[w]()
