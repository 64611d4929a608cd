"""PyTorch-ROCm custom ops over the C ABI: ``torch.ops.pafuse.*`` ("native surface").

Importing this module registers, for the HIP device only (a CPU tensor gets torch's "no kernel for backend" error -
there is no fallback), with fake-tensor shape functions so the ops trace under ``torch.export`` / ``torch.compile``:

  pafuse::linear(x, weight, bias, gelu)                    nn.Linear (+ exact GELU)       common/mixste.py:30-43,54,57
  pafuse::layer_norm(x, weight, bias, eps)                 nn.LayerNorm                   common/mixste.py:96,101
  pafuse::attention(qkv, heads, seq_len, joints)           softmax(q k^T d^-1/2) v        common/mixste.py:65-79
  pafuse::block(x, weights[12], heads, precision)          Block.forward on [S,L,C]       common/mixste.py:113-116
  pafuse::mixste_eval(x2d, x3d, t, weights, depth, heads, precision)  MixSTE2.forward, eval  common/mixste.py:278-298
  pafuse::ddim_loop(..., precision)                        D3DP.ddim_sample[_flip]        common/diffusionpose.py:227-316

``precision`` ('bf16x3' = the modules' default; 'bf16x3_images', 'f16x2', 'f32', 'bf16': pafuse_amd.D3DP.precision) is the
matrix-product mode of the linear layers; in the split modes the ops build and cache the pre-split weight images themselves (cached_split_image).

``weights`` lists are in ``named_parameters()`` order of the corresponding module (= the reference's state-dict
order), so ``list(model.parameters())`` is the argument.  The modules in pafuse_amd call the C ABI directly; these ops
are the same entry points for callers that want schema'd operators instead of nn.Modules.
"""
import ctypes as C
from functools import lru_cache
from typing import List

import torch

from . import _lib
from .mixste2 import (FOLDED_LINEAR, FUSED_MLP_DEFAULT_WIDTHS, MixSTE2, SPLIT_SUFFIXES, _ptr, fill_weights_struct, folded_linear,
                      fused_mlp_fc2_image, head_major_qkv, image_layout, sinusoid_frequencies, split_image)

BLOCK_KEYS = ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
              "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need(cond, msg):
    """host-side operand check: a kernel launched on mismatched shapes faults on the device"""
    if not cond:
        raise ValueError(msg)


@lru_cache(maxsize=None)
def mixste_param_names(frames, joints, channels, depth, heads):
    """named_parameters() order of a MixSTE2 of these dimensions (built on the meta device: no storage)."""
    with torch.device("meta"):
        m = MixSTE2(num_frame=frames, num_joints=joints, in_chans=5, embed_dim_ratio=channels, depth=depth,
                    num_heads=heads, is_train=False)
    return tuple(n for n, _ in m.named_parameters())


@lru_cache(maxsize=None)
def mixste_param_shapes(frames, joints, channels, depth, heads):
    with torch.device("meta"):
        m = MixSTE2(num_frame=frames, num_joints=joints, in_chans=5, embed_dim_ratio=channels, depth=depth,
                    num_heads=heads, is_train=False)
    return tuple(tuple(p.shape) for _, p in m.named_parameters())


_freq_cache = {}


def _freqs(channels, device):
    key = (channels, device)
    if key not in _freq_cache:
        _freq_cache[key] = sinusoid_frequencies(channels).to(device)
    return _freq_cache[key]


PRECISIONS = {"f32": 0, "bf16": 1, "bf16x3": 2, "f16x2": 3, "bf16x3_images": 4}     # as pafuse_amd.D3DP.PRECISIONS
_image_cache = {}        # (data_ptr, _version, device, layout, shape, scheme) of a linear weight -> (its split image, the weight)
_IMAGE_CACHE_MAX = 2048  # ~ three models' worth of linear weights; the oldest entries go first


def cached_split_image(weight, layout, f16=False, x=False):
    """The pre-split image of a linear weight, made once per (storage, version, layout, scheme): the schema'd ops take plain
    parameter tensors, so they build and cache the images the split-precision kernels read (the modules keep theirs per
    module).  layout: pafuse_split_weights' (0 fc1, 1 proj / fc2, 2 qkv).
    An entry keeps a reference to the tensor it was made from: while the entry lives that storage cannot be freed, so no
    other tensor can come to sit at its address with a matching version and shape and hit a stale image (a freed temporary
    such as `w.float()` or `w.detach().clone()` made per call would otherwise do exactly that)."""
    key = (weight.data_ptr(), weight._version, weight.device, int(layout), tuple(weight.shape), bool(f16), bool(x))
    hit = _image_cache.get(key)
    if hit is None:
        while len(_image_cache) >= _IMAGE_CACHE_MAX:
            _image_cache.pop(next(iter(_image_cache)))
        hit = _image_cache[key] = (split_image(weight, layout, f16, x), weight)
    return hit[0]


def cached_folded_linear(table, name, f16=False, x=False, image=True):
    """(image, ls, lt) of a qkv / fc1 layer with its LayerNorm folded in (mixste2.folded_linear), made once per version of
    the four tensors it is built from - the modules' own default in the split modes, so the ops return the modules' bits.
    The entry pins its four source tensors (see cached_split_image)."""
    block, norm = name.rsplit(".", 3)[0], FOLDED_LINEAR[name.split(".", 2)[2]]
    parts = (table[name], table[f"{block}.{norm}.weight"], table[f"{block}.{norm}.bias"], table[name[:-len("weight")] + "bias"])
    key = ("fold", bool(f16), bool(x), bool(image)) + tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in parts) + (parts[0].device,)
    hit = _image_cache.get(key)
    if hit is None:
        while len(_image_cache) >= _IMAGE_CACHE_MAX:
            _image_cache.pop(next(iter(_image_cache)))
        hit = _image_cache[key] = (folded_linear(table.__getitem__, name, f16, x, image), parts)
    return hit[0]


def cached_head_major_qkv(table, name, heads, f16=False, x=False):
    """(image, hb, hl) of a qkv layer for the fused qkv + attention kernel (mixste2.head_major_qkv, LayerNorm folded), made once
    per version of the four tensors it is built from; the entry pins them (see cached_split_image)."""
    block, norm = name.rsplit(".", 3)[0], FOLDED_LINEAR[name.split(".", 2)[2]]
    parts = (table[name], table[f"{block}.{norm}.weight"], table[f"{block}.{norm}.bias"], table[name[:-len("weight")] + "bias"])
    key = ("head-major", bool(f16), bool(x), int(heads)) + tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in parts) + (parts[0].device,)
    hit = _image_cache.get(key)
    if hit is None:
        while len(_image_cache) >= _IMAGE_CACHE_MAX:
            _image_cache.pop(next(iter(_image_cache)))
        hit = _image_cache[key] = (head_major_qkv(table.__getitem__, name, heads, True, f16, x), parts)
    return hit[0]


def cached_fused_mlp_fc2(weight):
    """The fused MLP kernel's image of an fc2 weight (mixste2.fused_mlp_fc2_image), made once per (storage, version); the entry
    pins the tensor (see cached_split_image)."""
    key = ("fused-mlp", weight.data_ptr(), weight._version, weight.device, tuple(weight.shape))
    hit = _image_cache.get(key)
    if hit is None:
        while len(_image_cache) >= _IMAGE_CACHE_MAX:
            _image_cache.pop(next(iter(_image_cache)))
        hit = _image_cache[key] = (fused_mlp_fc2_image(weight), weight)
    return hit[0]


def _mode(precision):
    if precision not in PRECISIONS:
        raise _lib.PafuseError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
    return PRECISIONS[precision]


def mixste_struct(weights, frames, joints, depth, heads, precision="f32"):
    """pafuse_mixste2_weights from a flat parameter list; returns (struct, keep-alive)."""
    mode = _mode(precision)
    channels = weights[0].shape[-1]                           # Spatial_pos_embed [1,J,C] comes first
    if mode == 4 and not _lib.load().pafuse_mode_supported(4, channels, 0, heads, joints, frames):
        mode = 2                                              # as MixSTE2.effective_mode: the same products on the round-3 kernels
    names = mixste_param_names(frames, joints, channels, depth, heads)
    if len(weights) != len(names):
        raise _lib.PafuseError(f"expected {len(names)} weight tensors (named_parameters() order), got {len(weights)}")
    table = dict(zip(names, weights))
    for (n, t), shape in zip(table.items(), mixste_param_shapes(frames, joints, channels, depth, heads)):
        _need(tuple(t.shape) == shape, f"{n} must be {shape}, got {tuple(t.shape)}")
        _ptr(t, n)
    fr = _freqs(channels, weights[0].device)
    w = _lib.MixSTE2Weights()
    images = None
    if mode in (2, 3, 4):
        images = {}
        f16, x = mode == 3, mode == 4
        for n, t in table.items():
            plain = not (x and n.endswith("attn.qkv.weight"))      # mode 4 multiplies the head-major qkv image only
            if n.endswith(tuple(FOLDED_LINEAR)):    # the modules' default in every split mode: LayerNorm folded into qkv / fc1
                images[n], images[n[:-len("weight")] + "ls"], images[n[:-len("weight")] + "lt"] = cached_folded_linear(table, n, f16, x, plain)
                if not plain:
                    del images[n]
            elif n.endswith(SPLIT_SUFFIXES):
                images[n] = cached_split_image(t, image_layout(n), f16, x)
            if mode >= 3 and n.endswith("attn.qkv.weight"):     # the modules' default there: qkv + attention in one kernel
                stem = n[:-len("weight")]
                images[stem + "hs"], images[stem + "hb"], images[stem + "hl"] = cached_head_major_qkv(table, n, heads, f16, x)
            if mode == 3 and n.endswith("mlp.fc2.weight") and channels in FUSED_MLP_DEFAULT_WIDTHS and t.shape[1] == 2 * channels:
                images[n[:-len("weight")] + "hp"] = cached_fused_mlp_fc2(t)    # ... and the MLP in one kernel where the modules do
    fill_weights_struct(w, table.__getitem__, fr, frames, joints, channels, depth, heads, 5, mode, images)
    return w, (table, fr, images)


# ------------------------------------------------------------------------------------------------ unit ops
@torch.library.custom_op("pafuse::linear", mutates_args=(), device_types="cuda")
def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, gelu: bool = False) -> torch.Tensor:
    lib = _lib.load()
    K, N = x.shape[-1], weight.shape[0]
    _need(tuple(weight.shape) == (N, K) and bias.numel() == N, f"linear: x [..,{K}] needs weight [N,{K}] and bias [N]")
    x2 = x.contiguous().view(-1, K)
    out = torch.empty(x2.shape[0], N, device=x.device, dtype=torch.float32)
    _lib.check(lib.pafuse_linear(_ptr(x2, "x"), _ptr(weight, "weight"), _ptr(bias, "bias"), out.data_ptr(),
                                 x2.shape[0], N, K, int(gelu), _stream(x)))
    return out.view(*x.shape[:-1], N)


@linear.register_fake
def _(x, weight, bias, gelu=False):
    return x.new_empty(*x.shape[:-1], weight.shape[0])


@torch.library.custom_op("pafuse::layer_norm", mutates_args=(), device_types="cuda")
def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    lib = _lib.load()
    Cc = x.shape[-1]
    _need(weight.numel() == Cc and bias.numel() == Cc, f"layer_norm: weight and bias must have {Cc} elements")
    x2 = x.contiguous().view(-1, Cc)
    out = torch.empty_like(x2)
    _lib.check(lib.pafuse_layernorm(_ptr(x2, "x"), _ptr(weight, "weight"), _ptr(bias, "bias"), out.data_ptr(),
                                    x2.shape[0], Cc, eps, _stream(x)))
    return out.view(x.shape)


@layer_norm.register_fake
def _(x, weight, bias, eps):
    return torch.empty_like(x)


@torch.library.custom_op("pafuse::attention", mutates_args=(), device_types="cuda")
def attention(qkv: torch.Tensor, heads: int, seq_len: int, joints: int = 0) -> torch.Tensor:
    """qkv [M,3C] in the fixed (r,f,j) token order.  joints == 0: M/seq_len contiguous sequences (spatial,
    seq_len = J); joints > 0: temporal sequences of seq_len = F tokens at stride `joints`."""
    lib = _lib.load()
    M, C3 = qkv.shape
    Cc = C3 // 3
    _need(C3 % 3 == 0 and heads > 0 and Cc % heads == 0 and seq_len > 0 and joints >= 0,
          "attention: qkv must be [M, 3*heads*d]")
    if M % seq_len or (joints and M % (seq_len * joints)):
        raise _lib.PafuseError(f"attention: {M} rows do not split into sequences of {seq_len}")
    o = torch.empty(M, Cc, device=qkv.device, dtype=torch.float32)
    if joints:
        args = (M // seq_len, seq_len, Cc, heads, joints, seq_len * joints, 1, joints)
    else:
        args = (M // seq_len, seq_len, Cc, heads, 1, seq_len, 0, 1)
    _lib.check(lib.pafuse_attention(_ptr(qkv, "qkv"), o.data_ptr(), *args, _stream(qkv)))
    return o


@attention.register_fake
def _(qkv, heads, seq_len, joints=0):
    return qkv.new_empty(qkv.shape[0], qkv.shape[1] // 3)


@torch.library.custom_op("pafuse::block", mutates_args=(), device_types="cuda")
def block(x: torch.Tensor, weights: List[torch.Tensor], heads: int, precision: str = "bf16x3") -> torch.Tensor:
    """Block.forward on [S,L,C]: S sequences of L tokens (spatial: L = J; temporal: the caller passes [.., F, C]).
    precision: matrix-product mode of the linear layers, as pafuse_amd.D3DP.precision (default: the inference default)."""
    lib = _lib.load()
    mode = _mode(precision)
    if len(weights) != len(BLOCK_KEYS):
        raise _lib.PafuseError(f"block: expected {len(BLOCK_KEYS)} tensors in order {BLOCK_KEYS}")
    S, L, Cc = x.shape
    shapes = ((Cc,), (Cc,), (3 * Cc, Cc), (3 * Cc,), (Cc, Cc), (Cc,), (Cc,), (Cc,), (2 * Cc, Cc), (2 * Cc,), (Cc, 2 * Cc), (Cc,))
    for t, shape, name in zip(weights, shapes, BLOCK_KEYS):
        _need(tuple(t.shape) == shape, f"block: {name} must be {shape}, got {tuple(t.shape)}")
    _need(heads > 0 and Cc % heads == 0, "block: heads must divide the width")
    y = x.contiguous().clone()
    w = _lib.BlockWeights()
    for field, t, name in zip(_lib.BLOCK_FIELDS, weights, BLOCK_KEYS):
        setattr(w, field, _ptr(t, name))
    images = []
    if mode == 4 and not lib.pafuse_mode_supported(4, Cc, 0, heads, L, L):
        mode = 2
    if mode in (2, 3, 4):
        for field, idx, layout in (("qkv_ws", 2, 2), ("proj_ws", 4, 1), ("fc1_ws", 8, 2), ("fc2_ws", 10, 1)):
            if mode == 4 and field == "qkv_ws":
                continue
            images.append(cached_split_image(weights[idx], layout, mode == 3, mode == 4))
            setattr(w, field, images[-1].data_ptr())
    if mode == 4:       # qkv + attention is one kernel there: the head-major image (LayerNorm not folded: Block.forward normalises)
        table = {"b.0.attn.qkv.weight": weights[2], "b.0.attn.qkv.bias": weights[3]}
        hs, hb, _ = head_major_qkv(table.__getitem__, "b.0.attn.qkv.weight", heads, False, False, True)
        images += [hs, hb]
        w.qkv_hs, w.qkv_hb = hs.data_ptr(), hb.data_ptr()
    nbytes = lib.pafuse_block_workspace_bytes(S * L, Cc)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.check(lib.pafuse_block_forward(C.byref(w), _ptr(y, "x"), S, L, Cc, heads, mode, ws.data_ptr(), nbytes, _stream(x)))
    return y


@block.register_fake
def _(x, weights, heads, precision="bf16x3"):
    return torch.empty_like(x)


# ------------------------------------------------------------------------------------------- model-level ops
@torch.library.custom_op("pafuse::mixste_eval", mutates_args=(), device_types="cuda")
def mixste_eval(x2d: torch.Tensor, x3d: torch.Tensor, t: torch.Tensor, weights: List[torch.Tensor], depth: int,
                heads: int, precision: str = "bf16x3") -> torch.Tensor:
    lib = _lib.load()
    _need(x3d.dim() == 5 and x3d.shape[-1] == 3, "mixste_eval: x3d must be [B,P,F,J,3]")
    B, P, F, J, _ = x3d.shape
    _need(tuple(x2d.shape) == (B, F, J, 2) and tuple(t.shape) == (B,), "mixste_eval: x2d must be [B,F,J,2], t [B]")
    w, keep = mixste_struct(weights, F, J, depth, heads, precision)
    x2d, x3d, t = x2d.contiguous().float(), x3d.contiguous().float(), t.contiguous().long()
    out = torch.empty(B, P, F, J, 3, device=x3d.device, dtype=torch.float32)
    nbytes = lib.pafuse_mixste2_workspace_bytes(C.byref(w), B, P)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x3d.device)
    _lib.check(lib.pafuse_mixste2_forward(C.byref(w), _ptr(x2d, "x2d"), _ptr(x3d, "x3d"), t.data_ptr(), B, P,
                                          out.data_ptr(), ws.data_ptr(), nbytes, _stream(x3d)))
    return out


@mixste_eval.register_fake
def _(x2d, x3d, t, weights, depth, heads, precision="bf16x3"):
    return torch.empty_like(x3d)


@torch.library.custom_op("pafuse::ddim_loop", mutates_args=(), device_types="cuda")
def ddim_loop(x2d: torch.Tensor, x2d_flip: torch.Tensor, noise: torch.Tensor, weights: List[torch.Tensor],
              part_joints: List[torch.Tensor], flip_perm: torch.Tensor, depth: int, heads: int, times: List[int],
              sched: List[float], flip: bool, scale: float, precision: str = "bf16x3") -> torch.Tensor:
    """The whole sampler.  noise [n_draws,B,P,F,J,3] in the reference's draw order; weights = the per-part
    parameter lists concatenated in part order; part_joints[i] int32 joint indices of part i; flip_perm int32 [J];
    times[k] the k-th timestep (the last one is the step whose time_next < 0); sched 5 doubles per step:
    sqrt_recip_alphas_cumprod[t], sqrt_recipm1_alphas_cumprod[t], sqrt(alpha_next), c, sigma.  precision: the
    matrix-product mode of every part's linear layers (default = the modules' inference default).  -> [B,T,P,F,J,3]"""
    lib = _lib.load()
    _mode(precision)
    _need(noise.dim() == 6 and noise.shape[-1] == 3, "ddim_loop: noise must be [n_draws,B,P,F,J,3]")
    n_draws, B, P, F, J, _ = noise.shape
    _need(tuple(x2d.shape) == (B, F, J, 2) and (not flip or tuple(x2d_flip.shape) == (B, F, J, 2)),
          "ddim_loop: x2d / x2d_flip must be [B,F,J,2]")
    _need(flip_perm.numel() == J and int(flip_perm.min()) >= 0 and int(flip_perm.max()) < J, "ddim_loop: flip_perm")
    _need(all(int(i.min()) >= 0 and int(i.max()) < J for i in part_joints), "ddim_loop: joint index out of range")
    T = len(times)
    if len(sched) != 5 * T or len(part_joints) > _lib.MAX_PARTS:
        raise _lib.PafuseError("ddim_loop: sched needs 5 values per step; at most %d parts" % _lib.MAX_PARTS)
    dev = noise.device
    cfg = _lib.D3DPConfig()
    cfg.num_parts, cfg.num_kps, cfg.frames, cfg.flip, cfg.scale = len(part_joints), J, F, int(flip), float(scale)
    keep, at = [], 0
    joint_part = torch.full((J,), -1, dtype=torch.int32)
    joint_local = torch.zeros(J, dtype=torch.int32)
    for i, idx in enumerate(part_joints):
        Jp = idx.numel()
        n = len(mixste_param_names(F, Jp, weights[at].shape[-1], depth, heads))
        w, k = mixste_struct(weights[at:at + n], F, Jp, depth, heads, precision)
        at += n
        idx32 = idx.to(device=dev, dtype=torch.int32).contiguous()
        cfg.part[i], cfg.part_joints[i] = w, idx32.data_ptr()
        host = idx.cpu().long()
        joint_part[host] = i
        joint_local[host] = torch.arange(Jp, dtype=torch.int32)
        keep += [k, idx32]
    if at != len(weights) or int(joint_part.min()) < 0:
        raise _lib.PafuseError("ddim_loop: weights / part_joints do not cover the model")
    joint_part, joint_local = joint_part.to(dev), joint_local.to(dev)
    perm = flip_perm.to(device=dev, dtype=torch.int32).contiguous()
    cfg.joint_part, cfg.joint_local, cfg.flip_perm = joint_part.data_ptr(), joint_local.data_ptr(), perm.data_ptr()
    # one stream, every layer part by part: the launches (and so the bits) of the modules' three-stream schedule - the shared grids
    # of the library's own single-stream schedule keep another whole-row epilogue (same function, rounding-level differences)
    cfg.part_by_part_launches = 1
    steps = (_lib.DDIMStep * T)()
    for k in range(T):
        st = steps[k]
        st.time, st.last = int(times[k]), int(k == T - 1)
        st.sqrt_recip_acp, st.sqrt_recipm1_acp, st.sqrt_alpha_next, st.c, st.sigma = sched[5 * k:5 * k + 5]
    if n_draws < T:
        raise _lib.PafuseError(f"ddim_loop: {T} steps need {T} noise draws, got {n_draws}")
    x2d = x2d.contiguous().float()
    x2f = x2d_flip.contiguous().float() if flip else x2d
    noise = noise.contiguous().float()
    out = torch.empty(B, T, P, F, J, 3, device=dev, dtype=torch.float32)
    nbytes = lib.pafuse_d3dp_workspace_bytes(C.byref(cfg), B, P)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.pafuse_d3dp_sample(C.byref(cfg), steps, T, _ptr(x2d, "x2d"), _ptr(x2f, "x2d_flip"),
                                      _ptr(noise, "noise"), n_draws, B, P, out.data_ptr(), ws.data_ptr(), nbytes,
                                      _stream(noise), None, 0))
    return out


@ddim_loop.register_fake
def _(x2d, x2d_flip, noise, weights, part_joints, flip_perm, depth, heads, times, sched, flip, scale, precision="bf16x3"):
    n_draws, B, P, F, J, _ = noise.shape
    return noise.new_empty(B, len(times), P, F, J, 3)


def ddim_loop_args(model):
    """The (weights, part_joints, flip_perm, depth, heads, times, sched, flip, scale) tail of pafuse::ddim_loop for a
    pafuse_amd.D3DP instance."""
    weights, joints = [], []
    for part, m in model.denoisers().items():
        weights += list(m.parameters())
        joints.append(getattr(model, f"_joints_{part}"))
    first = next(iter(model.denoisers().values()))
    steps = model.ddim_steps()
    times = [int(s.time) for s in steps]
    sched = [v for s in steps for v in (s.sqrt_recip_acp, s.sqrt_recipm1_acp, s.sqrt_alpha_next, s.c, s.sigma)]
    return (weights, joints, model._flip_perm, first.block_depth, first.num_heads, times, sched, bool(model.flip),
            float(model.scale))
