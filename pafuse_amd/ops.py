"""Thin tensor-level wrappers over the unit entry points of the C ABI (used by the parity tests).

Every wrapper checks operand shapes on the host and launches with the operands' device current."""
import ctypes as C

import torch

from . import _lib
from .mixste2 import _ptr, fill_block_struct, split_image


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need(cond, msg):
    """host-side operand check: a kernel launched on mismatched shapes faults on the device"""
    if not cond:
        raise ValueError(msg)


def linear(x, weight, bias, act=None, bf16=False):
    """nn.Linear (+ exact GELU when act == 'gelu') on the last dim of a contiguous fp32 tensor; bf16=True rounds both
    operands to bf16 in registers (fp32 accumulate)."""
    lib = _lib.load()
    K = x.shape[-1]
    N = weight.shape[0]
    _need(tuple(weight.shape) == (N, K) and bias.numel() == N, f"linear: x [..,{K}] needs weight [N,{K}] and bias [N]")
    x2 = x.contiguous().view(-1, K)
    out = torch.empty(x2.shape[0], N, device=x.device, dtype=torch.float32)
    if x2.shape[0] == 0:      # no rows: nothing to launch (an empty tensor has no storage to point the library at)
        return out.view(*x.shape[:-1], N)
    with torch.cuda.device(x.device):
        _lib.check(lib.pafuse_linear(_ptr(x2, "x"), _ptr(weight, "weight"), _ptr(bias, "bias"), out.data_ptr(),
                                     x2.shape[0], N, K, (1 if act == "gelu" else 0) | (2 if bf16 else 0), _stream(x)))
    return out.view(*x.shape[:-1], N)


def layer_norm(x, weight, bias, eps):
    lib = _lib.load()
    Cc = x.shape[-1]
    _need(weight.numel() == Cc and bias.numel() == Cc, f"layer_norm: weight and bias must have {Cc} elements")
    x2 = x.contiguous().view(-1, Cc)
    out = torch.empty_like(x2)
    if x2.shape[0] == 0:
        return out.view_as(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.pafuse_layernorm(_ptr(x2, "x"), _ptr(weight, "w"), _ptr(bias, "b"), out.data_ptr(), x2.shape[0],
                                        Cc, eps, _stream(x)))
    return out.view_as(x)


def attention(qkv, heads, nseq, L, group=1, group_stride=None, seq_stride=0, tok_stride=1):
    """softmax(q k^T d^-1/2) v on a [M,3C] qkv matrix; default addressing = contiguous sequences of L rows."""
    lib = _lib.load()
    M, C3 = qkv.shape
    Cc = C3 // 3
    gs = L if group_stride is None else group_stride
    _need(C3 % 3 == 0 and Cc % heads == 0 and nseq >= 0 and group >= 1, "attention: qkv must be [M, 3*heads*d]")
    if nseq:   # the last row any sequence touches must exist
        last = ((nseq - 1) // group) * gs + ((nseq - 1) % group) * seq_stride + (L - 1) * tok_stride
        _need(0 <= last < M and min(gs, seq_stride, tok_stride) >= 0, f"attention: sequences reach row {last} of {M}")
    o = torch.zeros(M, Cc, device=qkv.device, dtype=torch.float32)
    if M == 0 or nseq == 0:
        return o
    with torch.cuda.device(qkv.device):
        _lib.check(lib.pafuse_attention(_ptr(qkv, "qkv"), o.data_ptr(), nseq, L, Cc, heads, group,
                                        L if group_stride is None else group_stride, seq_stride, tok_stride,
                                        _stream(qkv)))
    return o


def attention_backward(qkv, d_o, heads, nseq, L, group=1, group_stride=None, seq_stride=0, tok_stride=1):
    """dqkv [M,3C] of attention(): the training step's attention backward kernel on its own (rows no sequence touches: 0)."""
    lib = _lib.load()
    M, C3 = qkv.shape
    Cc = C3 // 3
    gs = L if group_stride is None else group_stride
    _need(C3 % 3 == 0 and Cc % heads == 0 and nseq >= 0 and group >= 1 and tuple(d_o.shape) == (M, Cc),
          "attention_backward: qkv must be [M, 3*heads*d] and d_o [M, heads*d]")
    if nseq:
        last = ((nseq - 1) // group) * gs + ((nseq - 1) % group) * seq_stride + (L - 1) * tok_stride
        _need(0 <= last < M and min(gs, seq_stride, tok_stride) >= 0, f"attention_backward: sequences reach row {last} of {M}")
    dqkv = torch.zeros(M, C3, device=qkv.device, dtype=torch.float32)
    with torch.cuda.device(qkv.device):
        _lib.check(lib.pafuse_attention_backward(_ptr(qkv, "qkv"), _ptr(d_o, "d_o"), dqkv.data_ptr(), nseq, L, Cc, heads, group,
                                                 gs, seq_stride, tok_stride, _stream(qkv)))
    return dqkv


def linear_weight_grad(d_y, x, precision="bf16x3", bias=True):
    """(dW [N,K], db [N] or None) = (d_y^T x, column sums of d_y) for d_y [M,N], x [M,K]: the weight / bias gradient kernels of
    the training step on their own; precision 'bf16x3' (split products) or 'f32'."""
    lib = _lib.load()
    _need(precision in ("bf16x3", "f32"), "linear_weight_grad: precision 'bf16x3' or 'f32'")
    M, N = d_y.shape
    K = x.shape[1]
    _need(x.shape[0] == M, "linear_weight_grad: d_y [M,N] and x [M,K]")
    dw = torch.zeros(N, K, device=x.device, dtype=torch.float32)
    db = torch.zeros(N, device=x.device, dtype=torch.float32) if bias else None
    nbytes = lib.pafuse_linear_weight_grad_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.pafuse_linear_weight_grad(_ptr(d_y, "d_y"), _ptr(x, "x"), dw.data_ptr(), db.data_ptr() if bias else None, M, N, K,
                                                 PRECISIONS[precision], ws.data_ptr(), nbytes, _stream(x)))
    return dw, db


PRECISIONS = {"f32": 0, "bf16": 1, "bf16x3": 2, "f16x2": 3, "bf16x3_images": 4}       # as pafuse_amd.D3DP.PRECISIONS


def block_forward(block_params, x, heads=8, precision="f32"):
    """Block.forward (common/mixste.py:113-116) on [S,L,C]; ``block_params`` is a pafuse_amd.mixste2._BlockParams;
    ``precision`` the matrix-product mode of its four linear layers (the split modes make their weight images here;
    'bf16x3_images' runs the image pipeline where it has kernels for the shape, else the round-3 kernels)."""
    lib = _lib.load()
    S, L, Cc = x.shape
    _need(block_params.norm1.weight.numel() == Cc and Cc % heads == 0, f"block: parameters are not for width {Cc}")
    _need(precision in PRECISIONS, f"precision must be one of {sorted(PRECISIONS)}")
    y = x.contiguous().clone()
    w = _lib.BlockWeights()
    fill_block_struct(w, block_params)
    images = []
    mode = PRECISIONS[precision]
    if mode == 4 and not lib.pafuse_mode_supported(4, Cc, 0, heads, L, L):
        mode = 2
    if mode >= 2:
        for field, lin, layout in (("qkv_ws", block_params.attn.qkv, 2), ("proj_ws", block_params.attn.proj, 1),
                                   ("fc1_ws", block_params.mlp.fc1, 2), ("fc2_ws", block_params.mlp.fc2, 1)):
            if mode == 4 and field == "qkv_ws":
                continue
            images.append(split_image(lin.weight, layout, mode == 3, mode == 4))
            setattr(w, field, images[-1].data_ptr())
    if mode == 4:       # qkv + attention is one kernel there: the head-major image (LayerNorm not folded: Block.forward normalises)
        from .mixste2 import head_major_qkv
        table = {"b.0.attn.qkv.weight": block_params.attn.qkv.weight, "b.0.attn.qkv.bias": block_params.attn.qkv.bias}
        hs, hb, _ = head_major_qkv(table.__getitem__, "b.0.attn.qkv.weight", heads, False, False, True)
        images += [hs, hb]
        w.qkv_hs, w.qkv_hb = hs.data_ptr(), hb.data_ptr()
    nbytes = lib.pafuse_block_workspace_bytes(S * L, Cc)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.pafuse_block_forward(C.byref(w), y.data_ptr(), S, L, Cc, heads, mode, ws.data_ptr(),
                                            nbytes, _stream(x)))
    del images      # (stream-ordered: the caching allocator reuses the blocks only for later work on this stream)
    return y


def time_embed(model, t):
    """MixSTE2.time_mlp(t) for a pafuse_amd.MixSTE2."""
    lib = _lib.load()
    w = model.weights_struct()
    _need(t.dim() == 1, "time_embed: t must be [B]")
    t = t.contiguous().long()
    out = torch.empty(t.shape[0], model.embed_dim, device=t.device, dtype=torch.float32)
    hid = torch.empty(t.shape[0], 2 * model.embed_dim, device=t.device, dtype=torch.float32)
    with torch.cuda.device(t.device):
        _lib.check(lib.pafuse_time_embed(C.byref(w), t.data_ptr(), t.shape[0], out.data_ptr(), hid.data_ptr(), _stream(t)))
    return out


def embed(x3d, x2d, patch_w, patch_b, pos_spatial, temb, norm_w, norm_b, norm_eps=1e-6, joints=None, perm=None,
          x2d_flip=None, do_clamp=False, scale=1.0):
    """Input stage of one body part (pafuse_embed, include/pafuse_hip.h): clamp / scale / flip / part gather of the
    noised 3-D pose, 2-D broadcast over P, Linear(5->C) + pos-embed + time-embed -> x, and LayerNorm(x) -> xn.

    x3d [B,P,F,J3,3], x2d [B,F,J3,2]; ``joints`` = this part's indices into the J3 axis (None: all joints);
    ``x2d_flip`` + ``perm`` switch on the flipped half (rows of the second half).  Returns (x, xn) as
    [nflip, B, P, F, J, C]."""
    lib = _lib.load()
    B, P, F, J3, _ = x3d.shape
    Cc = patch_w.shape[0]
    nflip = 2 if x2d_flip is not None else 1
    J = J3 if joints is None else int(joints.numel())
    _need(tuple(x3d.shape) == (B, P, F, J3, 3) and tuple(x2d.shape) == (B, F, J3, 2), "embed: x3d [B,P,F,J3,3], x2d [B,F,J3,2]")
    _need(tuple(patch_w.shape) == (Cc, 5) and patch_b.numel() == Cc and tuple(pos_spatial.shape[-2:]) == (J, Cc) and
          tuple(temb.shape) == (B, Cc) and norm_w.numel() == Cc and norm_b.numel() == Cc, "embed: parameter shapes")
    idx = {}
    for name, t, n in (("joints", joints, J), ("perm", perm, J3)):
        if t is not None:
            _need(t.dtype == torch.int32 and t.is_cuda and t.is_contiguous() and t.numel() == n, f"embed: {name} int32 [{n}]")
            _need(int(t.min()) >= 0 and int(t.max()) < J3, f"embed: {name} values must index the {J3} joints")
            idx[name] = t.data_ptr()
    if nflip == 2:
        _need(perm is not None and tuple(x2d_flip.shape) == (B, F, J3, 2), "embed: the flipped half needs perm and x2d_flip")
    x = torch.empty(nflip, B, P, F, J, Cc, device=x3d.device, dtype=torch.float32)
    xn = torch.empty_like(x)
    with torch.cuda.device(x3d.device):
        _lib.check(lib.pafuse_embed(_ptr(x3d, "x3d"), _ptr(x2d, "x2d"), _ptr(x2d_flip, "x2d_flip") if nflip == 2 else None,
                                    idx.get("joints"), idx.get("perm"), _ptr(patch_w, "patch_w"), _ptr(patch_b, "patch_b"),
                                    _ptr(pos_spatial, "pos"), _ptr(temb, "temb"), _ptr(norm_w, "norm_w"),
                                    _ptr(norm_b, "norm_b"), float(norm_eps), B, P, F, J, J3, Cc, nflip, int(do_clamp),
                                    float(scale), x.data_ptr(), xn.data_ptr(), _stream(x3d)))
    return x, xn


def ddim_finalize(preds, joint_part, joint_local, img, step_scalars, noise=None, flip_perm=None, scale=1.0, T=1, step=0,
                  out=None):
    """Output stage of one DDIM step (pafuse_ddim_finalize): concat of the parts' predictions, un-flip + TTA mean,
    scale + clamp -> x_start (written to out[:, step]); epsilon in fp64; img update in place.

    preds: list of per-part tensors [nflip,B,P,F,Jp,3]; img [B,P,F,J,3] (updated in place); step_scalars: a
    ``_lib.DDIMStep``.  Returns (out [B,T,P,F,J,3], img)."""
    lib = _lib.load()
    B, P, F, J, _ = img.shape
    flip = flip_perm is not None
    nflip = 2 if flip else 1
    for p in preds:
        _need(tuple(p.shape[:4]) == (nflip, B, P, F) and p.shape[-1] == 3, f"ddim_finalize: prediction {tuple(p.shape)}")
    _need(sum(p.shape[4] for p in preds) == J, "ddim_finalize: the parts must cover every joint")
    for name, t in (("joint_part", joint_part), ("joint_local", joint_local), ("flip_perm", flip_perm)):
        if t is not None:
            _need(t.dtype == torch.int32 and t.is_cuda and t.is_contiguous() and t.numel() == J, f"ddim_finalize: {name}")
    _need(int(joint_part.min()) >= 0 and int(joint_part.max()) < len(preds), "ddim_finalize: joint_part range")
    for i, p in enumerate(preds):
        sel = joint_local[joint_part == i]
        _need(sel.numel() == p.shape[4] and int(sel.max()) < p.shape[4] and int(sel.min()) >= 0,
              f"ddim_finalize: joint_local of part {i}")
    if flip:
        _need(int(flip_perm.min()) >= 0 and int(flip_perm.max()) < J, "ddim_finalize: flip_perm range")
    if not step_scalars.last:
        _need(noise is not None and tuple(noise.shape) == tuple(img.shape), "ddim_finalize: an update step needs noise")
    if out is None:
        out = torch.zeros(B, T, P, F, J, 3, device=img.device, dtype=torch.float32)
    _need(tuple(out.shape) == (B, T, P, F, J, 3) and 0 <= step < T, "ddim_finalize: out [B,T,P,F,J,3]")
    n = len(preds)
    ptrs = (C.c_void_p * n)(*[_ptr(p, "pred") for p in preds])
    counts = (C.c_int32 * n)(*[p.shape[4] for p in preds])
    with torch.cuda.device(img.device):
        _lib.check(lib.pafuse_ddim_finalize(ptrs, counts, n, joint_part.data_ptr(), joint_local.data_ptr(),
                                            flip_perm.data_ptr() if flip else None, _ptr(img, "img"),
                                            _ptr(noise, "noise") if noise is not None else None, _ptr(out, "out"),
                                            B, P, F, J, T, step, int(flip), float(scale), C.byref(step_scalars),
                                            _stream(img)))
    return out, img


def hsplit_rows(x):
    """The activation H image of a contiguous fp32 [R,K] tensor (pafuse_hsplit_rows): per sub-block of 8 k the fp16 slices
    hi = f16(a) and lo = f16((a - hi) 2^11), 4 R K bytes.  In the loop every producer writes this form itself; the unit tests
    and pafuse_block_forward's first LayerNorm use this pass."""
    lib = _lib.load()
    _need(x.dim() == 2 and x.shape[1] % 8 == 0, "hsplit_rows: x must be [R, K] with K % 8 == 0")
    R, K = x.shape
    out = torch.empty(R * K * 4, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.pafuse_hsplit_rows(_ptr(x, "x"), R, K, out.data_ptr(), _stream(x)))
    return out


def xsplit_rows(x):
    """The X image of a contiguous fp32 [R,K] tensor (pafuse_xsplit_rows): [R][K/32][3][32 x bf16], the three exact bf16 slices
    of every element, 6 R K bytes."""
    lib = _lib.load()
    _need(x.dim() == 2 and x.shape[1] % 32 == 0, "xsplit_rows: x must be [R, K] with K % 32 == 0")
    R, K = x.shape
    out = torch.empty(R * K * 6, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.pafuse_xsplit_rows(_ptr(x, "x"), R, K, out.data_ptr(), _stream(x)))
    return out


def xjoin_rows(image, R, K):
    """fp32 [R,K] tensor an X image stands for: s0 + (s1 + s2), exact in fp32."""
    v = image.view(torch.bfloat16).view(R, K // 32, 3, 32).float()
    return (v[:, :, 0] + (v[:, :, 1] + v[:, :, 2])).reshape(R, K)


def hjoin_rows(image, R, K):
    """fp32 [R,K] tensor an activation H image stands for: hi + 2^-11 lo (exact in fp32)."""
    v = image.view(torch.float16).view(R, K // 8, 2, 8).float()
    return (v[:, :, 0] + v[:, :, 1] * 2.0 ** -11).reshape(R, K)


def qkv_attention_fused(x, qkv_weight, qkv_bias, heads, nseq, L, scheme="bf16x3_images", rstd=None, group=1, group_stride=None,
                        seq_stride=0, tok_stride=1):
    """The fused qkv + attention kernel on its own (pafuse_qkv_attention_image): x [M,C] fp32 (with `rstd` [M]: the centred rows of
    a folded LayerNorm) -> o [M,C] fp32 decoded from the image the kernel writes; q | k | v = rstd (x W^T) + b per head, then
    softmax(q k^T d^-1/2) v per (sequence, head).  scheme 'f16x2' (hfqa_kernel, H images) or 'bf16x3_images' (xfqa_kernel, X
    images).  Sequence map as `attention`."""
    from .mixste2 import head_major_qkv
    lib = _lib.load()
    M, Cw = x.shape
    _need(scheme in ("f16x2", "bf16x3_images"), "qkv_attention_fused: scheme 'f16x2' or 'bf16x3_images'")
    _need(tuple(qkv_weight.shape) == (3 * Cw, Cw) and qkv_bias.numel() == 3 * Cw and Cw % heads == 0, "qkv_attention_fused: qkv must be [3C, C] + [3C]")
    gs = L if group_stride is None else group_stride
    if nseq:   # the last row any sequence touches must exist
        last = ((nseq - 1) // group) * gs + ((nseq - 1) % group) * seq_stride + (L - 1) * tok_stride
        _need(0 <= last < M and min(gs, seq_stride, tok_stride) >= 0, f"qkv_attention_fused: sequences reach row {last} of {M}")
    f16 = scheme == "f16x2"
    table = {"b.0.attn.qkv.weight": qkv_weight.contiguous(), "b.0.attn.qkv.bias": qkv_bias.contiguous()}
    hs, hb, _ = head_major_qkv(table.__getitem__, "b.0.attn.qkv.weight", heads, False, f16, not f16)
    xi = hsplit_rows(x.contiguous()) if f16 else xsplit_rows(x.contiguous())
    stats = None
    if rstd is not None:
        stats = torch.stack([torch.zeros_like(rstd), rstd], dim=1).contiguous()
    eb = 4 if f16 else 6
    oi = torch.zeros(M * Cw * eb, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.pafuse_qkv_attention_image(4 if f16 else 8, xi.data_ptr(), stats.data_ptr() if stats is not None else None,
                                                  hs.data_ptr(), _ptr(hb, "qkv_hb"), oi.data_ptr(), M, nseq, L, Cw, heads, group, gs,
                                                  seq_stride, tok_stride, 0.0, _stream(x)))
    return hjoin_rows(oi, M, Cw) if f16 else xjoin_rows(oi, M, Cw)


def mlp_fused(xc, rstd, fc1_weight, fc1_bias, fc2_weight, fc2_bias, eps=1e-6, in_place=False):
    """The fused MLP kernel on its own (pafuse_mlp_h): xc [M,C] = rows of the residual stream centred on their means, rstd [M] or
    None -> (y - mean(y) as fp32 [M,C] decoded from the H image the kernel writes, stats [M,2] = (mean(y), rstd(y))) with
    y = xc + GELU(rstd (xc W1^T) + b1) W2^T + b2."""
    from .mixste2 import fused_mlp_fc2_image, split_image
    lib = _lib.load()
    M, Cw = xc.shape
    _need(tuple(fc1_weight.shape) == (2 * Cw, Cw) and tuple(fc2_weight.shape) == (Cw, 2 * Cw), "mlp_fused: weights must be [2C,C] and [C,2C]")
    xh = hsplit_rows(xc.contiguous())
    stats_in = None
    if rstd is not None:
        stats_in = torch.stack([torch.zeros_like(rstd), rstd], dim=1).contiguous()
    w1h = split_image(fc1_weight.contiguous(), 0, True)
    w2hp = fused_mlp_fc2_image(fc2_weight)
    out = xh if in_place else torch.empty(M * Cw * 4, dtype=torch.uint8, device=xc.device)
    stats = stats_in if in_place and stats_in is not None else torch.empty(M, 2, device=xc.device, dtype=torch.float32)
    with torch.cuda.device(xc.device):
        _lib.check(lib.pafuse_mlp_h(xh.data_ptr(), stats_in.data_ptr() if stats_in is not None else None, w1h.data_ptr(),
                                    _ptr(fc1_bias, "fc1_bias"), w2hp.data_ptr(), _ptr(fc2_bias, "fc2_bias"), out.data_ptr(),
                                    stats.data_ptr(), M, Cw, float(eps), _stream(xc)))
    return hjoin_rows(out, M, Cw), stats


SCHEME_FLAG = {"bf16x3": 0, "f16x2": 4, "bf16x3_images": 8}      # include/pafuse_hip.h: PAFUSE_SPLIT_F16X2, PAFUSE_SPLIT_X


def split_weights(weight, layout=0, scheme="bf16x3"):
    """Pre-split image of a [N,K] fp32 weight for the split-precision products (pafuse_split_weights);
    scheme 'bf16x3_images' (the X image of the image pipeline: three bf16 slices, one geometry for every layer), 'f16x2' (the H image:
    two fp16 slices of the power-of-two-scaled weight) or 'bf16x3' (the round-3 kernels' images; layout 0: the
    32x32x16-MFMA plain kernel (mlp.fc1), 2: the 16x16x32-MFMA kernel of the qkv layers)."""
    lib = _lib.load()
    _need(weight.dim() == 2 and weight.shape[1] % 32 == 0, "split_weights: weight must be [N, K] with K % 32 == 0")
    _need(layout in (0, 2), "split_weights: layout 0 (fc1 kernel) or 2 (qkv kernel)")
    _need(scheme in SCHEME_FLAG, f"split_weights: scheme must be one of {sorted(SCHEME_FLAG)}")
    N, K = weight.shape
    img = torch.empty(lib.pafuse_split_image_bytes(N, K, layout | SCHEME_FLAG[scheme]), dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.check(lib.pafuse_split_weights(_ptr(weight, "weight"), N, K, layout | SCHEME_FLAG[scheme], img.data_ptr(),
                                            _stream(weight)))
    return img


def linear_split(x, weight, bias, act=None, image=None, layout=0, scheme="bf16x3", out_image=False):
    """nn.Linear (+ exact GELU) with split-precision products, fp32 accumulation: scheme 'bf16x3_images' / 'bf16x3' - fp32 operands
    as three bf16 slices each, six bf16 MFMA products per pair (on the image pipeline: both operands as X images; on the
    round-3 kernels: A split in registers); 'f16x2' - two fp16 slices of the activation, three of the scaled weight, three
    fp16 MFMA products.  ``image`` = split_weights(weight, layout, scheme) to reuse a cached image; ``layout`` picks the
    round-3 kernel: 0 the 32x32x16-MFMA tiles (mlp.fc1), 2 the 16x16x32-MFMA tiles (attn.qkv).  ``out_image`` (image schemes):
    return the image of the output (what a producer hands its consumer) instead of fp32 rows."""
    lib = _lib.load()
    K = x.shape[-1]
    N = weight.shape[0]
    _need(tuple(weight.shape) == (N, K) and bias.numel() == N, f"linear: x [..,{K}] needs weight [N,{K}] and bias [N]")
    img = split_weights(weight, layout, scheme) if image is None else image
    _need(img.numel() == lib.pafuse_split_image_bytes(N, K, layout | SCHEME_FLAG[scheme]), "linear_split: image size does not match the weight")
    x2 = x.contiguous().view(-1, K)
    out = torch.empty(x2.shape[0], N, device=x.device, dtype=torch.float32)
    if scheme == "f16x2":      # both operands as H images (pafuse_linear_h)
        _need(N % 128 == 0 or N % 224 == 0, "linear_split: f16x2 serves N that is a multiple of 128 or 224")
        ah = hsplit_rows(x2)
        oh = torch.empty(x2.shape[0] * N * 4, dtype=torch.uint8, device=x.device) if out_image else None
        with torch.cuda.device(x.device):
            _lib.check(lib.pafuse_linear_h(ah.data_ptr(), img.data_ptr(), _ptr(bias, "bias"), None if out_image else out.data_ptr(),
                                           oh.data_ptr() if out_image else None, x2.shape[0], N, K,
                                           1 if act == "gelu" else 0, _stream(x)))
        return oh if out_image else out.view(*x.shape[:-1], N)
    if scheme == "bf16x3_images":     # both operands as X images (pafuse_linear_x)
        _need(N % 128 == 0 or N % 224 == 0 or N % 96 == 0, "linear_split: the image pipeline serves N that is a multiple of 128, 224 or 96")
        ax = xsplit_rows(x2)
        ox = torch.empty(x2.shape[0] * N * 6, dtype=torch.uint8, device=x.device) if out_image else None
        with torch.cuda.device(x.device):
            _lib.check(lib.pafuse_linear_x(ax.data_ptr(), img.data_ptr(), _ptr(bias, "bias"), None if out_image else out.data_ptr(),
                                           ox.data_ptr() if out_image else None, x2.shape[0], N, K,
                                           1 if act == "gelu" else 0, _stream(x)))
        return ox if out_image else out.view(*x.shape[:-1], N)
    _need(not out_image, "linear_split: the round-3 kernels write fp32 rows")
    with torch.cuda.device(x.device):
        _lib.check(lib.pafuse_linear_split(_ptr(x2, "x"), img.data_ptr(), _ptr(bias, "bias"), out.data_ptr(),
                                           x2.shape[0], N, K,
                                           (1 if act == "gelu" else 0) + (2 if layout == 2 else 0), _stream(x)))
    return out.view(*x.shape[:-1], N)
