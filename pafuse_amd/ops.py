"""Thin tensor-level wrappers over the unit entry points of the C ABI (used by the parity tests)."""
import ctypes as C

import torch

from . import _lib
from .mixste2 import _ptr, fill_block_struct


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
    _lib.check(lib.pafuse_linear(_ptr(x2, "x"), _ptr(weight, "weight"), _ptr(bias, "bias"), out.data_ptr(),
                                 x2.shape[0], N, K, (1 if act == "gelu" else 0) | (2 if bf16 else 0), _stream(x)))
    return out.view(*x.shape[:-1], N)


def layer_norm(x, weight, bias, eps):
    lib = _lib.load()
    Cc = x.shape[-1]
    _need(weight.numel() == Cc and bias.numel() == Cc, f"layer_norm: weight and bias must have {Cc} elements")
    x2 = x.contiguous().view(-1, Cc)
    out = torch.empty_like(x2)
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
    _lib.check(lib.pafuse_attention(_ptr(qkv, "qkv"), o.data_ptr(), nseq, L, Cc, heads, group,
                                    L if group_stride is None else group_stride, seq_stride, tok_stride,
                                    _stream(qkv)))
    return o


def block_forward(block_params, x, heads=8):
    """Block.forward (common/mixste.py:113-116) on [S,L,C]; ``block_params`` is a pafuse_amd.mixste2._BlockParams."""
    lib = _lib.load()
    S, L, Cc = x.shape
    _need(block_params.norm1.weight.numel() == Cc and Cc % heads == 0, f"block: parameters are not for width {Cc}")
    y = x.contiguous().clone()
    w = _lib.BlockWeights()
    fill_block_struct(w, block_params)
    nbytes = lib.pafuse_block_workspace_bytes(S * L, Cc)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.check(lib.pafuse_block_forward(C.byref(w), y.data_ptr(), S, L, Cc, heads, ws.data_ptr(), nbytes, _stream(x)))
    return y


def time_embed(model, t):
    """MixSTE2.time_mlp(t) for a pafuse_amd.MixSTE2."""
    lib = _lib.load()
    w = model.weights_struct()
    _need(t.dim() == 1, "time_embed: t must be [B]")
    t = t.contiguous().long()
    out = torch.empty(t.shape[0], model.embed_dim, device=t.device, dtype=torch.float32)
    hid = torch.empty(t.shape[0], 2 * model.embed_dim, device=t.device, dtype=torch.float32)
    _lib.check(lib.pafuse_time_embed(C.byref(w), t.data_ptr(), t.shape[0], out.data_ptr(), hid.data_ptr(), _stream(t)))
    return out
