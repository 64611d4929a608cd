"""MixSTE2 - drop-in for the reference denoiser (common/mixste.py:141-298) backed by the HIP library.

The module owns exactly the reference's parameters (same names, shapes, creation order and default init, so
``torch.manual_seed(s); MixSTE2(...)`` yields the reference's weights and ``load_state_dict`` accepts its
checkpoints), but has no per-layer Python forward: ``forward`` hands raw device pointers to
``pafuse_mixste2_forward`` (include/pafuse_hip.h), which runs the fused gfx950 kernels; in train mode it is one
autograd node over ``pafuse_mixste2_train_forward`` / ``_backward`` (DropPath factors drawn here, as timm draws them).
"""
import ctypes as C
import math
from functools import partial
from operator import attrgetter

import torch
import torch.nn as nn

from . import _lib


class _AttentionParams(nn.Module):
    """qkv / proj of common/mixste.py:46-61 (parameters only)."""

    def __init__(self, dim, qkv_bias):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class _MlpParams(nn.Module):
    """fc1 / fc2 of common/mixste.py:24-35 (parameters only)."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _BlockParams(nn.Module):
    """norm1, attn, norm2, mlp of common/mixste.py:84-103 (parameters only, in the reference's creation order)."""

    def __init__(self, dim, mlp_ratio, qkv_bias, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _AttentionParams(dim, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = _MlpParams(dim, int(dim * mlp_ratio))


def sinusoid_frequencies(dim):
    """omega table of SinusoidalPositionEmbeddings (common/mixste.py:134-136), evaluated by the same CPU ops."""
    half = dim // 2
    step = math.log(10000) / (half - 1)
    return torch.exp(torch.arange(half) * -step)


def _ptr(t, what):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise _lib.PafuseError(f"{what}: expected a contiguous fp32 tensor on the HIP device, got "
                               f"{t.dtype} contiguous={t.is_contiguous()} device={t.device}")
    return t.data_ptr()


class MixSTE2(nn.Module):
    def __init__(self, num_frame=9, num_joints=17, in_chans=5, embed_dim_ratio=32, depth=4,
                 num_heads=8, mlp_ratio=2., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer=None, is_train=True):
        super().__init__()
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        C_ = embed_dim_ratio
        # Constructor options PAFUSE never sets (it builds qkv_bias=True, qk_scale=None, mlp_ratio=2, no dropout,
        # common/diffusionpose.py:144-147) are inference options here: the training kernels implement the PAFUSE
        # configuration and refuse anything else in forward().
        self.qk_scale = None if qk_scale is None else float(qk_scale)     # common/mixste.py:52: qk_scale or d ** -0.5
        self.mlp_hidden = int(C_ * mlp_ratio)                             # common/mixste.py:101
        self.qkv_bias = bool(qkv_bias)
        self.drop_rate, self.attn_drop_rate = float(drop_rate), float(attn_drop_rate)   # nn.Dropout: identity in eval
        if self.qk_scale is not None and not self.qk_scale > 0:
            # `qk_scale or default` treats 0 as unset in the reference; a negative scale has no kernel here
            self.qk_scale = None if self.qk_scale == 0 else self.qk_scale
            if self.qk_scale is not None:
                raise ValueError("qk_scale must be positive")
        if self.mlp_hidden % 32 or not 0 < self.mlp_hidden <= 3 * C_:
            raise NotImplementedError(f"mlp hidden width int({C_} * {mlp_ratio}) = {self.mlp_hidden}: the kernels need a "
                                      f"multiple of 32, at most 3 * embed_dim_ratio")
        self.is_train = is_train
        self.num_frame, self.num_joints, self.in_chans = num_frame, num_joints, in_chans
        self.embed_dim, self.block_depth, self.num_heads = C_, depth, num_heads
        self.drop_path_rate = drop_path_rate

        self.Spatial_patch_to_embedding = nn.Linear(in_chans, C_)
        self.Spatial_pos_embed = nn.Parameter(torch.zeros(1, num_joints, C_))
        self.Temporal_pos_embed = nn.Parameter(torch.zeros(1, num_frame, C_))
        # index 0 stands for the parameter-free sinusoid, 2 for GELU: keys time_mlp.1.* / time_mlp.3.*
        self.time_mlp = nn.Sequential(nn.Identity(), nn.Linear(C_, C_ * 2), nn.GELU(), nn.Linear(C_ * 2, C_))
        self.STEblocks = nn.ModuleList([_BlockParams(C_, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.TTEblocks = nn.ModuleList([_BlockParams(C_, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.Spatial_norm = norm_layer(C_)
        self.Temporal_norm = norm_layer(C_)
        self.head = nn.Sequential(nn.LayerNorm(C_), nn.Linear(C_, 3))
        self.register_buffer("_freqs", sinusoid_frequencies(C_), persistent=False)
        if not self.qkv_bias:        # nn.Linear(bias=False): no key in the state dict; the kernels add these zeros
            self.register_buffer("_zero_qkv_bias", torch.zeros(3 * C_), persistent=False)
        self._wcache_by_device = {}    # device index -> (key, struct, split images, event, stream): shared with DataParallel replicas
        self.split_generation = {}     # device index -> how many times the split images were (re)made (graph caches key on it)
        self._param_names = tuple(n for n, _ in self.named_parameters())
        self.drop_fn = None            # tests: callable(block, branch, nseq, rate) -> DropPath factors [nseq] or None
        self.operand_bf16 = 0          # matrix-product mode of the linear layers (include/pafuse_hip.h): 0 fp32 MFMA, 2 split precision
        #                                "bf16x3" (the D3DP default), 4 the same products on the image pipeline ('bf16x3_images',
        #                                inference; where its kernels do not exist - training, the single-model variant -
        #                                effective_mode() runs mode 2), 3 split precision "f16x2" (inference only), 1 opt-in bf16 operands
        self.fold_layernorm = None     # split-precision inference: norm1 / norm2 applied inside the qkv / fc1 GEMMs
        #                                (pafuse_block_weights.qkv_lt ...: g-scaled weight images + a vector per layer) - the producing
        #                                whole-row kernel stores the residual stream CENTRED on its row means and emits (mean, rstd), the
        #                                consumer's epilogue is rstd acc + lt: nothing cancels whatever the row means are (round 5 in mode
        #                                2; the image pipelines since round 4).  None = on; False = the whole-row kernels write the
        #                                normalised rows (same function, one more [M,C] store and normalise pass per whole-row launch)
        self.fuse_qkv_attention = None   # split-precision inference: qkv projection + attention of a block in ONE kernel where the
        #                                sequence length has a fused form (include/pafuse_hip.h pafuse_block_weights.qkv_hs): q, k, v
        #                                never reach memory.  None = on in 'f16x2' (+7 % on the loop), always on in 'bf16x3_images' (its
        #                                only form), off in 'bf16x3' (mode 2: equal in time)
        self.fuse_mlp = None           # 'f16x2' with the LayerNorm folded and the residual stream as its image only: fc1 -> GELU -> fc2
        #                                of a block in ONE kernel, the hidden activations [M, 2C] stay in registers (include/pafuse_hip.h
        #                                pafuse_block_weights.fc2_hp).  None = on at the widths where it wins in the loop (224 and 256: the face and
        #                                the hands; equal or slower as a launch of its own, faster beside the other parts' kernels because it moves
        #                                8 C bytes per token less); True = wherever the kernel exists (also 384, where its 192 accumulator registers
        #                                leave one workgroup per CU: slower), mlp_ratio 2 only
        self.keep_f32_residual = False  # 'f16x2' with the LayerNorm folded: False keeps the residual stream between the blocks in memory
        #                                only as the two-fp16-slice image the GEMMs read (22-23 significant bits; measured: the loop stays
        #                                closer to an fp64 evaluation than the reference's fp32 arithmetic, tests/test_hip_parity.py);
        #                                True also keeps the fp32 rows (include/pafuse_hip.h pafuse_mixste2_weights.keep_f32_residual)
        self.use_side_stream = False   # training backward: weight-gradient GEMMs on a second stream (identical
        #                                results; measured 3 % slower than one stream per part at B=37, so off)
        self._side_by_device = {}

    # ------------------------------------------------------------------------------------------- C structs
    def effective_mode(self, training=False):
        """The matrix-product mode the library is handed: the requested one, except that 'bf16x3_images' on the image pipeline (4)
        falls back to the same products on the round-3 kernels (2) where the image pipeline has no kernels - training and
        shapes outside its set (pafuse_mode_supported)."""
        mode = int(self.operand_bf16)
        if mode == 4 and (training or not _lib.load().pafuse_mode_supported(4, self.embed_dim, self.mlp_hidden, self.num_heads,
                                                                           self.num_joints, self.num_frame)):
            return 2
        return mode

    def weights_struct(self, images=True):
        """pafuse_mixste2_weights pointing at the live parameter storage (cached until a pointer changes).
        images=False (training): no pre-split / folded images - the training entry points split the weight a GEMM is
        about to read themselves (weights change every step)."""
        # attribute access, not named_parameters(): nn.DataParallel replicas keep their copies as plain attributes
        def get(name):
            if name.endswith("attn.qkv.bias") and not self.qkv_bias:
                return self._zero_qkv_bias
            return attrgetter(name)(self)
        mode = self.effective_mode(training=not images)
        key = tuple(get(n).data_ptr() for n in self._param_names) + (self._freqs.data_ptr(), mode, bool(images),
                                                                     bool(self.keep_f32_residual), self.fuse_mlp)
        if not images:
            hit = self._wcache_by_device.get(("train", self._freqs.device.index))
            if hit is not None and hit[0] == key:
                return hit[1]
            w = _lib.MixSTE2Weights()
            fill_weights_struct(w, get, self._freqs, self.num_frame, self.num_joints, self.embed_dim,
                                self.block_depth, self.num_heads, self.in_chans, mode, None)
            w.mlp_hidden = self.mlp_hidden
            w.qk_scale = 0.0 if self.qk_scale is None else self.qk_scale
            self._wcache_by_device[("train", self._freqs.device.index)] = (key, w)
            return w
        split = mode in (2, 3, 4)
        fold = split and (True if self.fold_layernorm is None else bool(self.fold_layernorm))
        if split:           # the split images are values, not views: an in-place update of a weight must remake them
            key += tuple(get(n)._version for n in self._param_names if n.endswith(SPLIT_SUFFIXES))
        # (mode 4 has no unfused attention: qkv + attention is one kernel in every block)
        fuse = split and (mode == 4 or (mode == 3 if self.fuse_qkv_attention is None else bool(self.fuse_qkv_attention)))
        if fold or fuse:    # ... and so are the folded / head-major images and vectors: they also hold LayerNorm and bias values
            key += ("fold", fold, fuse) + tuple(get(n)._version for n in self._param_names if n.endswith(FOLD_SUFFIXES))
        dev = self._freqs.device
        # per device, in a dict the replicas of nn.DataParallel share with their parent (replicate() copies attributes
        # shallowly and makes fresh module objects on every forward: a per-object cache would never hit there, and every
        # replica would re-split all its weights on every call)
        # A replica made by nn.DataParallel never reads the cache: its parameters are fresh broadcast copies on every forward
        # (version 0, and the caching allocator hands out the same addresses again), so pointer + version cannot tell the
        # copies of two different parent states apart - the images (value copies) are remade per forward there, and kept
        # alive on the replica object itself.
        replica = bool(getattr(self, "_is_replica", False))
        hit = None if replica else self._wcache_by_device.get(dev.index)
        if hit is not None and hit[0] == key:
            if split and hit[3] is not None and torch.cuda.current_stream(dev) != hit[4]:
                torch.cuda.current_stream(dev).wait_event(hit[3])     # images were made on another stream
            return hit[1]
        w = _lib.MixSTE2Weights()
        images, event, stream = None, None, None
        if split:
            fuse_mlp = (mode == 3 and fold and not self.keep_f32_residual and self.mlp_hidden == 2 * self.embed_dim
                        and self.embed_dim in FUSED_MLP_WIDTHS
                        and (self.embed_dim in FUSED_MLP_DEFAULT_WIDTHS if self.fuse_mlp is None else bool(self.fuse_mlp)))
            images = self._split_images(get, fold, fuse, f16=(mode == 3), fuse_mlp=fuse_mlp, x=(mode == 4))
            stream = torch.cuda.current_stream(dev)
            event = torch.cuda.Event()
            event.record(stream)
            self.split_generation[dev.index] = self.split_generation.get(dev.index, 0) + 1
        fill_weights_struct(w, get, self._freqs, self.num_frame, self.num_joints, self.embed_dim,
                            self.block_depth, self.num_heads, self.in_chans, mode, images)
        w.mlp_hidden = self.mlp_hidden
        w.qk_scale = 0.0 if self.qk_scale is None else self.qk_scale
        w.keep_f32_residual = int(bool(self.keep_f32_residual))
        if replica:
            self._replica_keep = (w, images, event)      # alive as long as the replica (one forward)
        else:
            self._wcache_by_device[dev.index] = (key, w, images, event, stream)   # (images: keeps the storage the struct points into alive)
        return w

    def _split_images(self, get, fold=False, fuse=False, f16=False, fuse_mlp=False, x=False):
        """Pre-split images (bf16x3; f16x2 with `f16`; X images - bf16x3 on the image pipeline - with `x`) of every linear weight,
        made on the device by pafuse_split_weights: one uint8 tensor per weight, kept until a weight changes (the cache key of
        weights_struct).  `fold`: qkv / fc1 get the image of W (.) g and the two vectors of the folded LayerNorm (norm1 / norm2
        of their block, include/pafuse_hip.h pafuse_block_weights.qkv_ls): ls = W g, lt = W beta + b, formed in fp64."""
        images = {}
        for name in self._param_names:
            if not name.endswith(SPLIT_SUFFIXES):
                continue
            stem = name[:-len("weight")]
            # mode 4 multiplies the head-major qkv image only (qkv + attention fused in every block): no plain qkv image
            plain = not (x and name.endswith("attn.qkv.weight"))
            if fold and name.endswith(tuple(FOLDED_LINEAR)):
                images[name], images[stem + "ls"], images[stem + "lt"] = folded_linear(get, name, f16, x, image=plain)
            elif plain:
                images[name] = split_image(get(name), image_layout(name), f16, x)
            if fuse and name.endswith("attn.qkv.weight"):
                images[stem + "hs"], images[stem + "hb"], images[stem + "hl"] = head_major_qkv(get, name, self.num_heads, fold, f16, x)
            if fuse_mlp and name.endswith("mlp.fc2.weight"):
                images[stem + "hp"] = fused_mlp_fc2_image(get(name))
        return images

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, x_2d, x_3d, t):
        """eval: x_2d [B,F,J,2], x_3d [B,P,F,J,3], t [B] int64 -> [B,P,F,J,3]  (common/mixste.py:278-298);
        train: x_3d [B,F,J,3] -> [B,F,J,3], differentiable w.r.t. the parameters (common/mixste.py:215-225)."""
        lib = _lib.load()
        if not x_3d.is_cuda:
            raise _lib.PafuseError("MixSTE2 runs on the HIP device only (no CPU fallback)")
        if self.is_train:
            if (self.qk_scale is not None or not self.qkv_bias or self.mlp_hidden != 2 * self.embed_dim or self.drop_rate
                    or self.attn_drop_rate):
                raise NotImplementedError("training implements the PAFUSE configuration: qkv_bias=True, qk_scale=None, "
                                          "mlp_ratio=2, no dropout (common/diffusionpose.py:144-147)")
            if self.operand_bf16 in (1, 3):
                raise NotImplementedError("rounded-bf16 and f16x2 products are inference options; training runs fp32 ('f32') "
                                          "or split-precision ('bf16x3') products")
            return self._forward_train(x_2d, x_3d, t)
        B, P, F, J = self._check_inputs(x_2d, x_3d, t, 5)
        x_2d = x_2d.contiguous().float()
        x_3d = x_3d.contiguous().float()
        t = t.contiguous().long()
        w = self.weights_struct()
        out = torch.empty(B, P, F, J, 3, device=x_3d.device, dtype=torch.float32)
        nbytes = lib.pafuse_mixste2_workspace_bytes(C.byref(w), B, P)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x_3d.device)
        stream = torch.cuda.current_stream(x_3d.device).cuda_stream
        _lib.check(lib.pafuse_mixste2_forward(C.byref(w), x_2d.data_ptr(), x_3d.data_ptr(), t.data_ptr(), B, P,
                                              out.data_ptr(), ws.data_ptr(), nbytes, stream))
        return out

    def _check_inputs(self, x_2d, x_3d, t, ndim):
        """The kernels index by (B, P, F, J): refuse anything else on the host instead of faulting on the device."""
        if x_3d.dim() != ndim or x_3d.shape[-1] != 3:
            raise ValueError(f"x_3d must be [B,{'P,' if ndim == 5 else ''}F,J,3], got {tuple(x_3d.shape)}")
        B, P = x_3d.shape[0], (x_3d.shape[1] if ndim == 5 else 1)
        F, J = x_3d.shape[-3], x_3d.shape[-2]
        if (F, J) != (self.num_frame, self.num_joints):
            raise ValueError(f"this MixSTE2 was built for {self.num_frame} frames x {self.num_joints} joints, got {F} x {J}")
        if tuple(x_2d.shape) != (B, F, J, 2) or tuple(t.shape) != (B,):
            raise ValueError(f"x_2d must be [{B},{F},{J},2] and t [{B}], got {tuple(x_2d.shape)} and {tuple(t.shape)}")
        if x_2d.device != x_3d.device or t.device != x_3d.device:
            raise ValueError("x_2d, x_3d and t must live on the same device")
        return B, P, F, J

    # ------------------------------------------------------------------------------------------------ training
    def side_stream(self, device):
        if not self.use_side_stream:
            return None
        if device.index not in self._side_by_device:
            self._side_by_device[device.index] = torch.cuda.Stream(device=device)
        return self._side_by_device[device.index]

    def drop_path_factors(self, B, device):
        """DropPath factors (mask / keep) of one forward, [2*depth, 2, B*max(F,J)] or None, drawn as the reference's
        blocks draw them: execution order STE0 (attn, mlp), TTE0, STE1, ...; one Bernoulli(keep) per sequence of the
        block's batch axis ((b f) spatial, (b n) temporal); rate 0 is nn.Identity and draws nothing
        (common/mixste.py:100,113-116,187; timm.layers.drop_path with scale_by_keep=True)."""
        rates = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.block_depth)]
        if not self.training or all(r <= 0.0 for r in rates):
            return None
        smax = B * max(self.num_frame, self.num_joints)
        out = torch.ones(2 * self.block_depth, 2, smax, device=device, dtype=torch.float32)
        for i, rate in enumerate(rates):
            for k, nseq in ((2 * i, B * self.num_frame), (2 * i + 1, B * self.num_joints)):
                for branch in range(2):
                    if self.drop_fn is not None:
                        m = self.drop_fn(k, branch, nseq, rate)
                    elif rate > 0.0:
                        keep = 1.0 - rate
                        m = torch.empty((nseq, 1, 1), device=device, dtype=torch.float32).bernoulli_(keep)
                        if keep > 0.0:
                            m.div_(keep)
                    else:
                        m = None
                    if m is not None:
                        out[k, branch, :nseq] = m.reshape(nseq).to(device=device, dtype=torch.float32)
        return out

    def _forward_train(self, x_2d, x_3d, t):
        B, _, F, J = self._check_inputs(x_2d, x_3d, t, 4)
        drop = self.drop_path_factors(B, x_3d.device)
        params = [attrgetter(n)(self) for n in self._param_names]
        return _TrainFunction.apply(self, x_2d.contiguous().float(), x_3d.contiguous().float(), t.contiguous().long(),
                                    drop, *params)


class _TrainFunction(torch.autograd.Function):
    """pafuse_mixste2_train_forward / _backward (include/pafuse_hip.h) as one autograd node: the forward leaves its
    activations in a library-owned layout inside one byte tensor, the backward fills one gradient per parameter."""

    @staticmethod
    def forward(ctx, module, x_2d, x_3d, t, drop, *params):
        lib = _lib.load()
        w = module.weights_struct(images=False)
        B, F, J, _ = x_3d.shape
        nbytes = lib.pafuse_mixste2_train_bytes(C.byref(w), B)
        if nbytes == 0:
            raise _lib.PafuseError("pafuse_mixste2_train_bytes: unsupported configuration")
        saved = torch.empty(nbytes, dtype=torch.uint8, device=x_3d.device)
        out = torch.empty(B, F, J, 3, device=x_3d.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(x_3d.device).cuda_stream
        _lib.check(lib.pafuse_mixste2_train_forward(C.byref(w), x_2d.data_ptr(), x_3d.data_ptr(), t.data_ptr(), B,
                                                    drop.data_ptr() if drop is not None else None, out.data_ptr(),
                                                    saved.data_ptr(), nbytes, stream))
        ctx.module, ctx.saved, ctx.drop, ctx.B, ctx.nbytes = module, saved, drop, B, nbytes
        ctx.keep = (x_2d, x_3d, t)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        module = ctx.module
        w = module.weights_struct(images=False)
        shapes = [attrgetter(n)(module).shape for n in module._param_names]
        sizes = [(s.numel() + 3) // 4 * 4 for s in shapes]               # keep every gradient 16-byte aligned
        flat = torch.zeros(sum(sizes), device=dout.device, dtype=torch.float32)    # one fill instead of 208
        grads, at = {}, 0
        for n, shape, size in zip(module._param_names, shapes, sizes):
            grads[n] = flat[at:at + shape.numel()].view(shape)
            at += size
        gw = _lib.MixSTE2Weights()
        fill_weights_struct(gw, grads.__getitem__, module._freqs, module.num_frame, module.num_joints,
                            module.embed_dim, module.block_depth, module.num_heads, module.in_chans)
        dout = dout.contiguous().float()
        stream = torch.cuda.current_stream(dout.device)
        side = module.side_stream(dout.device)
        _lib.check(lib.pafuse_mixste2_train_backward(C.byref(w), C.byref(gw), dout.data_ptr(), ctx.B,
                                                     ctx.drop.data_ptr() if ctx.drop is not None else None,
                                                     ctx.saved.data_ptr(), ctx.nbytes, stream.cuda_stream,
                                                     side.cuda_stream if side is not None else None))
        if side is not None:                     # the library joined the streams; tell the allocator about the use
            for tensor in (flat, ctx.saved, dout):
                tensor.record_stream(side)
        ctx.saved = None
        return (None, None, None, None, None) + tuple(grads[n] for n in module._param_names)


SPLIT_SUFFIXES = ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")
# the LayerNorm folded into a linear layer: linear weight suffix -> the norm of the same block in front of it
FOLDED_LINEAR = {"attn.qkv.weight": "norm1", "mlp.fc1.weight": "norm2"}
FOLD_SUFFIXES = ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "attn.qkv.bias", "mlp.fc1.bias")
BLOCK_FOLD = (("qkv_ls", "attn.qkv.ls"), ("qkv_lt", "attn.qkv.lt"), ("fc1_ls", "mlp.fc1.ls"), ("fc1_lt", "mlp.fc1.lt"),
              ("qkv_hs", "attn.qkv.hs"), ("qkv_hb", "attn.qkv.hb"), ("qkv_hl", "attn.qkv.hl"), ("fc2_hp", "mlp.fc2.hp"))
BLOCK_SPLIT = (("qkv_ws", "attn.qkv.weight"), ("proj_ws", "attn.proj.weight"), ("fc1_ws", "mlp.fc1.weight"),
               ("fc2_ws", "mlp.fc2.weight"))
BLOCK_PARAMS = (("norm1_w", "norm1.weight"), ("norm1_b", "norm1.bias"), ("qkv_w", "attn.qkv.weight"),
                ("qkv_b", "attn.qkv.bias"), ("proj_w", "attn.proj.weight"), ("proj_b", "attn.proj.bias"),
                ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"), ("fc1_w", "mlp.fc1.weight"),
                ("fc1_b", "mlp.fc1.bias"), ("fc2_w", "mlp.fc2.weight"), ("fc2_b", "mlp.fc2.bias"))
MODEL_PARAMS = (("patch_w", "Spatial_patch_to_embedding.weight"), ("patch_b", "Spatial_patch_to_embedding.bias"),
                ("pos_spatial", "Spatial_pos_embed"), ("pos_temporal", "Temporal_pos_embed"),
                ("tm1_w", "time_mlp.1.weight"), ("tm1_b", "time_mlp.1.bias"), ("tm3_w", "time_mlp.3.weight"),
                ("tm3_b", "time_mlp.3.bias"), ("snorm_w", "Spatial_norm.weight"), ("snorm_b", "Spatial_norm.bias"),
                ("tnorm_w", "Temporal_norm.weight"), ("tnorm_b", "Temporal_norm.bias"), ("hnorm_w", "head.0.weight"),
                ("hnorm_b", "head.0.bias"), ("head_w", "head.1.weight"), ("head_b", "head.1.bias"))


def fill_weights_struct(w, get, freqs, frames, joints, channels, depth, heads, in_chans, operand_bf16=0, split=None):
    """Fill a pafuse_mixste2_weights from ``get(state-dict key) -> tensor`` (keys as in common/mixste.py); ``split``
    maps the names of the linear weights to their pre-split images (mode 2 only)."""
    if depth > _lib.MAX_DEPTH:
        raise _lib.PafuseError(f"depth {depth} > {_lib.MAX_DEPTH}")
    w.frames, w.joints, w.channels, w.depth, w.heads, w.in_chans = frames, joints, channels, depth, heads, in_chans
    w.operand_bf16 = int(operand_bf16)
    for field, key in MODEL_PARAMS:
        setattr(w, field, _ptr(get(key), key))
    w.freqs = _ptr(freqs, "freqs")
    for dst, prefix in ((w.ste, "STEblocks"), (w.tte, "TTEblocks")):
        for i in range(depth):
            for field, key in BLOCK_PARAMS:
                setattr(dst[i], field, _ptr(get(f"{prefix}.{i}.{key}"), f"{prefix}.{i}.{key}"))
            for field, key in BLOCK_SPLIT:       # (mode 4 has no plain qkv image: qkv_hs is what it multiplies)
                img = split.get(f"{prefix}.{i}.{key}") if split is not None else None
                setattr(dst[i], field, img.data_ptr() if img is not None else None)
            for field, key in BLOCK_FOLD:       # present only when the images were made with the LayerNorm folded in
                vec = split.get(f"{prefix}.{i}.{key}") if split is not None else None
                setattr(dst[i], field, vec.data_ptr() if vec is not None else None)


LAYOUT_OF = {"attn.qkv.weight": 2, "attn.proj.weight": 1, "mlp.fc2.weight": 1, "mlp.fc1.weight": 2}     # (round 6: fc1 runs the 16x16x32 kernels, its image is the M16 one like qkv's)


def image_layout(name):
    """pafuse_split_weights' layout argument for a state-dict key of a linear weight (include/pafuse_hip.h)."""
    return LAYOUT_OF[name.split(".", 2)[2] if name.count(".") >= 3 else name]


SPLIT_F16X2 = 4      # include/pafuse_hip.h PAFUSE_SPLIT_F16X2: added to the layout, the image is for the f16x2 scheme
SPLIT_X = 8          # include/pafuse_hip.h PAFUSE_SPLIT_X: the X image of the bf16x3 image pipeline (mode 4)


def split_image(weight, layout, f16=False, x=False):
    """The pre-split image of one linear weight [N,K] on its device (pafuse_split_weights): a uint8 tensor of 6 bytes per
    element; `layout`: 0 mlp.fc1 / the unit op, 1 the whole-row layers (attn.proj, mlp.fc2), 2 attn.qkv (mode 2's three
    geometries; the image pipelines have one each); `f16`: the f16x2 scheme's H image (two fp16 slices of the power-of-two-scaled
    weight, 4 bytes per element + a 256-byte tail); `x`: the X image of mode 4 (three bf16 slices, [N][K/32][3][32])."""
    lib = _lib.load()
    N, K = weight.shape
    flags = int(layout) | (SPLIT_F16X2 if f16 else 0) | (SPLIT_X if x else 0)
    img = torch.empty(lib.pafuse_split_image_bytes(N, K, flags), dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.check(lib.pafuse_split_weights(_ptr(weight.detach(), "weight"), N, K, flags, img.data_ptr(),
                                            torch.cuda.current_stream(weight.device).cuda_stream))
    return img


FUSED_MLP_WIDTHS = (224, 256, 384)     # channel widths hmlp_kernel is built for (csrc/pafuse_hip.hip fused_mlp)
FUSED_MLP_DEFAULT_WIDTHS = (224, 256)  # ... and where MixSTE2.fuse_mlp = None turns it on


def fused_mlp_column_order(hidden, device=None):
    """Column order of fc2.weight in the fused MLP kernel's image (include/pafuse_hip.h pafuse_block_weights.fc2_hp): column
    16 b + 8 h + i of the image is column 16 b + 8 (i >> 2) + 4 h + (i & 3) of the weight - inside every group of 16 hidden
    units, the order in which fc1's accumulators hold them."""
    c = torch.arange(hidden, device=device)
    b, h, i = c // 16, (c // 8) % 2, c % 8
    return 16 * b + 8 * (i // 4) + 4 * h + (i % 4)


def fused_mlp_fc2_image(weight):
    """The f16x2 image of fc2.weight [C, 2C] for the fused MLP kernel: the image of W[:, fused_mlp_column_order]."""
    w = weight.detach()
    return split_image(w[:, fused_mlp_column_order(w.shape[1], w.device)].contiguous(), LAYOUT_OF["mlp.fc2.weight"], True)


def folded_linear(get, name, f16=False, x=False, image=True):
    """(image, ls, lt) of the linear layer `name` (a state-dict key ending in attn.qkv.weight / mlp.fc1.weight) with the
    LayerNorm in front of it folded in (include/pafuse_hip.h, pafuse_block_weights.qkv_ls): the split image of W (.) g
    (one fp32 rounding per element, then split exactly), lt = W beta + b formed in fp64, rounded once; ls is None (the
    centred fold of round 5 has no mean term)."""
    block, norm = name.rsplit(".", 3)[0], FOLDED_LINEAR[name.split(".", 2)[2]]
    weight = get(name).detach()
    g, beta = get(f"{block}.{norm}.weight").detach(), get(f"{block}.{norm}.bias").detach()
    bias = get(name[:-len("weight")] + "bias").detach()
    lt = (weight.double() @ beta.double() + bias.double()).float().contiguous()
    # (ls = W g, the mean term of round 3's uncentred fold, is not read by any kernel since round 5 and no longer formed: None)
    return (split_image((weight * g[None, :]).contiguous(), image_layout(name), f16, x) if image else None), None, lt


def head_major_qkv(get, name, heads, fold, f16=False, x=False):
    """(image, hb, hl) for the fused qkv + attention kernel (include/pafuse_hip.h pafuse_block_weights.qkv_hs): the qkv weight
    [3C, C] re-ordered head by head - q_h, k_h, v_h, each zero-padded from d to DP rows (DP = 32 for d <= 32, else 48) - as
    a layout-2 image; hb = the bias in that order; with the LayerNorm folded the image is that of W (.) g and hb = W beta + b
    (folded_linear's vector, re-ordered); hl is None (not read by any kernel since the centred fold of round 5)."""
    weight = get(name).detach()
    C3, C = weight.shape
    d = C3 // 3 // heads
    dp = 32 if d <= 32 else 48
    stem = name[:-len("weight")]
    bias = get(stem + "bias").detach()
    hl = None
    if fold:
        block, norm = name.rsplit(".", 3)[0], FOLDED_LINEAR[name.split(".", 2)[2]]
        g, beta = get(f"{block}.{norm}.weight").detach(), get(f"{block}.{norm}.bias").detach()
        bias = (weight.double() @ beta.double() + bias.double()).float()     # (hl = W g: not read since round 5, not formed)
        weight = weight * g[None, :]

    def reorder(t):                     # [3C, ...] -> [heads * 3 * dp, ...]
        t = t.reshape(3, heads, d, *t.shape[1:])
        out = t.new_zeros((heads, 3, dp) + tuple(t.shape[3:]))
        out[:, :, :d] = t.transpose(0, 1)
        return out.reshape(heads * 3 * dp, *t.shape[3:]).contiguous()
    image = split_image(reorder(weight), 2, f16, x)
    return image, reorder(bias), (reorder(hl) if hl is not None else None)


def fill_block_struct(dst, blk):
    dst.norm1_w, dst.norm1_b = _ptr(blk.norm1.weight, "norm1"), _ptr(blk.norm1.bias, "norm1")
    dst.qkv_w, dst.qkv_b = _ptr(blk.attn.qkv.weight, "qkv"), _ptr(blk.attn.qkv.bias, "qkv")
    dst.proj_w, dst.proj_b = _ptr(blk.attn.proj.weight, "proj"), _ptr(blk.attn.proj.bias, "proj")
    dst.norm2_w, dst.norm2_b = _ptr(blk.norm2.weight, "norm2"), _ptr(blk.norm2.bias, "norm2")
    dst.fc1_w, dst.fc1_b = _ptr(blk.mlp.fc1.weight, "fc1"), _ptr(blk.mlp.fc1.bias, "fc1")
    dst.fc2_w, dst.fc2_b = _ptr(blk.mlp.fc2.weight, "fc2"), _ptr(blk.mlp.fc2.bias, "fc2")
