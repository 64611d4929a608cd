"""ctypes binding of libpafuse_hip.so (the C ABI declared in include/pafuse_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no CPU fallback: if the
shared object is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpafuse_hip.so")   # the in-tree build; no environment variable selects another one

MAX_DEPTH = 16
MAX_PARTS = 4
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


BLOCK_FIELDS = ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
                "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


class BlockWeights(C.Structure):
    _names = ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
              "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
              "qkv_ws", "proj_ws", "fc1_ws", "fc2_ws",
              "qkv_ls", "qkv_lt", "fc1_ls", "fc1_lt",
              "qkv_hs", "qkv_hb", "qkv_hl", "fc2_hp")
    _fields_ = [(n, C.c_void_p) for n in _names]


class MixSTE2Weights(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("frames", "joints", "channels", "depth", "heads", "in_chans",
                                         "operand_bf16", "mlp_hidden")] + [("qk_scale", C.c_float), ("keep_f32_residual", C.c_int32)] +
                [(n, C.c_void_p) for n in ("patch_w", "patch_b", "pos_spatial", "pos_temporal",
                                           "tm1_w", "tm1_b", "tm3_w", "tm3_b", "freqs",
                                           "snorm_w", "snorm_b", "tnorm_w", "tnorm_b",
                                           "hnorm_w", "hnorm_b", "head_w", "head_b")] +
                [("ste", BlockWeights * MAX_DEPTH), ("tte", BlockWeights * MAX_DEPTH)])


class D3DPConfig(C.Structure):
    _fields_ = [("num_parts", C.c_int32), ("num_kps", C.c_int32), ("frames", C.c_int32), ("flip", C.c_int32),
                ("scale", C.c_double),
                ("part", MixSTE2Weights * MAX_PARTS),
                ("part_joints", C.c_void_p * MAX_PARTS),
                ("joint_part", C.c_void_p), ("joint_local", C.c_void_p), ("flip_perm", C.c_void_p),
                ("part_by_part_launches", C.c_int32)]


class DDIMStep(C.Structure):
    _fields_ = [("time", C.c_int64), ("last", C.c_int32),
                ("sqrt_recip_acp", C.c_double), ("sqrt_recipm1_acp", C.c_double),
                ("sqrt_alpha_next", C.c_double), ("c", C.c_double), ("sigma", C.c_double)]


# every symbol include/pafuse_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "pafuse_version": (C.c_char_p, []),
    "pafuse_last_error": (C.c_char_p, []),
    "pafuse_abi_version": (C.c_int, []),
    "pafuse_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                C.c_int32, C.c_void_p]),
    "pafuse_split_weights_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "pafuse_split_image_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "pafuse_split_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pafuse_linear_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_void_p]),
    "pafuse_hsplit_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "pafuse_xsplit_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "pafuse_linear_x": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p]),
    "pafuse_linear_h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_void_p]),
    "pafuse_qkv_attention_image": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                             C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_void_p]),
    "pafuse_mlp_h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "pafuse_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                   C.c_void_p]),
    "pafuse_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                   C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "pafuse_attention_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "pafuse_linear_weight_grad_bytes": (C.c_size_t, []),
    "pafuse_linear_weight_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pafuse_block_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "pafuse_block_forward": (C.c_int, [C.POINTER(BlockWeights), C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pafuse_time_embed": (C.c_int, [C.POINTER(MixSTE2Weights), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "pafuse_mixste2_fused_blocks": (C.c_int, [C.POINTER(MixSTE2Weights)]),
    "pafuse_mode_supported": (C.c_int, [C.c_int32] * 6),
    "pafuse_mixste2_workspace_bytes": (C.c_size_t, [C.POINTER(MixSTE2Weights), C.c_int32, C.c_int32]),
    "pafuse_mixste2_forward": (C.c_int, [C.POINTER(MixSTE2Weights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pafuse_d3dp_workspace_bytes": (C.c_size_t, [C.POINTER(D3DPConfig), C.c_int32, C.c_int32]),
    "pafuse_d3dp_sample": (C.c_int, [C.POINTER(D3DPConfig), C.POINTER(DDIMStep), C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32]),
    "pafuse_d3dp_lanes": (C.c_int, [C.POINTER(D3DPConfig), C.c_int32, C.c_int32, C.c_int32]),
    "pafuse_d3dp_range_flag_offset": (C.c_size_t, [C.POINTER(D3DPConfig), C.c_int32, C.c_int32]),
    "pafuse_d3dp_check_range": (C.c_int, [C.POINTER(D3DPConfig), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pafuse_embed": (C.c_int, [C.c_void_p] * 11 + [C.c_float] + [C.c_int32] * 8 + [C.c_double, C.c_void_p, C.c_void_p,
                                                                                    C.c_void_p]),
    "pafuse_ddim_finalize": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32] + [C.c_void_p] * 6 +
                             [C.c_int32] * 7 + [C.c_double, C.POINTER(DDIMStep), C.c_void_p]),
    "pafuse_hypothesis_errors": (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 5 + [C.c_void_p] * 7),
    "pafuse_d3dp_replay_gemms": (C.c_int, [C.POINTER(D3DPConfig), C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.POINTER(C.c_double)]),
    "pafuse_d3dp_replay_layers": (C.c_int, [C.POINTER(D3DPConfig), C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "pafuse_mixste2_train_bytes": (C.c_size_t, [C.POINTER(MixSTE2Weights), C.c_int32]),
    "pafuse_mixste2_train_forward": (C.c_int, [C.POINTER(MixSTE2Weights), C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                               C.c_void_p]),
    "pafuse_mixste2_train_backward": (C.c_int, [C.POINTER(MixSTE2Weights), C.POINTER(MixSTE2Weights), C.c_void_p,
                                                C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pafuse_d3dp_qsample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                      C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
}

_lib = None

ABI_VERSION = 6      # include/pafuse_hip.h PAFUSE_ABI_VERSION: the struct layouts mirrored above

KERNEL_SOURCES = ("pafuse_amd/csrc/kernels.hpp", "pafuse_amd/csrc/hgemm.hpp", "pafuse_amd/csrc/xgemm.hpp", "pafuse_amd/csrc/sgemm.hpp", "pafuse_amd/csrc/train_kernels.hpp", "pafuse_amd/csrc/pafuse_hip.hip",
                  "pafuse_amd/csrc/train_host.inc", "include/pafuse_hip.h")


def kernel_source_digest():
    """SHA-256 over the native sources the library is built from and the compiler flags.  Measurements kept under
    profiles/ carry it, and bench.py only quotes a committed counter profile whose digest equals the running tree's (the
    GPU box has no .git)."""
    import hashlib
    from .build_flags import HIPCC_FLAGS
    root = os.path.dirname(_HERE)
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for rel in KERNEL_SOURCES:
        h.update(rel.encode())
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


class PafuseError(RuntimeError):
    pass


def load(path=None):
    """Load the shared library (once) and attach the prototypes.  Raises if it has not been built.
    `path` (development tools only, before anything else has loaded the library): another build of it, for A/B timing
    of two builds inside one process image - an explicit argument, never the environment."""
    global _lib
    if _lib is None:
        lib_path = path or LIB_PATH
        if not os.path.exists(lib_path):
            raise PafuseError(f"{lib_path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = C.CDLL(lib_path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                raise PafuseError(f"{lib_path} does not export {name}: a stale build - rebuild it (__graft_entry__.build())")
            fn.restype, fn.argtypes = res, args
        if lib.pafuse_abi_version() != ABI_VERSION:    # struct layouts carry no size fields: a mismatch would shift pointers silently
            raise PafuseError(f"{lib_path} has struct layout version {lib.pafuse_abi_version()}, these bindings {ABI_VERSION}: rebuild")
        _lib = lib
    elif path is not None and os.path.abspath(path) != os.path.abspath(getattr(_lib, "_name", "")):
        raise PafuseError("the library is already loaded from " + str(getattr(_lib, "_name", "?")))
    return _lib


def check(rc):
    if rc < 0:
        raise PafuseError(f"pafuse_hip error {rc}: {load().pafuse_last_error().decode()}")
    return rc
