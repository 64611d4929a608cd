"""Deterministic synthetic weights / inputs: the H3WB checkpoint and data files are not available offline, so the
benchmark, the driver entry points, the examples and the tests all run on tensors that are a pure function of a seed.

Pure torch-CPU helpers (no reference, no oracle, no GPU): the same functions run in the build container,
where tests/golden/make_golden.py feeds their output to the real reference, and on the GPU box, where the tests feed
the identical tensors (proved by SHA-256) to the HIP path and to the oracle.
"""
import hashlib
import zlib
from typing import Dict

import torch

NUM_KPS = 134
FRAMES = 27
PART_JOINTS = {"body": list(range(0, 24)), "face": list(range(24, 92)), "hands": list(range(92, 134))}
PART_WIDTH = {"body": 384, "face": 224, "hands": 256}
DATASET_PART_JOINTS = {"body": list(range(0, 24)), "face": list(range(24, 92)),
                       "left_hand": list(range(92, 113)), "right_hand": list(range(113, 134))}
ROOT_INDICES = {"body": 0, "face": 54, "left_hand": 92, "right_hand": 113}
CONNECTION_INDICES = {"face": 1, "left_hand": 10, "right_hand": 11}
# synthetic symmetric-joint lists of SURVEY.md section 8d (the real ones are in the absent H3WB npz)
SYN_JOINTS_LEFT = [j + 1 for j in (list(range(1, 16, 2)) + [17, 18, 19] + list(range(91, 112)))]
SYN_JOINTS_RIGHT = [j + 1 for j in (list(range(2, 17, 2)) + [20, 21, 22] + list(range(112, 133)))]


def _key_generator(seed: int, key: str) -> torch.Generator:
    return torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))


def seeded_tensor(key: str, shape, seed: int) -> torch.Tensor:
    """Value of one parameter as a function of (key, shape, seed) only."""
    g = _key_generator(seed, key)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    is_norm = ("norm" in key) or (".head.0." in key) or key.startswith("head.0.")
    if "pos_embed" in key:
        return torch.randn(shape, generator=g) * 0.1
    if is_norm and leaf == "weight":
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if is_norm and leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    fan_in = shape[-1]
    return torch.randn(shape, generator=g) * (0.7 / fan_in ** 0.5)


def seeded_like(template: Dict[str, torch.Tensor], seed: int, prefix: str = "") -> Dict[str, torch.Tensor]:
    return {k: seeded_tensor(prefix + k, v.shape, seed) for k, v in template.items()}


def seeded_state_dict(template: Dict[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """fp32 parameters re-drawn by key; the fp64 schedule buffers are kept as they are."""
    out = {}
    for k, v in template.items():
        out[k] = v.clone() if v.dtype == torch.float64 else seeded_tensor(k, v.shape, seed)
    return out


def sha256_of(sd: Dict[str, torch.Tensor]) -> bytes:
    """Digest of the fp32 entries.  The fp64 schedule buffers are left out on purpose: they are recomputed by
    libm-dependent CPU kernels (cos/log/sqrt) whose last bit differs between hosts."""
    h = hashlib.sha256()
    for k in sorted(sd):
        if sd[k].dtype != torch.float32:
            continue
        h.update(k.encode())
        h.update(sd[k].detach().contiguous().cpu().numpy().tobytes())
    return h.digest()


def synthetic_inputs_2d(B: int, seed: int = 1234):
    """inputs_2d ~ U(-1,1) [B,27,134,2] and its flipped copy (reference main_h3wb.py:268-270)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, FRAMES, NUM_KPS, 2, generator=g) * 2 - 1
    xf = x.clone()
    xf[:, :, :, 0] *= -1
    xf[:, :, SYN_JOINTS_LEFT + SYN_JOINTS_RIGHT, :] = xf[:, :, SYN_JOINTS_RIGHT + SYN_JOINTS_LEFT, :]
    return x, xf


def synthetic_noises(B: int, P: int, n: int, seed: int = 1):
    """n draws of [B,P,27,134,3]: the initial img and the randn_like of every DDIM update."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, P, FRAMES, NUM_KPS, 3, generator=g) for _ in range(n)]


def synthetic_target_3d(B: int, seed: int = 1235):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, FRAMES, NUM_KPS, 3, generator=g) * 0.25
