"""Shapes of the reference D3DP state dict (636 entries), without building any model."""
import torch

from tests.golden import golden_util as gu


def mixste2_template(prefix, J, C, F=27, depth=8, in_chans=5):
    t = {}

    def add(name, *shape):
        t[prefix + name] = torch.empty(*shape, device="meta")

    add("Spatial_pos_embed", 1, J, C)
    add("Temporal_pos_embed", 1, F, C)
    add("Spatial_patch_to_embedding.weight", C, in_chans)
    add("Spatial_patch_to_embedding.bias", C)
    add("time_mlp.1.weight", 2 * C, C)
    add("time_mlp.1.bias", 2 * C)
    add("time_mlp.3.weight", C, 2 * C)
    add("time_mlp.3.bias", C)
    for kind in ("STEblocks", "TTEblocks"):
        for i in range(depth):
            b = f"{kind}.{i}."
            add(b + "norm1.weight", C), add(b + "norm1.bias", C)
            add(b + "attn.qkv.weight", 3 * C, C), add(b + "attn.qkv.bias", 3 * C)
            add(b + "attn.proj.weight", C, C), add(b + "attn.proj.bias", C)
            add(b + "norm2.weight", C), add(b + "norm2.bias", C)
            add(b + "mlp.fc1.weight", 2 * C, C), add(b + "mlp.fc1.bias", 2 * C)
            add(b + "mlp.fc2.weight", C, 2 * C), add(b + "mlp.fc2.bias", C)
    for n in ("Spatial_norm", "Temporal_norm", "head.0"):
        add(n + ".weight", C), add(n + ".bias", C)
    add("head.1.weight", 3, C)
    add("head.1.bias", 3)
    return t


def d3dp_template(timesteps=1000, depth=8):
    """name -> tensor with the right shape/dtype; fp64 schedule buffers hold their real values."""
    from oracle.d3dp_oracle import schedule_buffers
    t = dict(schedule_buffers(timesteps))
    for part, C in gu.PART_WIDTH.items():
        t.update(mixste2_template(f"pose_estimator.{part}.", len(gu.PART_JOINTS[part]), C, depth=depth))
    return t
