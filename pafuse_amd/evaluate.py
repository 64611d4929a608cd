"""Hypothesis aggregation + the 14 accumulators of the reference's evaluate() on the device
(main_h3wb.py:327-379).

``hypothesis_errors`` is one HIP kernel over the gathered predictions (whole-body poses from parts, 2-D
reprojection, per-joint J-Best / P-Agg / J-Agg and the part-re-centred variants); the remaining means over
(b, f, j) and the P-Best argmin are a handful of reductions over small tensors.
"""
import torch

from . import _lib


def _tables(dataset, num_kps, device):
    conn = torch.zeros(num_kps, dtype=torch.int32)
    pbroot = torch.zeros(num_kps, dtype=torch.int32)
    connection = dict(dataset.parts_connection_indices)
    connection["body"] = 0                                   # common/utils.py:116
    for part, idx in dataset.parts_joint_indices.items():
        conn[idx] = connection[part]
        pbroot[idx] = dataset.root_indices[part]
    return conn.to(device), pbroot.to(device)


def evaluate_accumulators(pred_parts, gt_parts, inputs_2d, traj, cam, dataset):
    """pred_parts [B,T,P,F,J,3], gt_parts [B,F,J,3] (part-centred, metres), inputs_2d [B,F,J,2], traj [B,F,1,3],
    cam [9] or [1,9] -> dict of the 14 per-step vectors [T] the reference accumulates per batch (same names as
    oracle.d3dp_oracle.evaluate_accumulators; multiply by 1000 for mm)."""
    lib = _lib.load()
    if not pred_parts.is_cuda:
        raise _lib.PafuseError("evaluate_accumulators runs on the HIP device only (no CPU fallback)")
    dev = pred_parts.device
    if pred_parts.dim() != 6 or pred_parts.shape[-1] != 3:
        raise ValueError(f"pred_parts must be [B,T,P,F,J,3], got {tuple(pred_parts.shape)}")
    B, T, P, F, J, _ = pred_parts.shape
    # the kernel indexes every operand by (B, F, J): check before launching
    if (tuple(gt_parts.shape) != (B, F, J, 3) or tuple(inputs_2d.shape) != (B, F, J, 2) or traj.numel() != B * F * 3
            or cam.numel() < 9):
        raise ValueError(f"evaluate_accumulators: operands do not match pred {tuple(pred_parts.shape)}: gt "
                         f"{tuple(gt_parts.shape)}, 2d {tuple(inputs_2d.shape)}, traj {tuple(traj.shape)}, cam {tuple(cam.shape)}")
    if any(t.device != dev for t in (gt_parts, inputs_2d, traj)):
        raise ValueError("evaluate_accumulators: all tensors must live on the device of pred_parts")
    covered = sorted(j for idx in dataset.parts_joint_indices.values() for j in idx)
    if covered != list(range(J)):
        raise ValueError("dataset.parts_joint_indices must cover every joint exactly once")
    pred = pred_parts.contiguous().float()
    gt = gt_parts.contiguous().float()
    x2d = inputs_2d.contiguous().float()
    tr = traj.reshape(B, F, 3).contiguous().float()
    cm = cam.reshape(-1)[:9].contiguous().float().to(dev)
    conn, pbroot = _tables(dataset, J, dev)
    e3 = torch.empty(B, T, P, F, J, device=dev)
    epb = torch.empty_like(e3)
    jbest, pagg, jagg, paggpb = (torch.empty(B, T, F, J, device=dev) for _ in range(4))
    _lib.check(lib.pafuse_hypothesis_errors(
        pred.data_ptr(), gt.data_ptr(), x2d.data_ptr(), tr.data_ptr(), cm.data_ptr(), conn.data_ptr(),
        pbroot.data_ptr(), B, T, P, F, J, e3.data_ptr(), epb.data_ptr(), jbest.data_ptr(), pagg.data_ptr(),
        jagg.data_ptr(), paggpb.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))

    def per_t(x):                                            # [B,T,F,J'] -> [T]
        return x.permute(1, 0, 2, 3).reshape(T, -1).mean(dim=-1)

    out = {"j_best": per_t(jbest), "p_agg": per_t(pagg), "j_agg": per_t(jagg), "p_agg_pb": per_t(paggpb)}
    out["p_best"] = e3.permute(1, 2, 0, 3, 4).reshape(T, P, -1).mean(dim=-1).min(dim=1).values
    eh = epb.permute(1, 2, 0, 3, 4)                          # t h b f n
    per_h = eh.reshape(T, P, -1).mean(dim=-1)
    best = per_h.argmin(dim=1)
    out["p_best_pb"] = per_h.min(dim=1).values
    for part, idx in dataset.parts_joint_indices.items():
        ph = eh[..., idx].reshape(T, P, -1).mean(dim=-1)
        out["p_best_pb_" + part] = ph.gather(1, best.view(-1, 1)).squeeze(1)
        out["p_agg_pb_" + part] = per_t(paggpb[..., idx])
    return out
