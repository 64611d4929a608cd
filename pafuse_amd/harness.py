"""Caller side of the hot path: what the reference's evaluate() does around ``model_eval(...)`` for one sequence
(main_h3wb.py:261-379) - flip copy, cutting into 27-frame clips, part centring, batched sampling, the 14
accumulators - with tensors kept on the HIP device.  SURVEY.md section 8f rows n1/n3 (partial: no dataset files
or Hydra here; the caller hands in arrays).
"""
import torch

from .evaluate import evaluate_accumulators
from .parallel import ShardedSampler

ACCUMULATORS = ("j_best", "p_best", "p_agg", "j_agg", "p_best_pb", "p_best_pb_body", "p_best_pb_face",
                "p_best_pb_left_hand", "p_best_pb_right_hand", "p_agg_pb", "p_agg_pb_body", "p_agg_pb_face",
                "p_agg_pb_left_hand", "p_agg_pb_right_hand")


def flip_2d(inputs_2d, kps_left, kps_right):
    """Test-time-augmentation copy of the 2-D input: negate x, swap left/right joints (main_h3wb.py:268-270)."""
    out = inputs_2d.clone()
    out[..., 0] *= -1
    out[..., list(kps_left) + list(kps_right), :] = out[..., list(kps_right) + list(kps_left), :]
    return out


def cut_clips(seq, frames=27):
    """[N, J, c] (or [1, N, J, c]) -> [ceil(N/frames), frames, J, c]: consecutive clips, the last one holding the LAST
    `frames` frames (so it overlaps its predecessor), sequences shorter than a clip padded by repeating the final
    frame (eval_data_prepare, main_h3wb.py:122-154)."""
    if seq.dim() == 4:
        seq = seq.squeeze(0)
    n = seq.shape[0]
    if n < frames:
        seq = torch.cat([seq, seq[-1:].expand(frames - n, *seq.shape[1:])], dim=0)
        n = frames
    full = n // frames
    clips = [seq[i * frames:(i + 1) * frames] for i in range(full - (1 if n % frames == 0 else 0))]
    clips.append(seq[-frames:])
    return torch.stack(clips)


def center_pose_parts(pose, dataset):
    """Every part translated so that its own root joint is the origin (common/utils.py:97-112)."""
    out = torch.zeros_like(pose)
    for part, idx in dataset.parts_joint_indices.items():
        root = dataset.root_indices[part]
        out[..., idx, :] = pose[..., idx, :] - pose[..., root:root + 1, :]
    return out


def wb_pose_from_parts(pose, dataset):
    """Whole-body pose from part-centred poses: every joint shifted by its part's connection joint (face -> body
    joint 1, hands -> 10 / 11, body -> 0, which pins joint 0 at the origin).  Closed form of the reference's
    in-place pass (common/utils.py:113-126) - same values bit for bit, and the input is left untouched."""
    conn = dict(dataset.parts_connection_indices)
    conn["body"] = 0
    out = torch.zeros_like(pose)
    for part, idx in dataset.parts_joint_indices.items():
        if part in conn:
            out[..., idx, :] = pose[..., idx, :] + pose[..., conn[part]:conn[part] + 1, :]
    out[..., 0, :] = 0
    return out


@torch.no_grad()
def infer_sequence(model, dataset, seq_2d, kps_left, kps_right, batch_size=2, group=None):
    """In-the-wild inference for one 2-D keypoint sequence [N,J,2] (already screen-normalised): flip copy, 27-frame
    clips, batched sampling with ``input_3d=None``, whole-body poses (in_the_wild/utils.py:322-376).
    Returns a CPU tensor [clips, T, P, F, J, 3]."""
    dev = next(model.parameters()).device
    seq_2d = torch.as_tensor(seq_2d, dtype=torch.float32)
    x2d = cut_clips(seq_2d, model.frames).to(dev)
    x2d_flip = cut_clips(flip_2d(seq_2d, kps_left, kps_right), model.frames).to(dev)
    sampler = ShardedSampler(model, group)
    outs = []
    for lo in range(0, x2d.shape[0], batch_size):
        pred = sampler(x2d[lo:lo + batch_size], None, input_2d_flip=x2d_flip[lo:lo + batch_size])
        outs.append(wb_pose_from_parts(pred, dataset).cpu())
    return torch.cat(outs)


def load_pifpaf_keypoints(path, num_kps=134):
    """OpenPifPaf whole-body detections, one JSON object per line and frame, first person of each frame
    (in_the_wild/h3wb_diffusion.py:57-70): -> float32 [frames, num_kps, 2] in pixels with the root joint (midpoint of
    the shifted hip joints 12 and 13) inserted at index 0.  Confidences are dropped, as in the reference."""
    import json
    import numpy as np
    frames = []
    with open(path) as f:
        for line in f:
            if line.strip():
                frames.append(json.loads(line))
    kps = np.zeros((len(frames), num_kps, 2), dtype=np.float32)
    for i, frame in enumerate(frames):
        flat = frame["predictions"][0]["keypoints"]
        kps[i, 1:, 0] = flat[::3]
        kps[i, 1:, 1] = flat[1::3]
        kps[i, :1, :] = (kps[i, 12:13, :] + kps[i, 13:14, :]) / 2.
    return kps


def stitch_clips(prediction, total_frames, frames=27):
    """[clips, T, P, frames, J, 3] (clips cut by :func:`cut_clips`) -> [T, P, total_frames, J, 3]: consecutive clips laid
    end to end, the last one contributing only its final `total_frames - frames * (clips - 1)` frames
    (in_the_wild/h3wb_diffusion.py:118-131).  Sequences shorter than one clip keep their first total_frames frames."""
    clips, T, P, _, J, c = prediction.shape
    out = prediction.new_empty(T, P, total_frames, J, c)
    if total_frames <= frames:
        return prediction[0, :, :, :total_frames].clone()
    for i in range(clips - 1):
        out[:, :, i * frames:(i + 1) * frames] = prediction[i]
    left = total_frames - (clips - 1) * frames
    out[:, :, -left:] = prediction[-1, :, :, -left:]
    return out


def camera_to_world(X, R, t=0.0):
    """Rotate points [..., 3] by the unit quaternion R (w, x, y, z) and translate (common/camera.py:27-28,
    common/quaternion.py:3-17: v + 2 (w (q x v) + q x (q x v)))."""
    R = torch.as_tensor(R, dtype=X.dtype, device=X.device)
    q = R[1:].expand(X.shape)
    uv = torch.cross(q, X, dim=-1)
    uuv = torch.cross(q, uv, dim=-1)
    return X + 2 * (R[0] * uv + uuv) + t


def read_checkpoint(path):
    """``torch.load`` for the reference's checkpoint files (``best_epoch.bin`` / ``pafuse_model.bin``, written by
    common/logging.py:83-115) and for the ones :func:`pafuse_amd.h3wb.save_state` writes.  Besides tensors they hold
    ``random_state``, a pickled ``numpy.random.RandomState`` (main_h3wb.py:1047), which the ``weights_only=True``
    default of torch >= 2.6 refuses - so the file is loaded with ``weights_only=False`` (only open files you trust,
    as with the reference)."""
    return torch.load(path, map_location="cpu", weights_only=False)


def load_checkpoint(model, checkpoint):
    """Accept what the reference saves (common/logging.py:83-115): a dict with 'model_pos', DataParallel
    ``module.``-prefixed keys, or a bare state dict - or the path of such a file (:func:`read_checkpoint`)."""
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        checkpoint = read_checkpoint(checkpoint)
    sd = checkpoint.get("model_pos", checkpoint) if isinstance(checkpoint, dict) else checkpoint
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    return model.load_state_dict(sd)


@torch.no_grad()
def evaluate_sequence(model, dataset, seq_2d, seq_3d, cam, kps_left, kps_right, batch_size=1024, group=None):
    """One (camera, sequence) item of evaluate()'s loop (main_h3wb.py:261-396).

    seq_2d [N,J,2] normalised 2-D input, seq_3d [N,J,3] camera-space ground truth (metres), cam [9] intrinsics.
    Returns ``(sums, n)``: the 14 accumulators [T] already multiplied by the batch multiplier (clips x frames), and
    the multiplier total - add them over sequences and report ``sums[k] / n * 1000`` mm, like :func:`report`.
    With torch.distributed initialised the hypothesis axis is sharded and gathered (pafuse_amd.parallel)."""
    dev = next(model.parameters()).device
    seq_2d = torch.as_tensor(seq_2d, dtype=torch.float32)
    seq_3d = torch.as_tensor(seq_3d, dtype=torch.float32)
    x2d = cut_clips(seq_2d, model.frames).to(dev)
    x2d_flip = cut_clips(flip_2d(seq_2d, kps_left, kps_right), model.frames).to(dev)
    gt = cut_clips(seq_3d, model.frames).to(dev)
    traj = gt[:, :, :1].clone()                                   # main_h3wb.py:300
    gt_parts = center_pose_parts(gt, dataset)                     # main_h3wb.py:304
    cam = torch.as_tensor(cam, dtype=torch.float32).to(dev)
    sampler = ShardedSampler(model, group)
    sums, n = None, 0
    for lo in range(0, x2d.shape[0], batch_size):
        hi = min(lo + batch_size, x2d.shape[0])
        pred = sampler(x2d[lo:hi], gt_parts[lo:hi], input_2d_flip=x2d_flip[lo:hi])      # [b,T,P,F,J,3]
        acc = evaluate_accumulators(pred, gt_parts[lo:hi], x2d[lo:hi], traj[lo:hi], cam, dataset)
        mult = (hi - lo) * model.frames                           # main_h3wb.py:333
        sums = {k: mult * v for k, v in acc.items()} if sums is None else {k: sums[k] + mult * acc[k] for k in acc}
        n += mult
    return sums, n


def report(sums, n):
    """mm per protocol and step, as evaluate() prints them (main_h3wb.py:415-509)."""
    return {k: (v / n * 1000.0).tolist() for k, v in sums.items()}


def format_report(rep, test_time_augmentation=True, action=None):
    """The reference's evaluation log, line for line (main_h3wb.py:406-509): returns ``(printed, written)`` - the lines
    evaluate() prints and the lines it appends to ``h36m_test_log_H<P>_K<T>.txt`` (the reference prints P_Best but
    does not write it, and writes each part-based header twice; both quirks are kept so files diff clean)."""
    printed, written = [], []

    def both(line):
        printed.append(line), written.append(line)

    if action is None:
        printed.append("----------")
    else:
        both("----" + action + "----")
    printed.append(f"Test time augmentation: {test_time_augmentation}")
    for ii in range(len(rep["j_best"])):
        hands = lambda key: (rep[key + "_right_hand"][ii] + rep[key + "_left_hand"][ii]) / 2.
        both("step %d : Protocol #1 Error (MPJPE) J_Best: %f mm" % (ii, rep["j_best"][ii]))
        printed.append("step %d : Protocol #1 Error (MPJPE) P_Best: %f mm" % (ii, rep["p_best"][ii]))
        both("step %d : Protocol #1 Error (MPJPE) P_Agg: %f mm" % (ii, rep["p_agg"][ii]))
        both("step %d : Protocol #1 Error (MPJPE) J_Agg: %f mm" % (ii, rep["j_agg"][ii]))
        for header, key, name in (("-----------------> Part-Based Evaluation <-----------------", "p_best_pb", "P_Best"),
                                  ("-----------------> Part-Based Evaluation Aggregation <-----------------", "p_agg_pb",
                                   "P_Agg")):
            both(header)
            written.append(header)
            both("step %d : Protocol #1 Error (MPJPE) %s Part-Based: %f mm" % (ii, name, rep[key][ii]))
            both("step %d : Protocol #1 Error (MPJPE) %s Part-Based BODY: %f mm" % (ii, name, rep[key + "_body"][ii]))
            both("step %d : Protocol #1 Error (MPJPE) %s Part-Based FACE: %f mm" % (ii, name, rep[key + "_face"][ii]))
            both("step %d : Protocol #1 Error (MPJPE) %s Part-Based HANDS: %f mm" % (ii, name, hands(key)))
            both("step %d : Protocol #1 Error (MPJPE) %s Part-Based LEFT HAND: %f mm" % (ii, name, rep[key + "_left_hand"][ii]))
            both("step %d : Protocol #1 Error (MPJPE) %s Part-Based RIGHT HAND: %f mm" % (ii, name, rep[key + "_right_hand"][ii]))
    both("----------")
    return printed, written
