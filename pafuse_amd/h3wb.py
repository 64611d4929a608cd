"""H3WB data plumbing on the caller side of the hot path: the npz loader with root-joint
insertion and camera normalisation, the per-camera sequence list evaluate() iterates, and the evaluation driver.

Mirrors, for a user who has ``data/train_h3wb.npz`` + ``data/task1_test_3d.npz`` and ``pafuse_model.bin``:
  Human3WBDataset            reference common/h3wb_dataset.py:14-213 (attributes D3DP and evaluate() read)
  prepare_keypoints / fetch  reference main_h3wb.py:57-119,621-648 (mm -> m, screen normalisation, per-camera lists)
  iter_sequences             reference common/generators.py:174-249 (UnchunkedGenerator_Seq.next_epoch, no augmentation)
  evaluate                   reference main_h3wb.py:194-531 (the loop around model_eval + the mm report)
  ChunkedClips               reference common/generators.py:5-172 (ChunkedGenerator_Seq: the shuffled, flip-augmented
                             27-frame training clips)
  train_epoch / save_state   reference main_h3wb.py:820-870,1018-1040, common/logging.py:83-115
Host-side numpy; nothing here touches the device except through pafuse_amd.harness.evaluate_sequence.
"""
import copy
import json
import os

import numpy as np

KPS_ORDER = ("body", "left_foot", "right_foot", "face", "left_hand", "right_hand")
CAMERA_ORDER = ("54138969", "55011271", "58860488", "60457274")
ROOT_INDICES = {"body": 0, "face": 54, "left_hand": 92, "right_hand": 113}
PARTS_CONNECTION_INDICES = {"face": 1, "left_hand": 10, "right_hand": 11}
_CAMERA_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "h36m_cameras.json")


def normalize_screen_coordinates(X, w, h):
    """[0, w] -> [-1, 1], aspect ratio kept (reference common/camera.py:7-11)."""
    assert X.shape[-1] == 2
    return X / w * 2 - [1, h / w]


class Skeleton:
    """Parent table + left/right joint lists (the part of common/skeleton.py the evaluation path reads)."""

    def __init__(self, parents, joints_left, joints_right):
        self._parents = np.array(parents)
        self._joints_left, self._joints_right = list(joints_left), list(joints_right)

    def num_joints(self):
        return len(self._parents)

    def parents(self):
        return self._parents

    def joints_left(self):
        return self._joints_left

    def joints_right(self):
        return self._joints_right


def _hand_parents(wrist, body_wrist):
    """21 hand keypoints: the wrist hangs on the body's wrist joint, five 4-joint finger chains hang on the wrist."""
    out = [body_wrist]
    for finger in range(5):
        first = wrist + 4 * finger + 1
        out += [wrist, first, first + 1, first + 2]
    return out


def skeleton_parents(n_face, root_added):
    """Parent index of every keypoint in H3WB order (body, feet, face, hands), reference h3wb_dataset.py:137-161."""
    feet = [15] * 3 + [16] * 3
    hands = _hand_parents(91, 9) + _hand_parents(112, 10)
    if root_added:
        body = [-1] * 6 + [0, 0, 6, 7, 8, 9, 0, 0, 12, 13, 14, 15]          # inserted root first; head points free
        face = [-1] * n_face
        feet, hands = [p + 1 for p in feet], [p + 1 for p in hands]
    else:
        body = [-1, 0, 0, 0, 0, 0, 0, 5, 6, 7, 8, 5, 6, 11, 12, 13, 14]
        face = [0] * n_face
    return body + feet + face + hands


def with_root_joint(arr, hips=(11, 12)):
    """[N,133,c] -> [N,134,c] (float64, like the reference's np.zeros buffer): keypoint 0 = midpoint of the hips."""
    out = np.zeros((arr.shape[0], arr.shape[1] + 1, arr.shape[2]))
    out[:, 1:, :] = arr
    out[:, :1, :] = (arr[:, hips[0]:hips[0] + 1, :] + arr[:, hips[1]:hips[1] + 1, :]) / 2.
    return out


def load_cameras():
    """Per-subject camera list as the reference builds it (h3wb_dataset.py:101-121): H36M extrinsics + the four shared
    intrinsics, centre/focal length in normalised screen units, translation in metres, 9-float 'intrinsic' vector."""
    with open(_CAMERA_FILE) as f:
        table = json.load(f)
    cameras = copy.deepcopy(table["extrinsic"])
    for cams in cameras.values():
        for i, cam in enumerate(cams):
            cam.update(table["intrinsic"][i])
            for k, v in cam.items():
                if k not in ("id", "res_w", "res_h"):
                    cam[k] = np.array(v, dtype="float32")
            cam["center"] = normalize_screen_coordinates(cam["center"], w=cam["res_w"], h=cam["res_h"]).astype("float32")
            cam["focal_length"] = cam["focal_length"] / cam["res_w"] * 2
            if "translation" in cam:
                cam["translation"] = cam["translation"] / 1000
            cam["intrinsic"] = np.concatenate((cam["focal_length"], cam["center"], cam["radial_distortion"],
                                               cam["tangential_distortion"]))
    return cameras


class Human3WBDataset:
    """H3WB as PAFUSE reads it: ``Human3WBDataset('data/train_h3wb.npz')`` (the S8 test split is read from
    ``task1_test_3d.npz`` in the same directory)."""

    def __init__(self, path, add_root=True):
        blob = np.load(path, allow_pickle=True)
        self.metadata = blob["metadata"].item()
        data = blob["train_data"].item()
        test = np.load(os.path.join(os.path.dirname(path), "task1_test_3d.npz"), allow_pickle=True)["data"].item()
        data.update(test)

        shift = 1 if add_root else 0
        both = set(self.metadata["left_side"]) & set(self.metadata["right_side"])
        joints_left = [k + shift for k in self.metadata["left_side"] if k not in both]
        joints_right = [k + shift for k in self.metadata["right_side"] if k not in both]
        self.kps_order = list(KPS_ORDER)
        self.parents = skeleton_parents(len(self.metadata["face"]), add_root)
        self.root_indices = dict(ROOT_INDICES)
        self.parts_connection_indices = dict(PARTS_CONNECTION_INDICES)
        self.num_kps = len(self.parents)
        self.keypoints_metadata = {"layout_name": "h3wb", "num_joints": self.num_kps,
                                   "keypoints_symmetry": [joints_left, joints_right]}
        self._skeleton = Skeleton(self.parents, joints_left, joints_right)
        self._fps = 50
        self._cameras = load_cameras()
        self.camera_order_id = list(CAMERA_ORDER)

        self._data, self._cameras_full_data = {}, {}
        for subject, actions in data.items():
            cams = [self.metadata[subject][c] for c in self.camera_order_id]
            self._cameras_full_data[subject] = cams
            self._data[subject] = {}
            for action, rec in actions.items():
                def arr(a):
                    return with_root_joint(a) if add_root else a
                self._data[subject][action] = {
                    "positions": arr(rec["global_3d"]).squeeze(),
                    "cameras": list(cams),
                    "positions_3d": [arr(rec[c]["camera_3d"]).squeeze() for c in self.camera_order_id],
                    "pose_2d": [arr(rec[c]["pose_2d"]).squeeze() for c in self.camera_order_id],
                }
        # part -> joint indices after the root shift; the feet belong to the body part (h3wb_dataset.py:198-213)
        shifted = {p: [k + 1 for k in self.metadata[p]] for p in KPS_ORDER}
        self.parts_joint_indices = {
            "body": [0] + shifted["body"] + shifted["left_foot"] + shifted["right_foot"],
            "face": shifted["face"], "left_hand": shifted["left_hand"], "right_hand": shifted["right_hand"]}

    def __getitem__(self, subject):
        return self._data[subject]

    def subjects(self):
        return self._data.keys()

    def fps(self):
        return self._fps

    def skeleton(self):
        return self._skeleton

    def cameras(self):
        return self._cameras

    def supports_semi_supervised(self):
        return True


def prepare_keypoints(dataset):
    """main_h3wb.py:621-648: camera-space 3-D from millimetres to metres, 2-D keypoints to normalised screen
    coordinates of their camera (both in place, as the reference does).  Returns keypoints[subject][action][cam]."""
    keypoints = {}
    for subject in dataset.subjects():
        keypoints[subject] = {}
        for action, anim in dataset[subject].items():
            if "positions" in anim:
                anim["positions_3d"] = [p / 1000. for p in anim["positions_3d"]]
            per_cam = []
            for cam_idx, kps in enumerate(anim["pose_2d"]):
                cam = dataset.cameras()[subject][cam_idx]
                kps[..., :2] = normalize_screen_coordinates(kps[..., :2], w=cam["res_w"], h=cam["res_h"])
                per_cam.append(kps)
            keypoints[subject][action] = per_cam
    return keypoints


def fetch(subjects, keypoints, dataset, stride=1, action_filter=None, parse_3d_poses=True):
    """main_h3wb.py:57-119 without the `subset` sampling: flat per-(subject, action, camera) lists of camera
    intrinsics, 3-D and 2-D sequences, optionally strided."""
    cams, poses_3d, poses_2d = [], [], []
    for subject in subjects:
        for action, per_cam in keypoints[subject].items():
            if action_filter is not None and not any(action.startswith(a) for a in action_filter):
                continue
            poses_2d += list(per_cam)
            if subject in dataset.cameras():
                subject_cams = dataset.cameras()[subject]
                assert len(subject_cams) == len(per_cam), "Camera count mismatch"
                cams += [c["intrinsic"] for c in subject_cams if "intrinsic" in c]
            if parse_3d_poses and "positions_3d" in dataset[subject][action]:
                seqs = dataset[subject][action]["positions_3d"]
                assert len(seqs) == len(per_cam), "Camera count mismatch"
                poses_3d += list(seqs)
    if stride > 1:
        poses_2d = [p[::stride] for p in poses_2d]
        poses_3d = [p[::stride] for p in poses_3d]
    return (cams or None), (poses_3d or None), poses_2d


def iter_sequences(cams, poses_3d, poses_2d):
    """One (cam [1,9], seq_3d [1,N,J,3], seq_2d [1,N,J,2]) triple per video, batch axis added, no augmentation."""
    from itertools import zip_longest
    for c, p3, p2 in zip_longest(cams or [], poses_3d or [], poses_2d):
        yield (None if c is None else np.expand_dims(c, 0), None if p3 is None else np.expand_dims(p3, 0),
               np.expand_dims(p2, 0))


def evaluate(model, dataset, cams, poses_3d, poses_2d, kps_left, kps_right, batch_size=1024, group=None, log=print,
             action=None, log_dir=None):
    """evaluate() of main_h3wb.py:194-531 for an already loaded eval model: every video through
    harness.evaluate_sequence (flip-TTA, 27-frame clips, sharded sampling, device-side aggregation), frame-weighted
    means in millimetres per protocol and sampling step, printed in the reference's log format (:406-509) and, with
    `log_dir`, appended to ``h36m_test_log_H<P>_K<T>.txt`` there exactly as the reference writes it.  `action` is
    the per-action header of the reference's run_evaluation loop (:1117-1362): call once per action's sequences."""
    from . import harness
    total, n = None, 0
    for cam, seq_3d, seq_2d in iter_sequences(cams, poses_3d, poses_2d):
        sums, cnt = harness.evaluate_sequence(model, dataset, seq_2d[0].astype("float32"), seq_3d[0].astype("float32"),
                                              cam[0].astype("float32"), kps_left, kps_right, batch_size, group)
        total = sums if total is None else {k: total[k] + sums[k] for k in sums}
        n += cnt
    rep = harness.report(total, n)
    printed, written = harness.format_report(rep, bool(model.flip), action)
    for line in printed:
        log(line)
    if log_dir is not None:
        import os
        path = os.path.join(log_dir, "h36m_test_log_H%d_K%d.txt" % (model.num_proposals, model.sampling_timesteps))
        with open(path, mode="a") as f:
            f.write("".join(line + "\n" for line in written))
    return rep


class ChunkedClips:
    """Training batches: every video cut into `chunk_length`-frame clips centred on the video (the overhang split
    between both ends and filled by repeating the edge frame), each clip optionally a second time mirrored (x negated,
    left/right keypoints swapped, camera centre/tangential x negated), all (video, start, end, flip) items shuffled
    per epoch by a RandomState(1234) and served `batch_size` clips at a time.  Same item list, same permutation call,
    same buffers' dtype (float64) as the reference's ChunkedGenerator_Seq."""

    def __init__(self, batch_size, cameras, poses_3d, poses_2d, chunk_length, shuffle=True, random_seed=1234,
                 augment=False, kps_left=None, kps_right=None, joints_left=None, joints_right=None, shard=None):
        assert poses_3d is None or len(poses_3d) == len(poses_2d)
        assert cameras is None or len(cameras) == len(poses_2d)
        items = []
        for i, seq in enumerate(poses_2d):
            n = seq.shape[0]
            assert poses_3d is None or poses_3d[i].shape[0] == n
            n_chunks = (n + chunk_length - 1) // chunk_length
            offset = (n_chunks * chunk_length - n) // 2
            bounds = np.arange(n_chunks + 1) * chunk_length - offset
            # the reference zips len(bounds) flags with len(bounds) - 1 intervals: zip stops at the shorter list
            plain = np.full(len(bounds), False, dtype=bool)
            items += zip(np.repeat(i, len(bounds)), bounds[:-1], bounds[1:], plain)
            if augment:
                items += zip(np.repeat(i, len(bounds)), bounds[:-1], bounds[1:], ~plain)
        self.pairs, self.batch_size, self.chunk_length = items, batch_size, chunk_length
        self.num_batches = (len(items) + batch_size - 1) // batch_size
        self.random = np.random.RandomState(random_seed)
        self.shuffle, self.augment = shuffle, augment
        self.cameras, self.poses_3d, self.poses_2d = cameras, poses_3d, poses_2d
        self.kps_left, self.kps_right = kps_left, kps_right
        self.joints_left, self.joints_right = joints_left, joints_right
        # shard = (rank, world): one process per GPU.  Every rank keeps the reference's item list, seed and GLOBAL
        # batches (so an epoch is one pass and model.batch_size is the global batch, as with the reference's
        # nn.DataParallel, main_h3wb.py:699-705) and serves clips rank, rank + world, ... of each batch.
        self.shard = shard
        self.last_share = None

    def batch_num(self):
        return self.num_batches

    def num_frames(self):
        return self.num_batches * self.batch_size

    def random_state(self):
        return self.random

    def set_random_state(self, random):
        self.random = random

    def _clip(self, seq, start, end):
        lo, hi = max(start, 0), min(end, seq.shape[0])
        clip = seq[lo:hi]
        if lo != start or hi != end:
            clip = np.pad(clip, ((lo - start, end - hi), (0, 0), (0, 0)), "edge")
        return clip

    def next_epoch(self):
        pairs = self.random.permutation(self.pairs) if self.shuffle else self.pairs
        for b in range(self.num_batches):
            chunk = pairs[b * self.batch_size:(b + 1) * self.batch_size]
            if self.shard is not None:
                rank, world = self.shard
                n_global, chunk = len(chunk), chunk[rank::world]
                self.last_share = (len(chunk), n_global, world)
                if len(chunk) == 0:            # fewer clips than ranks: keep the collective's call count, weight 0
                    chunk = pairs[b * self.batch_size:b * self.batch_size + 1]
            n = len(chunk)
            b2 = np.empty((n, self.chunk_length) + self.poses_2d[0].shape[-2:])
            b3 = None if self.poses_3d is None else np.empty((n, self.chunk_length) + self.poses_3d[0].shape[-2:])
            bc = None if self.cameras is None else np.empty((n, self.cameras[0].shape[-1]))
            for i, (seq_i, start, end, flip) in enumerate(chunk):
                seq_i, start, end = int(seq_i), int(start), int(end)
                b2[i] = self._clip(self.poses_2d[seq_i], start, end)
                if flip:
                    b2[i, :, :, 0] *= -1
                    b2[i, :, self.kps_left + self.kps_right] = b2[i, :, self.kps_right + self.kps_left]
                if b3 is not None:
                    b3[i] = self._clip(self.poses_3d[seq_i], start, end)
                    if flip:
                        b3[i, :, :, 0] *= -1
                        b3[i, :, self.joints_left + self.joints_right] = b3[i, :, self.joints_right + self.joints_left]
                if bc is not None:
                    bc[i] = self.cameras[seq_i]
                    if flip:
                        bc[i, 2] *= -1
                        bc[i, 7] *= -1
            yield bc, b3, b2


# per-joint loss weights of model.weighted_loss (main_h3wb.py:724-727: 18 body weights "from MixSTE", 1.0 for the rest)
WEIGHTED_LOSS_HEAD = (1, 1, 1, 1, 1, 1, 1.5, 1.5, 4, 4, 4, 4, 1, 1, 2.5, 2.5, 2.5, 2.5)


def mpjpe_loss(predicted, target, weights=None, mse_loss=False):
    """The training loss, common/loss.py:9-27 (`mpjpe` without return_joints_err): mean joint distance, optionally
    per-joint weighted and/or squared.  Plain torch on the caller's side of the boundary, as in the reference."""
    import torch
    assert predicted.shape == target.shape
    err = torch.norm(predicted - target, dim=len(target.shape) - 1)
    if weights is not None:
        assert weights.shape[0] == target.shape[-2]
        err = weights[None, None, :].to(predicted.device) * err
    return torch.mean(torch.square(err)) if mse_loss else torch.mean(err)


def train_epoch(model, optimizer, generator, dataset, device, wb_loss=False, log=None, mse_loss=False,
                weighted_loss=False):
    """One epoch of main_h3wb.py:820-870: part-centred targets, train-mode D3DP forward (HIP), mpjpe loss
    (`model.mse_loss` / `model.weighted_loss` variants included, :724-727,:859), backward (HIP), optimizer step.
    `model` may be wrapped in DistributedDataParallel: with a rank-sharded generator (``ChunkedClips(..., shard=(rank,
    world))``) every rank holds its slice of the reference's global batch and the local loss is weighted by
    n_local * world / n_global, so the all-reduced (averaged) gradient is the gradient of the global-batch mean - what
    the reference's DataParallel computes on the gathered batch.  Returns the clip-weighted mean loss (m)."""
    import torch
    from . import harness
    total, frames = 0.0, 0
    weights = None
    if weighted_loss:
        J = dataset.skeleton().num_joints() if hasattr(dataset, "skeleton") else 134
        weights = torch.tensor(list(WEIGHTED_LOSS_HEAD) + [1.0] * (J - len(WEIGHTED_LOSS_HEAD)), device=device)
    for it, (_, batch_3d, batch_2d) in enumerate(generator.next_epoch()):
        inputs_3d = torch.from_numpy(batch_3d.astype("float32")).to(device)
        inputs_2d = torch.from_numpy(batch_2d.astype("float32")).to(device)
        inputs_3d = harness.center_pose_parts(inputs_3d, dataset)
        optimizer.zero_grad()
        pred = model(inputs_2d, inputs_3d)
        target = inputs_3d
        if wb_loss:
            pred, target = harness.wb_pose_from_parts(pred, dataset), harness.wb_pose_from_parts(inputs_3d, dataset)
        loss = mpjpe_loss(pred, target, weights, mse_loss)                              # common/loss.py:9-27
        share = getattr(generator, "last_share", None)        # (clips that count here, clips of the global batch, world)
        n_local = inputs_3d.shape[0]
        scale = 1.0
        if share is not None:
            n_local, n_global, world = share
            scale = n_local * world / n_global
        (loss * scale if scale != 1.0 else loss).backward()
        optimizer.step()
        n = n_local * inputs_3d.shape[1]
        total, frames = total + n * float(loss.detach()), frames + n
        if log is not None and it % 10 == 0:
            log("%d/%d" % (it, generator.batch_num()))
    return total / max(frames, 1)


def save_state(model, optimizer, epoch_no, lr, foldername, random_state=None, tag=None, reference_layout=True):
    """The reference's checkpoint dictionary (common/logging.py:83-115): evaluate() and --resume read it back.

    The reference always saves the state dict of an ``nn.DataParallel`` wrapper (main_h3wb.py:699-705,1029), so its
    'model_pos' keys carry a ``module.`` prefix and its loaders expect one (:252 strict, :713-714).  With
    ``reference_layout`` (default) the file written here has exactly those keys whether or not `model` is wrapped
    (DataParallel / DistributedDataParallel), so checkpoints are interchangeable with the reference in both directions
    (``harness.load_checkpoint`` strips the prefix); ``reference_layout=False`` writes bare keys."""
    import torch
    module = model.module if hasattr(model, "module") else model
    sd = module.state_dict()
    if reference_layout:
        sd = type(sd)(("module." + k, v) for k, v in sd.items())
    params = {"optimizer": optimizer.state_dict(), "epoch": epoch_no, "lr": lr, "model_pos": sd}
    if random_state is not None:
        params["random_state"] = random_state
    fname = f"{foldername}/{tag or f'epoch_{epoch_no}'}.bin"
    torch.save(params, fname)
    return fname
