"""H3WB data plumbing on the caller side of the hot path: the npz loader with root-joint
insertion and camera normalisation, the per-camera sequence list evaluate() iterates, and the evaluation driver.

Mirrors, for a user who has ``data/train_h3wb.npz`` + ``data/task1_test_3d.npz`` and ``pafuse_model.bin``:
  Human3WBDataset            reference common/h3wb_dataset.py:14-213 (attributes D3DP and evaluate() read)
  prepare_keypoints / fetch  reference main_h3wb.py:57-119,621-648 (mm -> m, screen normalisation, per-camera lists)
  iter_sequences             reference common/generators.py:174-249 (UnchunkedGenerator_Seq.next_epoch, no augmentation)
  evaluate                   reference main_h3wb.py:194-531 (the loop around model_eval + the mm report)
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


def evaluate(model, dataset, cams, poses_3d, poses_2d, kps_left, kps_right, batch_size=1024, group=None, log=print):
    """evaluate() of main_h3wb.py:194-531 for an already loaded eval model: every video through
    harness.evaluate_sequence (flip-TTA, 27-frame clips, sharded sampling, device-side aggregation), frame-weighted
    means in millimetres per protocol and sampling step."""
    from . import harness
    total, n = None, 0
    for cam, seq_3d, seq_2d in iter_sequences(cams, poses_3d, poses_2d):
        sums, cnt = harness.evaluate_sequence(model, dataset, seq_2d[0].astype("float32"), seq_3d[0].astype("float32"),
                                              cam[0].astype("float32"), kps_left, kps_right, batch_size, group)
        total = sums if total is None else {k: total[k] + sums[k] for k in sums}
        n += cnt
    rep = harness.report(total, n)
    for k in harness.ACCUMULATORS:
        log(f"{k:>22s}: " + " ".join(f"{v:8.3f}" for v in rep[k]) + " mm")
    return rep
