#!/usr/bin/env python3
"""Writes pafuse_amd/data/h36m_cameras.json: the Human3.6M camera calibration (4 cameras: intrinsics, image size; per
subject extrinsics) that H3WB shares (reference common/h3wb_dataset.py:101-104 reads it from
common/h36m_dataset.py:20-205).  Calibration constants are data, exported once in the build container by importing the
reference; the values are written with repr() precision so float32 conversion reproduces the reference's arrays."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PAFUSE_REFERENCE", "/root/reference")


def main():
    sys.path.insert(0, REF)
    from common import h36m_dataset as h
    intr = [{k: v for k, v in cam.items() if k != "azimuth"} for cam in h.h36m_cameras_intrinsic_params]
    extr = {s: cams for s, cams in h.h36m_cameras_extrinsic_params.items()}
    out = os.path.join(ROOT, "pafuse_amd", "data", "h36m_cameras.json")
    with open(out, "w") as f:
        json.dump({"intrinsic": intr, "extrinsic": extr}, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
