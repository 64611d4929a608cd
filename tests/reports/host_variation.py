#!/usr/bin/env python3
"""How much the fp32 CPU arithmetic of the reference's path (torch CPU kernels: the oracle, bit-identical to the reference
on the build host, tests/test_oracle_golden.py) moves between two x86 hosts, on the metric's own configuration.

    python tests/reports/host_variation.py --dump FILE.npz           on each host: the oracle's timestep embeddings (three
                                                                     parts x the ten DDIM timesteps) and the trajectories
                                                                     of hypotheses (0, 7, 19) of the P=20, T=10 loop
    python tests/reports/host_variation.py --compare A.npz B.npz --out profiles/r03_host_variation.json

Golden G19 (the reference's run on the build host) differs from the HIP path by up to 1.6e-3 mm MPJPE at two of the ten
steps, while the HIP path and the oracle run on the GPU box's host agree to 3.7e-4 mm: this report shows the two HOSTS
differ from each other by that much, where (which timesteps' sin/cos), and that the three-way picture is consistent.
"""
import json
import os
import platform
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import d3dp_oracle as orc  # noqa: E402
from pafuse_amd import synthetic as gu  # noqa: E402
from tests.golden.state_template import d3dp_template  # noqa: E402

SUB = (0, 7, 19)
TIMES = (999, 899, 799, 699, 599, 499, 399, 299, 199, 99)


def cpu_name():
    return next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), platform.processor())


def dump(path):
    sd = gu.seeded_state_dict(d3dp_template(), seed=51)
    out = {"cpu": np.frombuffer(cpu_name().encode(), dtype=np.uint8), "threads": np.asarray(torch.get_num_threads())}
    for part in orc.PART_JOINTS:
        C = gu.PART_WIDTH[part]
        pre = f"pose_estimator.{part}."
        out[f"temb.{part}"] = torch.stack([orc.timestep_embedding(sd, pre, torch.tensor([t]), C)[0] for t in TIMES]).numpy()
        half = C // 2
        freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(np.log(10000.0) / (half - 1)))
        arg = torch.tensor(TIMES, dtype=torch.float32)[:, None] * freqs[None, :]
        out[f"sin.{part}"], out[f"cos.{part}"] = arg.sin().numpy(), arg.cos().numpy()
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = [n[:, list(SUB)].contiguous() for n in gu.synthetic_noises(B=1, P=160, n=10, seed=160)]
    out["traj"] = orc.ddim_sample(sd, x2d, noises, 10, gu.SYN_JOINTS_LEFT, gu.SYN_JOINTS_RIGHT, inputs_2d_flip=x2f).numpy()
    np.savez_compressed(path, **out)
    print("wrote", path, cpu_name())


def compare(a_path, b_path, out_path):
    a, b = np.load(a_path), np.load(b_path)
    doc = {"what": "the oracle (torch CPU fp32, the reference's own ATen kernels) on two hosts, same weights / inputs / noise: "
                   "BASELINE configs[2] trajectories of hypotheses (0, 7, 19), timestep embeddings and their sin / cos inputs",
           "host_a": {"cpu": bytes(a["cpu"]).decode(), "threads": int(a["threads"])},
           "host_b": {"cpu": bytes(b["cpu"]).decode(), "threads": int(b["threads"])}, "timesteps": list(TIMES)}
    d = np.abs(a["traj"] - b["traj"])
    doc["trajectory_pointwise_max_abs_m_per_step"] = [float(d[:, k].max()) for k in range(10)]
    doc["trajectory_pointwise_mean_abs_m_per_step"] = [float(d[:, k].mean()) for k in range(10)]
    for part in orc.PART_JOINTS:
        for key in ("sin", "cos", "temb"):
            x = np.abs(a[f"{key}.{part}"] - b[f"{key}.{part}"])
            doc[f"{key}_max_abs_diff_per_timestep.{part}"] = [float(v) for v in x.max(axis=1)]
            doc[f"{key}_elements_that_differ_per_timestep.{part}"] = [int(v) for v in (x > 0).sum(axis=1)]
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: v for k, v in doc.items() if "per_step" in k or k.startswith("host")}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--dump":
        dump(sys.argv[2])
    else:
        compare(sys.argv[2], sys.argv[3], sys.argv[5])
