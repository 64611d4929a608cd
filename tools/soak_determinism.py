#!/usr/bin/env python3
"""Determinism soak of the product path: N flip-TTA DDIM loops at the metric's configuration (P=20, T=10, B=1) on the
same seeded inputs, every output compared bit for bit with the first.  The split-precision (bf16x3) default runs on one
stream with grouped launches; this is the check that nothing in that schedule depends on timing.

    python tools/soak_determinism.py [N=200] [precision=bf16x3] [aux_streams=2] [library.so]   ->  one JSON line

The first run (the one every other is compared with) is made with NO aux streams, so with aux streams > 0 the check is
"the multi-stream schedule returns the single-stream bits, every time".  A library path loads another build (A/B of
compile flags inside one GPU call).
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pafuse_amd import _lib  # noqa: E402

if len(sys.argv) > 4:
    _lib.load(os.path.abspath(sys.argv[4]))
from __graft_entry__ import make_model  # noqa: E402
from pafuse_amd import synthetic as gu  # noqa: E402
from pafuse_amd._lib import kernel_source_digest  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    precision = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
    aux = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    model, _ = make_model(20, 10, seed=77)
    model.precision = precision
    model.n_aux_streams = aux
    single, _ = make_model(20, 10, seed=77)
    single.precision = precision
    single.n_aux_streams = 0
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    noises = gu.synthetic_noises(B=1, P=20, n=10, seed=3)
    model.noise_fn = single.noise_fn = lambda k, shape, device: noises[k]
    x2d, x2f = x2d.cuda(), x2f.cuda()
    ref = single(x2d, None, input_2d_flip=x2f)
    import ctypes
    lanes = _lib.load().pafuse_d3dp_lanes(ctypes.byref(model.config_struct(True)), 1, 20, aux)
    bad, t0 = 0, time.time()
    for _ in range(n):
        bad += int(not torch.equal(model(x2d, None, input_2d_flip=x2f), ref))
    print(json.dumps({"what": "repeated D3DP.forward, P=20 T=10 B=1, bitwise comparison with the first run", "runs": n,
                      "precision": precision, "aux_streams": aux, "streams_in_use": lanes, "library": sys.argv[4] if len(sys.argv) > 4 else "product",
                      "runs_that_differ": bad, "seconds": round(time.time() - t0, 1),
                      "device": torch.cuda.get_device_name(0), "kernel_source_sha256": kernel_source_digest()}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
