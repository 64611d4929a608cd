#!/usr/bin/env python3
"""Experiment (round 6): the three body-part denoisers on three HIP streams that each own a SHARE OF THE CUs
(hipExtStreamCreateWithCUMask), against the default three unmasked streams.  With persistent GEMM kernels a launch holds
every CU's LDS until it ends, so kernels of different parts hardly overlap any more (loop time = sum of the kernel times);
a partition gives every part its own CUs: its memory-bound attention runs beside the other parts' matrix-bound GEMMs.
    python tools/cu_mask_experiment.py [steps]
"""
import ctypes
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
from pafuse_amd import synthetic as gu  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(lo, hi, ncu=256):
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for b in range(lo, hi):
        mask[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def run(model, x2d, x2f, main, steps):
    with torch.cuda.stream(main):
        for _ in range(3):
            out = model(x2d, None, input_2d_flip=x2f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(x2d, None, input_2d_flip=x2f)
        torch.cuda.synchronize()
    return 20 * steps / (time.perf_counter() - t0), out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ge.build()
    dev = torch.device("cuda", 0)
    x2d, x2f = gu.synthetic_inputs_2d(B=1)
    x2d, x2f = x2d.to(dev), x2f.to(dev)
    res = {}
    model, _ = ge.make_model(20, 10, seed=51, device=dev)
    torch.manual_seed(1)
    res["three unmasked streams"], ref = run(model, x2d, x2f, torch.cuda.current_stream(), steps)
    for name, cuts in (("shares 93 / 90 / 73 CUs (body / face / hands: their FLOP shares)", (0, 93, 183, 256)),
                       ("shares 96 / 96 / 64 CUs", (0, 96, 192, 256)),
                       ("shares 88 / 88 / 80 CUs", (0, 88, 176, 256))):
        streams = [masked_stream(cuts[i], cuts[i + 1]) for i in range(3)]
        m2, _ = ge.make_model(20, 10, seed=51, device=dev)
        m2.aux_streams = streams[1:]
        torch.manual_seed(1)
        res[name], out = run(m2, x2d, x2f, streams[0], steps)
        res[name + " | same bits as unmasked"] = bool(torch.equal(out, ref))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
