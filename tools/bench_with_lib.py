#!/usr/bin/env python3
"""A/B timing of another build of the library inside one gpurun call:
    python tools/bench_with_lib.py tools/bin/<variant>/libpafuse_hip.so [bench.py arguments]
loads that build (pafuse_amd._lib.load(path), an explicit argument - the library reads no environment) and runs bench.py."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pafuse_amd import _lib  # noqa: E402

lib = _lib.load(os.path.abspath(sys.argv[1]))
print("[bench_with_lib]", sys.argv[1], lib.pafuse_version().decode(), file=sys.stderr)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
