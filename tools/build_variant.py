#!/usr/bin/env python3
"""Build a variant of the library for an A/B inside one gpurun call:
    python tools/build_variant.py <name> -DSOMETHING=1 [...]   ->  tools/bin/<name>/libpafuse_hip.so
(run it with tools/bench_with_lib.py; tools/bin/ is git-ignored and travels to the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

name, defs = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "bin", name)
os.makedirs(out, exist_ok=True)
ge.hipcc_compile(defs + ["-shared", ge.SRC], os.path.join(out, "libpafuse_hip.so"))
print(os.path.join(out, "libpafuse_hip.so"))
