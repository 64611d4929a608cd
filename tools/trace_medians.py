#!/usr/bin/env python3
"""Per (kernel, grid) medians of a rocprofv3 --kernel-trace run, from its results database:
    python tools/trace_medians.py gpurun_out/prof_x/x_results.db [top N]"""
import sqlite3
import statistics as st
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
d = defaultdict(list)
for n, g, wg, s, e in db.execute("select name, grid_x, workgroup_x, start, end from kernels"):
    d[(n.replace("void pafuse::", "").replace("(pafuse::GemmParams)", "").replace("(pafuse::FqaParams)", "")[:80], g // max(wg, 1))].append((e - s) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{k[0]:80s} wgs={k[1]:7d} n={len(v):5d} med={st.median(v):8.1f} us  share={sum(v) / tot:6.3f}")
