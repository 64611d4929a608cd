"""Per-kernel register / scratch / occupancy table of the HIP library (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py [file.hip] [extra hipcc flags] > profiles/rNN_kernel_resources.txt

Device-only compile of pafuse_amd/csrc/pafuse_hip.hip for gfx950 (no GPU needed); one line per kernel.  A non-zero
`scratch` column is a spill: the split-precision kernels are expected to have none.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import HIPCC_FLAGS  # noqa: E402  (the product's own flags: no packed-fp32 VALU)

SRC = os.path.join(ROOT, "pafuse_amd", "csrc", "pafuse_hip.hip")


def table(extra=(), src=SRC):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", *HIPCC_FLAGS, "--cuda-device-only", "-c", src, "-o", os.path.join(tmp, "dev.o"),
               "-Rpass-analysis=kernel-resource-usage", *extra]
        txt = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].split(" [")[0]

        def g(key):
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        nm = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        nm = nm.replace("pafuse::", "").replace("(anonymous namespace)::", "")
        nm = re.sub(r"^void ", "", nm)
        nm = re.sub(r"\(.*\)$", "", nm)
        rows.append((nm, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                     g("SGPRs"), g(r"LDS Size \[bytes/block\]")))
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    src = SRC
    if args and args[0].endswith(".hip"):   # another translation unit (tools/gemm_bench.hip ...)
        src, args = os.path.abspath(args[0]), args[1:]
    rows = table(args, src)
    print(f"{'kernel':92s} {'vgpr':>4s} {'agpr':>4s} {'scratch':>7s} {'occ':>3s} {'sgpr':>4s}")
    for r in rows:
        print(f"{r[0][:92]:92s} {r[1]:4d} {r[2]:4d} {r[3]:7d} {r[4]:3d} {r[5]:4d}")
    spilled = [r[0] for r in rows if r[3] > 0]
    print(f"# {len(rows)} kernels, {len(spilled)} with scratch")
