"""Outline of one kernel in a hipcc -S listing: runs of MFMAs, barriers, branches, scratch (spill) traffic and LDS-DMA,
in program order - enough to see whether spills sit inside the K loop or in the epilogue.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S file.hip -o file.s
    python tools/asm_outline.py file.s <substring of the mangled kernel name>
"""
import re
import sys


def outline(path, needle):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and needle in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end") or lines[i].strip() == "s_endpgm")
    runs, last, cnt = [], None, 0
    for l in lines[start:end + 1]:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            key = "mfma"
        elif op.startswith("scratch_"):
            key = "SPILL_" + ("st" if "store" in op else "ld")
        elif op in ("s_barrier",) or op.startswith("s_cbranch") or op.startswith("global_load_lds") or op == "s_endpgm":
            key = op
        elif re.match(r"^\.LBB\d+_\d+:", t):
            key = t.split(":")[0]
        elif op.startswith("global_store") or op.startswith("global_load"):
            key = op.split("_dword")[0]
        else:
            continue
        if key == last:
            cnt += 1
        else:
            if last:
                runs.append(f"{last}x{cnt}" if cnt > 1 else last)
            last, cnt = key, 1
    runs.append(f"{last}x{cnt}" if cnt > 1 else last)
    return runs


if __name__ == "__main__":
    print(" ".join(outline(sys.argv[1], sys.argv[2])))
