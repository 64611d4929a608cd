#!/usr/bin/env python3
"""Census of the packed-fp32 VALU forms in the gfx950 kernels of the RCCL library torch loads (build container, no GPU).

Why: on MI355X a v_pk_{add,mul,fma}_f32 whose src1 operand selects the HIGH register of its pair for the LOW result (op_sel:[0,1..])
returns wrong lanes while a wave of ANOTHER kernel on the same SIMD issues v_mfma_f32_32x32x16_bf16 (profiles/r03_bf16_mfma_concurrency.md).
This library is compiled without packed-fp32 instructions, but under DistributedDataParallel RCCL's reduction kernels run on their
own stream beside the backward pass, whose 'bf16x3' GEMMs issue that MFMA.  Until round 6 a multi-rank process group therefore trained
on the fp32 matrix cores.  This script extracts the gfx950 code object from librccl.so (clang offload bundle, compressed), disassembles
it and classifies every v_pk_*_f32 by its op_sel / op_sel_hi modifiers: the failing forms are those with op_sel[1] == 1.
    python tools/rccl_packed_fp32_census.py [--out profiles/r06_rccl_packed_fp32_census.json]
"""
import argparse
import collections
import hashlib
import json
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None, help="librccl.so to inspect (default: the one bundled with torch)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    lib = args.lib
    if lib is None:
        import torch
        lib = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    sec = subprocess.run([f"{LLVM}/llvm-readelf", "-S", "-W", lib], capture_output=True, text=True, check=True).stdout
    m = re.search(r"\.hip_fatbin\s+\w+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
    off, size = int(m.group(2), 16), int(m.group(3), 16)
    counts, risky, total_bytes = collections.Counter(), [], 0
    with tempfile.TemporaryDirectory() as tmp:
        with open(lib, "rb") as f:
            f.seek(off)
            data = f.read(size)
        starts = [i.start() for i in re.finditer(b"CCOB|__CLANG_OFFLOAD_BUNDLE__", data)]
        for n, i in enumerate(starts):
            blob = os.path.join(tmp, f"b{n}.bin")
            with open(blob, "wb") as f:
                f.write(data[i:starts[n + 1] if n + 1 < len(starts) else len(data)])
            co = os.path.join(tmp, f"co{n}.o")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--input={blob}", f"--output={co}"], capture_output=True, text=True)
            os.remove(blob)
            if r.returncode or not os.path.exists(co) or not os.path.getsize(co):
                continue
            total_bytes += os.path.getsize(co)
            dis = subprocess.Popen([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            for line in dis.stdout:
                if "v_pk_" not in line or "_f32" not in line:
                    continue
                ins = line.split("//")[0].strip()
                if not re.match(r"v_pk_(add|mul|fma)_f32", ins):
                    continue
                sel = re.search(r"op_sel:\[([0-9,]+)\]", ins)
                hi = re.search(r"op_sel_hi:\[([0-9,]+)\]", ins)
                counts[f"{ins.split()[0]} op_sel:[{sel.group(1) if sel else '-'}] op_sel_hi:[{hi.group(1) if hi else '-'}]"] += 1
                if sel and len(sel.group(1).split(",")) >= 2 and sel.group(1).split(",")[1] == "1":
                    risky.append(ins)
            dis.wait()
            os.remove(co)
    res = {"library": lib, "library_sha256_first_64MiB": hashlib.sha256(open(lib, "rb").read(64 << 20)).hexdigest(),
           "gfx950_code_object_bytes": total_bytes, "packed_fp32_instructions": sum(counts.values()), "by_form": dict(counts),
           "failing_forms_src1_high_select": len(risky), "examples": risky[:8],
           "reading": "op_sel[1] == 1 (src1 takes the high register of its pair for the low result) is the form that returns wrong lanes beside "
                      "v_mfma_f32_32x32x16_bf16 waves of another queue (profiles/r03_bf16_mfma_concurrency.md); plain forms and src0 / src2 "
                      "selects never failed in the two-kernel reproducer (tools/mfma_queue_isolate.hip V9.x)"}
    print(json.dumps(res, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
