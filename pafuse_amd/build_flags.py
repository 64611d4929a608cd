"""The hipcc flags libpafuse_hip.so is built with (read by __graft_entry__.build(), tools/ and the source digest).

The device code is compiled WITHOUT the packed-fp32 VALU instructions (target feature packed-fp32-ops off).  On MI355X a
v_pk_{add,mul,fma}_f32 whose src1 operand takes the high register of its pair for the low result (op_sel:[0,1,..])
returns wrong lanes while a wave of ANOTHER kernel on the same SIMD issues v_mfma_f32_32x32x16_bf16 - two-kernel
reproducer tools/mfma_queue_isolate.hip (victims V8, V9.3, V9.9, V9.10, V9.12), write-up
profiles/r03_bf16_mfma_concurrency.md.  hipcc's SLP vectoriser emits that form freely in VALU-only kernels, which is why
the split-precision (bf16-MFMA) kernels could not share the chip with kernels of other queues in round 2.  With the
feature off the side streams are safe; -DPAFUSE_NO_PACKED_F32 tells the library it was built that way
(lanes_allowed in csrc/pafuse_hip.hip), and tests/test_host_cabi.py disassembles the shipped library to check it.

The feature switch has no driver spelling, hence -Xclang; the host half of the compile answers "not a recognized
feature for this target" for it, which build() drops from its output.
"""
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
               "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DPAFUSE_NO_PACKED_F32=1"]
