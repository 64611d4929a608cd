#!/bin/bash
# phase ablation of the opt-in fused qkv + attention kernel: the product build against one built with -DPAFUSE_FQA_ABL (no attention
# phase; tools/bin/fqa_abl1/libpafuse_hip.so), rocprofv3 --stats of the single-stream fused loop.  Body blocks: 166 us full, 145 us
# without phase 3; the two-kernel path: gemm16_kernel 126 + attn_kernel 40.
mkdir -p gpurun_out/r4k; O=gpurun_out/r4k
cd /tmp && export TMPDIR=/tmp
for v in product fqa_abl1; do
  L=$GRAFT_REPO_ROOT/tools/bin/$v/libpafuse_hip.so; [ $v = product ] && L=$GRAFT_REPO_ROOT/pafuse_amd/libpafuse_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o $v -- python3 $GRAFT_REPO_ROOT/tools/bench_with_lib.py $L --streams 0 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --fuse-qkv-attention > /dev/null 2> $GRAFT_REPO_ROOT/$O/$v.err
  echo "== $v"; grep "fqa_kernel" $GRAFT_REPO_ROOT/$O/${v}_kernel_stats.csv | cut -d, -f1-4,6 
done
