set -o pipefail
R=$PWD; O=$R/gpurun_out/issue; mkdir -p $O
python3 -c "import __graft_entry__ as ge; ge.build()" || exit 1
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-secondary --no-roofline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o a -- $B > $O/a.json 2> $O/a.err || { tail -5 $O/a.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o b -- $B > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_IFETCH --kernel-trace --output-format csv -d $O -o c -- $B > $O/c.json 2> $O/c.err || { tail -5 $O/c.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $O -o d -- $B > $O/d.json 2> $O/d.err || { tail -5 $O/d.err; exit 1; }
ls $O
