cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
T=gpurun_out/r02j
mkdir -p $T
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $T/a -- python3 tools/viterbi_batch_time.py > $T/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $T/b -- python3 tools/viterbi_batch_time.py > $T/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $T/c -- python3 tools/viterbi_batch_time.py > $T/c.log 2>&1
find $T -name "*counter_collection.csv"
tail -2 $T/a.log
