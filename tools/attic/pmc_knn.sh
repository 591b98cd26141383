cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r02a/mfma -- python3 tools/knn_time.py 6400 1 16 > gpurun_out/r02a/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/r02a/sq -- python3 tools/knn_time.py 6400 1 16 > gpurun_out/r02a/sq.log 2>&1
tail -2 gpurun_out/r02a/mfma.log
find gpurun_out/r02a -name "*counter_collection.csv"
