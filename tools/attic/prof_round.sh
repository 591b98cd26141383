cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r01d
python bench.py --steps 5 --warmup 1 > gpurun_out/r01d/bench.json 2> gpurun_out/r01d/bench.err
tail -c 600 gpurun_out/r01d/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01d/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01d/stats_bench.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01d/fetch -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01d/write -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r01d/mfma -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r01d/sq -- python3 tools/prof_knn.py > /dev/null 2>&1
find gpurun_out/r01d -name "*.csv" | head -30
cat gpurun_out/r01d/bench.json
