# rocprofv3 passes over the hoisted greedy search (tools/prof_hoist.py: B3 shape, N = 1.5 M, me = 6, 100 steps per launch); run through gpurun
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/ph
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ph/stats -- python3 tools/prof_hoist.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ph/fetch -- python3 tools/prof_hoist.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/ph/write -- python3 tools/prof_hoist.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/ph/a -- python3 tools/prof_hoist.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/ph/c -- python3 tools/prof_hoist.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ph/stats_b1 -- python3 tools/prof_hoist.py 65536 > /dev/null 2>&1
find gpurun_out/ph -name "*.csv" | wc -l
