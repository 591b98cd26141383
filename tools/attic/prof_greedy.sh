# PMC passes over the greedy step kernel (tools/prof_greedy.py: N = 1.5 M, me = 6, 100 steps); run through gpurun
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/pg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pg/stats -- python3 tools/prof_greedy.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pg/fetch -- python3 tools/prof_greedy.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pg/write -- python3 tools/prof_greedy.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pg/a -- python3 tools/prof_greedy.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pg/b -- python3 tools/prof_greedy.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pg/c -- python3 tools/prof_greedy.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pg/d -- python3 tools/prof_greedy.py > /dev/null 2>&1
find gpurun_out/pg -name "*.csv" | wc -l
