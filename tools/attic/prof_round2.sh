# round-2 profile set; usage (on the GPU box): bash tools/prof_round2.sh r02a
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
T=gpurun_out/$1
mkdir -p $T
python bench.py > $T/bench.json 2> $T/bench.err
tail -c 400 $T/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-greedy > $T/stats_bench.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/fetch -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/write -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $T/mfma -- python3 tools/prof_knn.py > /dev/null 2>&1
find $T -name "*.csv" | head -30
head -c 1500 $T/bench.json
