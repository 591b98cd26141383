# round-3 profile set; usage (on the GPU box): bash tools/prof_round3.sh r03a
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
T=gpurun_out/$1
mkdir -p $T
python bench.py > $T/bench.json 2> $T/bench.err
tail -c 300 $T/bench.err
# the same command (workload, steps, timed region) under the kernel trace; the CPU baseline and the greedy extras are other processes' / other voices' work
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats -- python3 bench.py --no-cpu-baseline --no-greedy > $T/stats_bench.json 2>/dev/null
# counters: separate passes, two B* batch steps through the batch entry point (tools/prof_knn.py)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/fetch -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/write -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $T/mfma -- python3 tools/prof_knn.py > /dev/null 2>&1
# the K-NN of one group alone (nothing else on the GPU): stage times and the matrix pipe of the filter kernels
python3 tools/knn_time.py 9600 1 16 > $T/knn_alone.log 2>&1; tail -1 $T/knn_alone.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $T/mfma_alone -- python3 tools/knn_time.py 9600 1 16 > /dev/null 2>&1
python3 tools/single_time.py 600 > $T/single.log 2>&1; grep -a "chunk 48 warm 16\|waves 4\|mode 0" $T/single.log | cut -c1-420
find $T -name "*.csv" | wc -l
head -c 600 $T/bench.json
