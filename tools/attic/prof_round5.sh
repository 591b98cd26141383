# round-5 profile set; usage (on the GPU box): bash tools/prof_round5.sh r05a [quick]   -> tools/summarise_round5.py gpurun_out/r05a r05_a
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
T=gpurun_out/$1
mkdir -p $T
if [ "$2" = "quick" ]; then
  python bench.py --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 20 --warmup 5 > $T/bench.json 2> $T/bench.err
else
  python bench.py --steps 20 --warmup 5 > $T/bench.json 2> $T/bench.err
fi
tail -c 300 $T/bench.err
# the same command (workload, steps, timed region) under the kernel trace; the CPU baseline, the greedy extras and the variant
# databases are other processes' / other voices' work
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats -- python3 bench.py --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 20 --warmup 5 > $T/stats_bench.json 2>/dev/null
# counters: separate passes, two B* batch steps through the batch entry point (tools/prof_knn.py)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/fetch -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/write -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $T/mfma -- python3 tools/prof_knn.py > /dev/null 2>&1
# the Viterbi side of one group alone (tools/joinlb_time.py): stage times, and the bounds kernel's counters
python3 tools/joinlb_time.py > $T/joinlb_alone.log 2>&1; grep -a "join_lb_variant" $T/joinlb_alone.log | cut -c1-420
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/jfetch -- python3 tools/joinlb_time.py 1 --reps 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/jwrite -- python3 tools/joinlb_time.py 1 --reps 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $T/jmfma -- python3 tools/joinlb_time.py 1 --reps 2 > /dev/null 2>&1
# the K-NN of one group alone
python3 tools/knn_time.py 9600 > $T/knn_alone.log 2>&1; grep -a "^prefilter" $T/knn_alone.log | cut -c1-500
python3 tools/single_time.py 600 > $T/single.log 2>&1; grep -a "chunk 48 warm 16\|mode 0" $T/single.log | cut -c1-420
python3 tools/onepass_time.py 9600 > $T/onepass.log 2>&1; grep -a "two_pass" $T/onepass.log | cut -c1-300
python3 tools/minima_time.py 600 > $T/minima.log 2>&1; tail -1 $T/minima.log
find $T -name "*.csv" | wc -l
head -c 400 $T/bench.json
