# round-6 profile set; usage (on the GPU box): bash tools/prof_round6.sh r06a [quick]   -> tools/summarise_round6.py gpurun_out/r06a r06_a
# Everything the bench line's roofline cites comes out of ONE pass over ONE tree: the sha256 of every kernel source is taken first
# and the summariser refuses a tree that differs.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
T=gpurun_out/$1
mkdir -p $T
sha256sum snickery_amd/csrc/*.hip snickery_amd/csrc/*.h bench.py > $T/csrc.sha256
# counters: separate passes (never with the stats), two B* batch steps through the batch entry point (tools/prof_knn.py)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/fetch -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/write -- python3 tools/prof_knn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $T/mfma -- python3 tools/prof_knn.py > /dev/null 2>&1
# the Viterbi side of one group alone (tools/joinlb_time.py; snk_viterbi_batch keeps ONE group: one launch of every pass): stage
# times, and the counters of the bounds kernel and of the exact sparse costs
python3 tools/joinlb_time.py > $T/joinlb_alone.log 2>&1; grep -a "join_lb_variant" $T/joinlb_alone.log | cut -c1-420
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/jfetch -- python3 tools/joinlb_time.py 1 --reps 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/jwrite -- python3 tools/joinlb_time.py 1 --reps 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $T/jmfma -- python3 tools/joinlb_time.py 1 --reps 2 > /dev/null 2>&1
# the counter files the bench line cites, from the passes above (this tree, this box): bench.py reads them through SNK_PROFILES_DIR
python3 tools/summarise_round6.py $T x --counters-only --out $T
export SNK_PROFILES_DIR=$PWD/$T
if [ "$2" = "quick" ]; then
  python bench.py --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 20 --warmup 5 --detail-out $T/bench_detail.json > $T/bench.json 2> $T/bench.err
else
  python bench.py --steps 20 --warmup 5 --detail-out $T/bench_detail.json > $T/bench.json 2> $T/bench.err
fi
tail -c 300 $T/bench.err; tail -c 600 $T/bench.json; echo
# the same command (workload, steps, timed region) under the kernel trace; the CPU baseline, the greedy extras and the variant
# databases are other processes' / other voices' work
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats -- python3 bench.py --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 20 --warmup 5 --detail-out $T/stats_bench_detail.json > $T/stats_bench.json 2>/dev/null
if [ "$2" != "quick" ]; then
  # the same kernel on the speech-like voice (AR(1) join rows: no L2 sharing between neighbouring candidates expected)
  python3 tools/joinlb_time.py 1 --speechlike > $T/joinlb_speechlike.log 2>&1; grep -a "join_lb_variant" $T/joinlb_speechlike.log | cut -c1-420
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/sfetch -- python3 tools/joinlb_time.py 1 --reps 2 --speechlike > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/swrite -- python3 tools/joinlb_time.py 1 --reps 2 --speechlike > /dev/null 2>&1
fi
# the K-NN of one group alone
python3 tools/knn_time.py 9600 > $T/knn_alone.log 2>&1; grep -a "^prefilter" $T/knn_alone.log | cut -c1-500
if [ "$2" != "quick" ]; then
  python3 tools/single_time.py 600 > $T/single.log 2>&1; grep -a "chunk 48 warm 16\|mode 0" $T/single.log | cut -c1-420
  python3 tools/onepass_time.py 9600 > $T/onepass.log 2>&1; grep -a "two_pass" $T/onepass.log | cut -c1-300
  python3 tools/minima_time.py 600 > $T/minima.log 2>&1; tail -1 $T/minima.log
fi
find $T -name "*.csv" | wc -l
