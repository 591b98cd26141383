cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b3prof
for cfg in "2 2" "1 1" "1 2" "2 1"; do
set -- $cfg
export SNK_GH16_T=$1 SNK_GH16_PH=$2
rocprofv3 --kernel-trace --stats -d gpurun_out/b3prof/t$1$2 --output-format csv -- python3 tools/b3_time.py > gpurun_out/b3prof/log$1$2.txt 2>&1
f=$(find gpurun_out/b3prof/t$1$2 -name "*kernel_stats.csv" | head -1)
echo "T=$1 PH=$2"; grep -i "hoist_product16" $f | cut -c1-60,150-260
grep "fast 1" gpurun_out/b3prof/log$1$2.txt | tail -1
done
