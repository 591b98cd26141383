# kernel times of one B3 greedy search (tools/b3_time.py) under rocprofv3; usage on the GPU box: bash tools/prof_b3.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b3prof
rocprofv3 --kernel-trace --stats -d gpurun_out/b3prof/t --output-format csv -- python3 tools/b3_time.py > gpurun_out/b3prof/log.txt 2>&1
f=$(find gpurun_out/b3prof/t -name "*kernel_stats.csv" | head -1)
grep -i "hoist_product\|greedy32_kernel\|Name" $f | cut -c1-70,150-260
grep "^fast" gpurun_out/b3prof/log.txt | tail -2
