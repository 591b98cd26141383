cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r02t
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02t/trace -- python3 bench.py --steps 3 --warmup 1 --no-greedy --no-cpu-baseline > gpurun_out/r02t/bench.json 2>/dev/null
find gpurun_out/r02t -name "*kernel_trace.csv" | head
