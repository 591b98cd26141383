# last GPU call of round 3: profile set + fuzz log + one-rank-of-eight stage times
cd "${GRAFT_REPO_ROOT:?}" || exit 1
bash tools/prof_round3.sh r03a
mkdir -p gpurun_out/r03a
SNK_FUZZ_PER_DB=16 SNK_FUZZ_SECONDS=${1:-520} timeout 700 python3 tests/fuzz_prefilter.py 6000 7 > gpurun_out/r03a/fuzz_prefilter.log 2>&1
tail -2 gpurun_out/r03a/fuzz_prefilter.log; grep -ac " : ok" gpurun_out/r03a/fuzz_prefilter.log; grep -ac "MISMATCH" gpurun_out/r03a/fuzz_prefilter.log
timeout 240 python3 tools/shard_library_time.py 8 32 > gpurun_out/r03a/shard8.log 2>&1; tail -25 gpurun_out/r03a/shard8.log
