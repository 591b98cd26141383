#!/bin/bash
# tools/clean_loop.sh FIRST COUNT: the whole GPU suite COUNT times in a row, one log per run under gpurun_out/r3/clean/
# (the harness names the running test on stderr: a process abort reads "[snk-test] <nodeid>" + the runtime's message).
first=$1; count=$2
mkdir -p gpurun_out/r3/clean
for i in $(seq $first $((first + count - 1))); do
    log=gpurun_out/r3/clean/run_$(printf %02d $i).log
    timeout 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=6 > $log 2>&1
    rc=$?
    echo "run $i rc=$rc $(grep -a ' passed\| failed\| error' $log | tail -1)" | tee -a gpurun_out/r3/clean/summary_$first.txt
done
