# B4-shaped main workload (1.5 M units, K 200) under engine options: bash tools/b4_ab.sh "name=v name2=v" ... ("-" = defaults)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for o in "$@"; do
args=""; if [ "$o" != "-" ]; then for kv in $o; do args="$args --opt $kv"; done; fi
python bench.py --units 1500000 --candidates 200 --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 10 --warmup 3 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('$o', round(d['value']), round(d['ms_per_step'],3), {k: round(x,2) for k,x in s.items() if k.startswith('knn') or k.startswith('join_') or k.startswith('viterbi')})"
done
