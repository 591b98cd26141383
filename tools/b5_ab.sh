# B5-shaped main workload (1.3 M units, Dt 184, Dj 151, 64 utterances of 120 frames) under engine options: bash tools/b5_ab.sh "name=v" ... ("-" = defaults)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for i in 1 2; do
for o in "$@"; do
args=""; if [ "$o" != "-" ]; then for kv in $o; do args="$args --opt $kv"; done; fi
python bench.py --units 1300000 --target-dim 184 --join-dim 151 --frames 120 --utts 64 --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 20 --warmup 5 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('$o', round(d['value']), round(d['ms_per_step'],3), {k: round(x,2) for k,x in s.items() if k.startswith('knn') or k.startswith('join_') or k.startswith('viterbi')})"
done; done
