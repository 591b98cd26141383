# A/B of engine options on the B* bench line: bash tools/opt_ab.sh "name=v" "name=v name2=v" ...   (on the GPU box; "-" = defaults)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for i in 1 2; do
for o in "$@"; do
args=""; if [ "$o" != "-" ]; then for kv in $o; do args="$args --opt $kv"; done; fi
python bench.py --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 20 --warmup 5 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('$o', round(d['value']), round(d['ms_per_step'],3), {k: round(x,2) for k,x in s.items() if k.startswith('knn') or k.startswith('join_') or k.startswith('viterbi')})"
done; done
