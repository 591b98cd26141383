cd "${GRAFT_REPO_ROOT:?}" || exit 1
for b in default 0 1e-4 2e-4 1e-3 2e-3; do
if [ $b = default ]; then a=""; else a="--join-beta $b"; fi
python bench.py --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 20 --warmup 5 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('beta $b', round(d['value']), round(d['ms_per_step'],3), {k: round(x,2) for k,x in s.items() if k.startswith('join_') or k.startswith('viterbi')}, d['viterbi'].get('cells_refined'), d['viterbi'].get('exact_costs_in_refinement'))"
done
