# B4-shaped main workload (1.5 M units, K 200) under an environment knob: bash tools/b4_ab.sh VAR v1 v2 ...
cd "${GRAFT_REPO_ROOT:?}" || exit 1
VAR=${1:-SNK_NONE}; shift
for v in "${@:-0}"; do
env $VAR=$v python bench.py --units 1500000 --candidates 200 --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('$VAR=$v', round(d['value']), round(d['ms_per_step'],3), {k: round(x,2) for k,x in s.items() if k.startswith('knn') or k.startswith('join_') or k.startswith('viterbi')}, d['viterbi'].get('cells_refined'), d['viterbi'].get('steps_with_refinement'))"
done
