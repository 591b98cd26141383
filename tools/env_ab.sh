# A/B of an environment knob on the B* bench line: bash tools/env_ab.sh VAR v1 v2 ...   (on the GPU box)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
VAR=$1; shift
for i in 1 2; do
for v in "$@"; do
env $VAR=$v python bench.py --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('$VAR=$v', round(d['value']), round(d['ms_per_step'],3), {k: round(x,2) for k,x in s.items() if k.startswith('knn') or k.startswith('join_l')})"
done; done
