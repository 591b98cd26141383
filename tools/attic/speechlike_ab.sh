cd "${GRAFT_REPO_ROOT:?}" || exit 1
for o in "-" "join_bounds_delay=5" "join_bounds_delay=3"; do
args=""; if [ "$o" != "-" ]; then for kv in $o; do args="$args --opt $kv"; done; fi
python bench.py --no-cpu-baseline --no-greedy --no-shapes --steps 20 --warmup 5 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=[x for x in d['noncompact'] if x['database']=='speechlike'][0]; print('$o', round(l['frames_per_s']), round(l['ms_per_step'],3), {k: round(v,2) for k,v in l['stages_ms_per_step'].items()})"
done
