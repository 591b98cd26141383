cd "${GRAFT_REPO_ROOT:?}" || exit 1
for o in "prefilter=1"; do
python bench.py --steps 40 --warmup 2 --no-greedy --no-cpu-baseline --opt $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$o', round(d['value']), round(d['ms_per_step'],2), 'one', round(d['one_in_flight']['ms_per_step'],2), {k:round(v,2) for k,v in d['stages_ms_per_step'].items()})"
done
