# bench A/B over engine options on one box; usage on the GPU box: bash tools/bq.sh
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for o in "batch_rows=12288" "batch_rows=6400" "batch_rows=19200" "batch_rows=4800" "batch_rows=12288 --opt viterbi_lb_chunk=32" "batch_rows=12288 --opt viterbi_lb_chunk=64" "batch_rows=12288 --utts 48" "batch_rows=12288 --utts 64"; do
python bench.py --no-cpu-baseline --no-greedy --no-variants --opt $o 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', round(r['value']), round(r['ms_per_step'],3), round(r['host_to_host']['value']), {k: round(v,2) for k,v in r['stages_ms_per_step'].items() if 'viterbi' in k or 'join' in k or 'filter' in k})"
done
