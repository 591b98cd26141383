for o in "viterbi_lb_chunk=32 --opt viterbi_lb_warm=16" "viterbi_lb_chunk=32 --opt viterbi_lb_warm=32" "viterbi_lb_chunk=48 --opt viterbi_lb_warm=16" "viterbi_lb_chunk=24 --opt viterbi_lb_warm=16"; do
python bench.py --no-cpu-baseline --no-greedy --opt viterbi_lb_chunk_max_utts=24 --opt $o 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$o', round(r['value']), round(r['ms_per_step'],3), {k: round(v,2) for k,v in r['stages_ms_per_step'].items() if 'viterbi' in k or 'join' in k}, round(r['with_upload']['value']), round(r['one_in_flight']['value']))"
done
