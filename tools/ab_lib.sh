# A/B of two builds of the library on the B* bench line (on the GPU box): build/libsnkhip_prev.so (built from another commit with
# `git archive <commit> | tar -x -C /tmp/x && make -C /tmp/x`) against the tree's
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for i in 1 2 3; do
for lib in build/libsnkhip_prev.so snickery_amd/libsnkhip.so; do
SNK_LIBRARY=$GRAFT_REPO_ROOT/$lib python bench.py --no-cpu-baseline --no-greedy --no-variants --no-shapes --steps 20 --warmup 5 --detail-out "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), d['config']['steps_in_flight'], 'resident', d['summary'].get('resident_rows_frames_per_s'), {k: round(x,2) for k,x in s.items() if k.startswith('knn') or k.startswith('join_') or k.startswith('viterbi')}, 'roofline', d['roofline'].get('frac'))"
done; done
