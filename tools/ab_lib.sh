cd $GRAFT_REPO_ROOT
for i in 1 2; do
for lib in build/libsnkhip_old.so snickery_amd/libsnkhip.so; do
SNK_LIBRARY=$GRAFT_REPO_ROOT/$lib python bench.py --no-cpu-baseline --no-greedy --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms_per_step']; print('$lib', round(d['value']), round(d['ms_per_step'],3), 'filter', round(s['knn_filter'],2), 'bucket', round(s['knn_bucket'],2), 'fin', round(s['knn_finalize'],2))"
done; done
