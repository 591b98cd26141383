cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
T=gpurun_out/r04k
mkdir -p $T
rocprofv3 --kernel-trace --stats --output-format csv -d $T/s -- python3 tools/knn_time.py > $T/s.log 2>&1
grep -v amdgpu.ids $T/s.log | tail -2
python3 - <<'PY'
import glob
for f in glob.glob('gpurun_out/r04k/s/**/*kernel_stats.csv', recursive=True):
    for l in open(f).read().splitlines()[:22]:
        p = l.split('",')
        print(p[0][:70], ','.join(p[1:])[:60])
PY
