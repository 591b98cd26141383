cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/kprof
rocprofv3 --kernel-trace --stats -d gpurun_out/kprof/s --output-format csv -- python3 tools/prof_knn.py > gpurun_out/kprof/log.txt 2>&1
f=$(find gpurun_out/kprof/s -name "*kernel_stats.csv" | head -1)
grep -i "refine\|balls\|bucket\|sweep16b\|finalize" $f | cut -c1-50,200-300
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/kprof/f -- python3 tools/prof_knn.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/kprof/f/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'refine' in k or 'balls' in k: acc[k.split('(')[0][:40]].append(float(row['Counter_Value']))
    for k, v in acc.items(): print(k, 'FETCH_SIZE KB avg', sum(v)/len(v), len(v))
PY
