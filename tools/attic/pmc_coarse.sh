# PMC passes of the two-pass filter alone (one K-NN call of 9600 rows against the B* database, nothing else on the GPU)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3c}
mkdir -p $O
python3 tools/knn_time.py 9600 1 16 > $O/alone.log 2>&1; tail -1 $O/alone.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/knn_time.py 9600 1 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/mfma -- python3 tools/knn_time.py 9600 1 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/sq -- python3 tools/knn_time.py 9600 1 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $O/sq2 -- python3 tools/knn_time.py 9600 1 16 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
def agg(pattern):
    out = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
            out[k][r['Counter_Name']] += float(r['Counter_Value']); 
            n[(k, r['Counter_Name'])] += 1
    return out, n
for d in ('mfma', 'sq', 'sq2'):
    out, n = agg('$O/%s/**/*counter_collection.csv' % d)
    for k in out:
        if 'coarse' in k or 'refine' in k or 'sweep16b' in k:
            print(d, k, {c: '%.3g' % (v / n[(k, c)]) for c, v in out[k].items()}, 'dispatches', max(n[(k, c)] for c in out[k]))
for f in glob.glob('$O/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(s in r['Name'] for s in ('coarse', 'refine', 'sweep16b', 'finalize', 'bucket')):
            print(r['Name'].split('(')[0][:50], r['Calls'], 'avg us', float(r['AverageNs']) / 1e3)
PY
