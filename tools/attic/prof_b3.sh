# B3 greedy search (tools/b3_time.py) under rocprofv3: kernel times, then FETCH_SIZE / WRITE_SIZE / MFMA-busy in separate passes.
# usage on the GPU box: bash tools/prof_b3.sh   -> gpurun_out/b3prof/summary.txt
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
T=gpurun_out/b3prof
mkdir -p $T
rocprofv3 --kernel-trace --stats -d $T/t --output-format csv -- python3 tools/b3_time.py > $T/log.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $T/f --output-format csv -- python3 tools/b3_time.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $T/w --output-format csv -- python3 tools/b3_time.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $T/m --output-format csv -- python3 tools/b3_time.py > /dev/null 2>&1
python3 - <<'PY' > $T/summary.txt
import csv, glob, collections
T = 'gpurun_out/b3prof'
print('# B3 greedy search (1.5 M units, 600 frames, me 6): tools/b3_time.py under rocprofv3 (each launch = one utterance of 100 steps)')
f = glob.glob(T + '/t/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('greedy32_kernel', 'hoist_product', 'hoist_prepare', 'greedy_tile16', 'greedy32_init')):
        print('%-60s calls %4s  avg %10.1f us  total %8.2f ms' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
for tag, name in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE'), ('m', None)):
    for f in glob.glob(T + '/%s/**/*counter_collection.csv' % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'greedy32_kernel' in k or 'hoist_product' in k:
                acc[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in acc.items():
            print(k, {c: '%.4g (n=%d)' % (sum(v) / len(v), len(v)) for c, v in d.items()})
print(open(T + '/log.txt').read().strip().splitlines()[-2:])
PY
cat $T/summary.txt
