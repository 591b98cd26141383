cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
T=gpurun_out/r04j
mkdir -p $T
python3 tools/joinlb_time.py > $T/time.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $T/s -- python3 tools/joinlb_time.py 1 --reps 3 > $T/s.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $T/a -- python3 tools/joinlb_time.py 1 --reps 2 > $T/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $T/b -- python3 tools/joinlb_time.py 1 --reps 2 > $T/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $T/c -- python3 tools/joinlb_time.py 1 --reps 2 > $T/c.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $T/d -- python3 tools/joinlb_time.py 1 --reps 2 > $T/d.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $T/e -- python3 tools/joinlb_time.py 1 --reps 2 > $T/e.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in 'abcde':
    for f in glob.glob('gpurun_out/r04j/%s/**/*counter_collection.csv' % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'join_lb' in k or 'viterbi_lb' in k or 'join_exact' in k or 'viterbi_sparse' in k:
                acc[k.split('(')[0][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, d in acc.items():
            print(tag, k, {c: (round(sum(v) / len(v)), len(v)) for c, v in d.items()})
for f in glob.glob('gpurun_out/r04j/s/**/*kernel_stats.csv', recursive=True):
    print(open(f).read()[:3000])
PY
cat $T/time.log
