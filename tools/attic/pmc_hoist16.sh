cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
T=gpurun_out/r04h
mkdir -p $T
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $T/a -- python3 tools/b3_time.py > $T/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $T/b -- python3 tools/b3_time.py > $T/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $T/c -- python3 tools/b3_time.py > $T/c.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $T/d -- python3 tools/b3_time.py > $T/d.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $T/e -- python3 tools/b3_time.py > $T/e.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in 'abcde':
    for f in glob.glob('gpurun_out/r04h/%s/**/*counter_collection.csv' % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if 'hoist_product' in k:
                acc[k.split('(')[0][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, d in acc.items():
            print(tag, k, {c: (round(sum(v) / len(v)), len(v)) for c, v in d.items()})
PY
