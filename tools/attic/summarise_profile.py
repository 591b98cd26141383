#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/prof_round.sh (gpurun_out/<tag>/) into the tracked
summaries under profiles/: <round>_bench.json, <round>_kernel_stats.csv, <round>_summary.md and
the traffic JSON bench.py reads.   usage: tools/summarise_profile.py gpurun_out/r01d r01_d"""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, 'profiles')


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern))
    assert hits, pattern
    return max(hits, key=os.path.getmtime)          # the newest run


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(sub + '/*/*_counter_collection.csv'))):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


def pick(agg, needle):
    for k, v in agg.items():
        if needle in k:
            return v
    raise KeyError(needle)


bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(out, tag + '_bench.json'), 'w'), indent=1)
shutil.copy(one('stats/*/*_kernel_stats.csv'), os.path.join(out, tag + '_kernel_stats.csv'))
stats = list(csv.DictReader(open(os.path.join(out, tag + '_kernel_stats.csv'))))

BF16 = 'sweep16b' in bench['roofline']['kernel']
FILT = 'knn_sweep16b<4, 1, 4, 3>' if BF16 else 'knn_sweep16<4, 1, 1, false'
fetch = pick(counters('fetch'), FILT)['FETCH_SIZE']
write = pick(counters('write'), FILT)['WRITE_SIZE']
mf = pick(counters('mfma'), FILT)
fetch_b = sum(fetch) / len(fetch) * 1024 * 2          # KB -> B, x2: gfx950 correction (MI355X_MICROARCH.md)
write_b = sum(write) / len(write) * 1024
rows_per_launch = bench['roofline']['rows_per_launch']
N, Dt = bench['config']['units'], bench['config']['target_dim']
traffic = {
    'kernel': 'knn_sweep16b<4,1,4,3> (bf16-split filter)' if BF16 else 'knn_sweep16<4,1,1,false> (f32 filter)', 'rows_per_launch': rows_per_launch,
    'fetch_size_kb_reported': sum(fetch) / len(fetch), 'fetch_bytes_corrected_x2': fetch_b, 'write_bytes': write_b,
    'hbm_bytes_per_launch': fetch_b + write_b,
    'algorithmic_bytes_per_launch': N * 64 * 4 + rows_per_launch * Dt * 8 + rows_per_launch * 100 * 16,
    'note': 'separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/prof_knn.py, averaged over the filter '
            'launches of a step; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; the writes are the '
            '16-byte survivor entries of the candidate pool',
}
json.dump(traffic, open(os.path.join(out, 'r02_traffic_bf16.json' if BF16 else 'r01_traffic_f32.json'), 'w'), indent=1)

lines = ['# Round %s, profile %s -- grouped batch pipeline (K-NN per group of utterances; join bounds, sparse exact '
         'recursion per group on side streams)' % (tag[1:3].lstrip('0'), tag.split('_')[-1].upper()), '',
         'Commands (MI355X, 1 GPU, B* workload, 32 utterances per step; tools/prof_round.sh):', '',
         '* `python bench.py` (%d steps, %d warm-up) -> %s_bench.json: **%.0f frames/s** (xRT %.0f), roofline.frac %.3f of '
         'the %.1f TFLOP/s matrix peak%s, cpu_baseline %.1f frames/s' % (
             bench['steps'], bench['warmup'], tag, bench['value'], bench['xRT'], bench['roofline']['frac'], bench['roofline']['peak'],
             ' (issued on the pipe: %.3f)' % bench['roofline']['issued']['frac'] if 'issued' in bench['roofline'] else '',
             bench['cpu_baseline']['value']),
         '* `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 '
         '--no-cpu-baseline --no-greedy` -> %s_kernel_stats.csv' % tag,
         '* `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/prof_knn.py`, same with `WRITE_SIZE`, '
         '`SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` and an SQ pass -> below / %s' % ('r02_traffic_bf16.json' if BF16 else 'r01_traffic_f32.json'), '',
         '| kernel | calls | avg us | total ms | % |', '|---|---|---|---|---|']
for r in stats[:14]:
    lines.append('| %s | %s | %.1f | %.2f | %s |' % (r['Name'][:62].replace('|', '/'), r['Calls'], float(r['AverageNs']) / 1e3,
                                                     float(r['TotalDurationNs']) / 1e6, r['Percentage']))
busy = sum(mf['SQ_VALU_MFMA_BUSY_CYCLES'])
# duration of the same dispatches in the counter pass -> clock
kt = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(one('mfma/*/*_kernel_trace.csv')))}
dur_ns = sum(kt[r['Dispatch_Id']] for r in csv.DictReader(open(one('mfma/*/*_counter_collection.csv')))
             if FILT in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE')
gui = sum(mf['GRBM_GUI_ACTIVE'])
clock_ghz = gui / 8 / dur_ns if dur_ns else 0.0
filt = [r for r in stats if FILT in r['Name']][0]
lines += ['', '(the T-step recursions run on side streams, a workgroup per utterance; they overlap the K-NN of the next group '
          'and of the next step.)', '',
          '## %s counters' % FILT, '',
          '* stats pass: %s launches, average %.3f ms (bench.py\'s HIP-event average: %.3f ms)' % (
              filt['Calls'], float(filt['AverageNs']) / 1e6, bench['roofline']['avg_launch_ms']),
          '* matrix pipe: SQ_VALU_MFMA_BUSY_CYCLES %.4g over GRBM_GUI_ACTIVE/8 x 1024 SIMDs = %.4g -> busy fraction **%.3f**' % (
              busy, gui / 8 * 1024, busy / (gui / 8 * 1024)),
          '* clock under this kernel: GRBM_GUI_ACTIVE / 8 / duration = %.2f GHz (counter pass)' % clock_ghz,
          '* HBM traffic per launch (average of the launches of a step): FETCH_SIZE %.0f KB reported -> x2 (gfx950 correction) '
          '= %.0f MB; WRITE_SIZE %.0f MB (survivor entries); algorithmic %.0f MB => %.2fx' % (
              traffic['fetch_size_kb_reported'], fetch_b / 1e6, write_b / 1e6, traffic['algorithmic_bytes_per_launch'] / 1e6,
              traffic['hbm_bytes_per_launch'] / traffic['algorithmic_bytes_per_launch'])]
open(os.path.join(out, tag + '_summary.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
