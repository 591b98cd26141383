#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/prof_round3.sh (gpurun_out/<tag>/) into the tracked summaries under profiles/:
<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_summary.md and r03_traffic_filter.json (what bench.py's `traffic` reads).
usage: tools/summarise_round3.py gpurun_out/r03a r03_a"""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, 'profiles')


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    assert hits, pattern
    return max(hits, key=os.path.getmtime)


def counters(sub):
    """kernel -> counter -> list of per-dispatch values; and kernel -> list of dispatch durations (ns) of that pass"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    f = one(sub + '/**/*counter_collection.csv')
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(one(sub + '/**/*kernel_trace.csv'))):
        dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return agg, dur


def pick(agg, needle):
    for k, v in agg.items():
        if needle in k:
            return v
    return None


bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(out, tag + '_bench.json'), 'w'), indent=1)
shutil.copy(one('stats/**/*kernel_stats.csv'), os.path.join(out, tag + '_kernel_stats.csv'))
stats = list(csv.DictReader(open(os.path.join(out, tag + '_kernel_stats.csv'))))
sb = json.loads(open(os.path.join(src, 'stats_bench.json')).read().strip().splitlines()[-1])

fetch, _ = counters('fetch')
write, _ = counters('write')
mfma, mdur = counters('mfma')
mfma_a, mdur_a = counters('mfma_alone')
rows_per_launch = bench['roofline']['rows_per_launch']
N, Dt, K = bench['config']['units'], bench['config']['target_dim'], bench['config']['n_candidates']
FILTER = ('knn_balls16b', 'knn_refine16b', 'knn_coarse16b')
per = {}
tot_f = tot_w = 0.0
for k in FILTER:
    f, w = pick(fetch, k), pick(write, k)
    if not f:
        continue
    fb = sum(f['FETCH_SIZE']) / len(f['FETCH_SIZE']) * 1024 * 2      # KB -> B, x2: gfx950 correction (MI355X_MICROARCH.md, HBM section)
    wb = sum(w['WRITE_SIZE']) / len(w['WRITE_SIZE']) * 1024
    per[k] = {'fetch_size_kb_reported': sum(f['FETCH_SIZE']) / len(f['FETCH_SIZE']), 'fetch_bytes_corrected_x2': fb, 'write_bytes': wb,
              'launches': len(f['FETCH_SIZE'])}
    tot_f += fb; tot_w += wb
traffic = {'kernel': ' + '.join(per.keys()) + ' (the filter stage of one group)', 'rows_per_launch': rows_per_launch, 'per_kernel': per,
           'hbm_bytes_per_launch': tot_f + tot_w,
           'algorithmic_bytes_per_launch': N * 64 * 4 + rows_per_launch * Dt * 8 + rows_per_launch * K * 16,
           'note': 'separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/prof_knn.py, averaged over the launches of the two steps; '
                   'FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section; algorithmic = the database once as two bf16 pieces of 64 '
                   'columns + the query rows + K survivor entries per row (SURVEY 8d)'}
json.dump(traffic, open(os.path.join(out, 'r03_traffic_filter.json'), 'w'), indent=1)


def busy_line(agg, dur, needle, where):
    c = pick(agg, needle)
    if not c:
        return None
    busy, gui = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']), sum(c['GRBM_GUI_ACTIVE'])
    d = None
    for k, v in dur.items():
        if needle in k:
            d = v
    clock = gui / 8 / sum(d) if d else 0.0
    return '* `%s` %s: %d launches, average %.3f ms; SQ_VALU_MFMA_BUSY_CYCLES %.4g over GRBM_GUI_ACTIVE/8 x 1024 SIMDs = %.4g -> matrix pipe busy **%.3f**; clock %.2f GHz' % (
        needle, where, len(c['GRBM_GUI_ACTIVE']), (sum(d) / len(d) / 1e6) if d else 0.0, busy, gui / 8 * 1024, busy / (gui / 8 * 1024), clock)


lines = ['# Round 3, profile %s' % tag.split('_')[-1].upper(), '',
         'Commands (MI355X, 1 GPU, B* workload, 32 utterances per step, two steps in flight, query rows resident; tools/prof_round3.sh):', '',
         '* `python bench.py` (%d steps, %d warm-up) -> %s_bench.json: **%.0f frames/s** (%.3f ms per step; with the rows uploaded every step %.0f; '
         'one step at a time %.0f), roofline.frac %.3f of the %.1f TFLOP/s bf16 matrix peak on the algorithmic flops (issued on the pipe: %.3f), '
         'cpu_baseline %.1f frames/s on one core, %.1f on %d' % (
             bench['steps'], bench['warmup'], tag, bench['value'], bench['ms_per_step'], bench.get('with_upload', {}).get('value', 0),
             bench.get('one_in_flight', {}).get('value', 0), bench['roofline']['frac'], bench['roofline']['peak'],
             bench['roofline'].get('issued', {}).get('frac', 0), bench['cpu_baseline']['value'],
             bench['cpu_baseline'].get('all_cores', {}).get('value', 0), bench['cpu_baseline'].get('all_cores', {}).get('cores', 0)),
         '* `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-greedy` -> %s_kernel_stats.csv '
         '(that run: %.0f frames/s; filter stage by its HIP events %.3f ms per launch)' % (tag, sb['value'], sb['roofline']['avg_launch_ms']),
         '* `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/prof_knn.py`, the same with `WRITE_SIZE` and with '
         '`SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`; the last one also over `tools/knn_time.py 9600 1 16` (the K-NN of one group alone) '
         '-> below / r03_traffic_filter.json', '',
         '| kernel | calls | avg us | total ms | % |', '|---|---|---|---|---|']
for r in stats[:16]:
    lines.append('| %s | %s | %.1f | %.2f | %s |' % (r['Name'][:62].replace('|', '/'), r['Calls'], float(r['AverageNs']) / 1e3,
                                                     float(r['TotalDurationNs']) / 1e6, r['Percentage']))
fs = {k: [r for r in stats if k in r['Name']] for k in FILTER}
fsum = sum(float(v[0]['AverageNs']) for v in fs.values() if v and k != 'knn_coarse16b') / 1e6
lines += ['', '## The filter stage (`roofline` of the bench line)', '',
          '* stats pass, average per launch: ' + ', '.join('%s %.3f ms (%s launches)' % (k, float(v[0]['AverageNs']) / 1e6, v[0]['Calls']) for k, v in fs.items() if v) +
          '; ball pass + refine pass together %.3f ms against %.3f ms between the stage\'s HIP events in the same run (%.3f ms in the '
          'unprofiled bench run): the events also cover `ball_query_terms_kernel` and the gap between the two launches, on a chip shared '
          'with the Viterbi side of the previous group' % (
              sum(float(v[0]['AverageNs']) for k, v in fs.items() if v and k != 'knn_coarse16b') / 1e6, sb['roofline']['avg_launch_ms'],
              bench['roofline']['avg_launch_ms'])]
for needle in FILTER + ('join_lb_kernel',):
    for agg, dur, where in ((mfma, mdur, 'inside the batch step'), (mfma_a, mdur_a, 'K-NN of one group alone')):
        l = busy_line(agg, dur, needle, where)
        if l:
            lines.append(l)
lines += ['* HBM traffic of the stage per launch (sum of its kernels; average of the launches of the two steps): ' +
          '; '.join('%s FETCH_SIZE %.0f KB reported -> x2 = %.1f MB, WRITE_SIZE %.1f MB' % (k, v['fetch_size_kb_reported'], v['fetch_bytes_corrected_x2'] / 1e6, v['write_bytes'] / 1e6)
                    for k, v in per.items()) +
          ' => %.1f MB against %.1f MB algorithmic (%.2fx; the ball pass reads the centres, 1 / 32 of the database, the refine pass the '
          'listed tiles)' % (traffic['hbm_bytes_per_launch'] / 1e6, traffic['algorithmic_bytes_per_launch'] / 1e6,
                             traffic['hbm_bytes_per_launch'] / traffic['algorithmic_bytes_per_launch'])]
if os.path.isfile(os.path.join(src, 'knn_alone.log')):
    lines += ['', '## K-NN of one group (9 600 rows) alone, stage times (tools/knn_time.py)', '', '```', open(os.path.join(src, 'knn_alone.log')).read().strip().splitlines()[-1][:900], '```']
if os.path.isfile(os.path.join(src, 'single.log')):
    lines += ['', '## One utterance per call (snk_knn_viterbi, T = 600; tools/single_time.py)', '', '```'] + \
             [l[:700] for l in open(os.path.join(src, 'single.log'), errors='replace').read().splitlines() if 'viterbi_mode' in l and ('chunk 48 warm 16' in l or 'chunk 0' in l)] + ['```']
open(os.path.join(out, tag + '_summary.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
