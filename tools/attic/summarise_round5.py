#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/prof_round5.sh (gpurun_out/<tag>/) into the tracked summaries under profiles/:
<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_summary.md, r05_traffic_joinlb2.json (what bench.py's `roofline.traffic` reads)
and r05_filter_counters.json (bench.py's `filter_stage`).   usage: tools/summarise_round5.py gpurun_out/r05a r05_a"""
import collections, csv, glob, hashlib, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, 'profiles')


def source_sha(name):
    """sha256 of a kernel source as it stands: bench.py counts a counter file as this build's only while it still matches"""
    with open(os.path.join(ROOT, 'snickery_amd', 'csrc', name), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    assert hits, pattern
    return max(hits, key=os.path.getmtime)


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(sub + '/**/*counter_collection.csv'))):
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


def avg(v):
    return sum(v) / max(len(v), 1)


bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(out, tag + '_bench.json'), 'w'), indent=1)
shutil.copy(one('stats/**/*kernel_stats.csv'), os.path.join(out, tag + '_kernel_stats.csv'))
stats = list(csv.DictReader(open(os.path.join(out, tag + '_kernel_stats.csv'))))
sb = json.loads(open(os.path.join(src, 'stats_bench.json')).read().strip().splitlines()[-1])
N, Dt, Dj, K = bench['config']['units'], bench['config']['target_dim'], bench['config']['join_dim'], bench['config']['n_candidates']

# ---- the dominant whole-chip kernel: join_lb2 ----
jf, _ = counters('jfetch')
jw, _ = counters('jwrite')
jm, jdur = counters('jmfma')
f, w, m = pick(jf, 'join_lb2_kernel'), pick(jw, 'join_lb2_kernel'), pick(jm, 'join_lb2_kernel')
rows = 16 * bench['config']['frames'] - 1
fetch_kb, write_kb = avg(f['FETCH_SIZE']), avg(w['WRITE_SIZE'])
alg = rows * (2 * K * Dj * 4 + K * K * 4)
jt = {'kernel': 'join_lb2_kernel<4> (one group of 16 utterances: %d row pairs)' % rows, 'rows_per_launch': float(rows),
      'fetch_size_kb_reported': fetch_kb, 'write_size_kb_reported': write_kb,
      'hbm_bytes_per_launch': fetch_kb * 1024 * 2 + write_kb * 1024, 'hbm_bytes_per_launch_fetch_not_doubled': fetch_kb * 1024 + write_kb * 1024,
      'algorithmic_bytes_per_launch': float(alg), 'source_sha256': source_sha('joinlb2_kernels.hip'),
      'note': 'separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/joinlb_time.py 1, averaged over its launches; FETCH_SIZE doubled as '
              'MI355X_MICROARCH.md (HBM section) prescribes for 16-byte-per-lane reads on gfx950 -- the kernel\'s loads are 16 bytes per lane but '
              'gathered (32 bytes of a row per lane pair), a width the guide calls uncalibrated, so the undoubled figure is kept beside it; '
              'algorithmic = 2 K rows of Dj float32 gathered + K^2 float32 bounds written per row pair (SURVEY 8d); rows that consecutive steps '
              'share (unit_end of a, unit_start of a + 1 are one row) are served from L2: traffic below algorithmic'}
json.dump(jt, open(os.path.join(out, 'r05_traffic_joinlb2.json'), 'w'), indent=1)

# ---- the filter stage ----
fetch, _ = counters('fetch')
write, _ = counters('write')
mfma, mdur = counters('mfma')
FILTER = ('knn_balls16b', 'knn_refine16b', 'knn_coarse16b')
per, tot = {}, 0.0
for k in FILTER:
    a, b = pick(fetch, k), pick(write, k)
    if not a:
        continue
    per[k] = {'fetch_bytes_corrected_x2': avg(a['FETCH_SIZE']) * 2048, 'write_bytes': avg(b['WRITE_SIZE']) * 1024}
    tot += per[k]['fetch_bytes_corrected_x2'] + per[k]['write_bytes']
rpl = bench['filter_stage']['rows_per_launch']
alg_f = N * 64 * 4 + rpl * Dt * 8 + rpl * K * 16
busy = {}
for k in FILTER + ('join_lb2_kernel',):
    c = pick(mfma, k) or pick(jm, k)
    if c:
        busy[k] = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / (sum(c['GRBM_GUI_ACTIVE']) / 8 * 1024)
fc = {'rows_per_launch': rpl, 'per_kernel': per, 'hbm_bytes_per_launch': tot, 'algorithmic_bytes_per_launch': alg_f,
      'traffic_ratio': tot / alg_f, 'mfma_busy': busy, 'source_sha256': source_sha('knn16_kernels.hip'),
      'note': 'separate --pmc passes over tools/prof_knn.py (two B* steps); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)'}
json.dump(fc, open(os.path.join(out, 'r05_filter_counters.json'), 'w'), indent=1)

tot_ns = sum(float(r['TotalDurationNs']) for r in stats)
lines = ['# Round 5, profile %s' % tag.split('_')[-1].upper(), '',
         'Commands (MI355X, 1 GPU, B* workload, 32 utterances per step, two steps in flight, query rows resident; tools/prof_round5.sh):', '',
         '* `python bench.py` (%d steps, %d warm-up) -> %s_bench.json: **%.0f frames/s** (%.3f ms per step; host to host, rows uploaded every '
         'step: %.0f); `roofline` = %s: %.0f GB/s of algorithmic bytes = **%.3f** of 8 TB/s, %.3f ms per launch by HIP events' % (
             bench['steps'], bench['warmup'], tag, bench['value'], bench['ms_per_step'], bench.get('with_upload', {}).get('value', 0),
             bench['roofline']['kernel'].split(' ')[0], bench['roofline']['achieved'], bench['roofline']['frac'], bench['roofline']['avg_launch_ms']),
         '* `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-greedy --no-variants` -> '
         '%s_kernel_stats.csv (that run: %.0f frames/s; join_lb2 by its HIP events %.3f ms per launch)' % (tag, sb['value'], sb['roofline']['avg_launch_ms']),
         '* counters: separate `--pmc` passes over tools/joinlb_time.py (the Viterbi side of one group alone) and tools/prof_knn.py', '',
         '| kernel | calls | avg us | total ms | % |', '|---|---|---|---|---|']
for r in stats[:18]:
    lines.append('| %s | %s | %.1f | %.2f | %s |' % (r['Name'][:62].replace('|', '/'), r['Calls'], float(r['AverageNs']) / 1e3,
                                                     float(r['TotalDurationNs']) / 1e6, r['Percentage']))
jl = [r for r in stats if 'join_lb2_kernel' in r['Name']]
lines += ['', '## The dominant whole-chip kernel: `join_lb2_kernel` (`roofline` of the bench line)', '']
if jl:
    lines.append('* stats pass (inside the pipeline, the K-NN of the next group sharing the chip): %s launches, average **%.3f ms** '
                 '(HIP events of the same run: %.3f ms; unprofiled bench run: %.3f ms)' % (jl[0]['Calls'], float(jl[0]['AverageNs']) / 1e6,
                                                                                           sb['roofline']['avg_launch_ms'], bench['roofline']['avg_launch_ms']))
d = None
for k, v in jdur.items():
    if 'join_lb2_kernel' in k:
        d = v
if m and d:
    gui = sum(m['GRBM_GUI_ACTIVE'])
    lines.append('* alone (tools/joinlb_time.py): average %.3f ms per launch of %d row pairs = %.0f GB/s of algorithmic bytes = %.3f of 8 TB/s; '
                 'clock %.2f GHz; matrix pipe busy %.3f; %.0f vector instructions per wavefront; wavefront cycles waiting for an issue slot %.2f, '
                 'parked at a wait or barrier %.2f' % (
                     avg(d) / 1e6, rows, alg / (avg(d) * 1e-9) / 1e9, alg / (avg(d) * 1e-9) / 8e12, gui / 8 / sum(d),
                     sum(m['SQ_VALU_MFMA_BUSY_CYCLES']) / (gui / 8 * 1024), avg(m['SQ_INSTS_VALU']) / (rows * 4.0),
                     sum(m['SQ_WAIT_INST_ANY']) / sum(m['SQ_WAVE_CYCLES']), sum(m['SQ_WAIT_ANY']) / sum(m['SQ_WAVE_CYCLES'])))
lines.append('* HBM traffic per launch: FETCH_SIZE %.0f KB reported (x2 = %.1f MB), WRITE_SIZE %.1f MB => %.1f MB (fetch doubled; %.1f MB undoubled) '
             'against %.1f MB algorithmic' % (fetch_kb, fetch_kb * 2048 / 1e6, write_kb * 1024 / 1e6, jt['hbm_bytes_per_launch'] / 1e6,
                                              jt['hbm_bytes_per_launch_fetch_not_doubled'] / 1e6, alg / 1e6))
lines += ['', '## The filter stage (`filter_stage` of the bench line)', '',
          '* HBM traffic per launch: ' + '; '.join('%s fetch x2 %.1f MB, write %.1f MB' % (k, v['fetch_bytes_corrected_x2'] / 1e6, v['write_bytes'] / 1e6) for k, v in per.items()) +
          ' => %.1f MB against %.1f MB algorithmic (%.2fx)' % (tot / 1e6, alg_f / 1e6, tot / alg_f),
          '* matrix pipe busy inside the batch step: ' + ', '.join('%s %.3f' % kv for kv in busy.items())]
for name, title in (('joinlb_alone.log', 'Viterbi side of one group (16 utterances) alone, per form of the bounds pass (tools/joinlb_time.py)'),
                    ('knn_alone.log', 'K-NN of one group (9 600 rows) alone (tools/knn_time.py)')):
    if os.path.isfile(os.path.join(src, name)):
        lines += ['', '## ' + title, '', '```'] + [l[:900] for l in open(os.path.join(src, name), errors='replace').read().splitlines()
                                                    if l.startswith(('join_lb_variant', 'prefilter'))] + ['```']
if os.path.isfile(os.path.join(src, 'onepass.log')):
    shutil.copy(os.path.join(src, 'onepass.log'), os.path.join(out, 'r05_onepass.log'))
    lines += ['', '## The one-pass three-term sweep alone (tools/onepass_time.py: 9 600 rows, B* database as generated / permuted / AR(1))', '', '```'] + \
             [l[:400] for l in open(os.path.join(src, 'onepass.log'), errors='replace').read().splitlines() if 'two_pass' in l] + ['```']
if os.path.isfile(os.path.join(src, 'minima.log')):
    lines += ['', '## The minima sweep (the kernel of stage A) alone over the whole database (tools/minima_time.py)', '', '```'] + \
             [l[:300] for l in open(os.path.join(src, 'minima.log'), errors='replace').read().splitlines() if l.startswith('minima sweep')] + ['```']
if os.path.isfile(os.path.join(src, 'single.log')):
    lines += ['', '## One utterance per call (snk_knn_viterbi, T = 600; tools/single_time.py)', '', '```'] + \
             [l[:700] for l in open(os.path.join(src, 'single.log'), errors='replace').read().splitlines() if 'viterbi_mode' in l and ('chunk 48 warm 16' in l or 'chunk 0' in l)] + ['```']
open(os.path.join(out, tag + '_summary.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
