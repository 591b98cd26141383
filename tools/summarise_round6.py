#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/prof_round6.sh (gpurun_out/<tag>/) into the tracked summaries under profiles/:
<tag>_bench.json (the compact line), <tag>_bench_detail.json, <tag>_kernel_stats.csv, <tag>_summary.md, r06_traffic_joinlb2.json
(what bench.py's `roofline.traffic` / `issue_frac` read), r06_filter_counters.json (bench.py's `filter_stage`) and
r06_other_rooflines.json (pass 3 and the re-rank).   usage: tools/summarise_round6.py gpurun_out/r06a r06_a [--stale-ok]

Rules this file enforces:
  * rows per launch of join_lb2_kernel are READ from the kernel trace of the counter pass (grid size / workgroup size = row pairs),
    never assumed -- r05's summariser assumed 9 599 while the launches had 4 800 and printed 1.38 of the HBM peak;
  * no fraction of a peak above 1 is written: the script stops instead;
  * the kernel sources must be the ones the pass ran on (gpurun_out/<tag>/csrc.sha256, taken by prof_round6.sh before its first
    command): a tree that has moved on is refused (--stale-ok writes the summaries with STALE in their names' text and no counter
    files, for looking at an old pass)."""
import collections, csv, glob, hashlib, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
stale_ok = '--stale-ok' in sys.argv[3:]
# --counters-only [--out DIR]: only the counter files (r06_traffic_joinlb2.json, r06_filter_counters.json), before any bench line
# exists -- prof_round6.sh runs this on the box after the counter passes so that the bench run that follows can cite them
# (SNK_PROFILES_DIR); the full run at home writes the same files into profiles/
counters_only = '--counters-only' in sys.argv[3:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else os.path.join(ROOT, 'profiles')
HBM_PEAK = 8e12


def source_sha(name):
    with open(os.path.join(ROOT, name), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


# ---- the tree must be the profiled one ----
stale = []
for line in open(os.path.join(src, 'csrc.sha256')):
    sha, name = line.split()
    if not os.path.isfile(os.path.join(ROOT, name)) or source_sha(name) != sha:
        stale.append(name)
if stale and not stale_ok:
    sys.exit('refused: these sources changed since the pass was taken (profile again, or --stale-ok to look at it):\n  ' + '\n  '.join(stale))


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    assert hits, pattern
    return max(hits, key=os.path.getmtime)


def counters(sub):
    """per kernel name: counter -> values per dispatch (in dispatch order), workgroups per dispatch, durations (ns) from the trace"""
    agg = collections.defaultdict(lambda: collections.defaultdict(dict))
    wgs = collections.defaultdict(dict)
    for r in csv.DictReader(open(one(sub + '/**/*counter_collection.csv'))):
        agg[r['Kernel_Name']][r['Counter_Name']][int(r['Dispatch_Id'])] = float(r['Counter_Value'])
        wgs[r['Kernel_Name']][int(r['Dispatch_Id'])] = int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(one(sub + '/**/*kernel_trace.csv'))):
        dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return agg, wgs, dur


def pick(d, needle):
    for k, v in d.items():
        if needle in k:
            return v
    return None


def avg(v):
    v = list(v.values()) if isinstance(v, dict) else list(v)
    return sum(v) / max(len(v), 1)


def frac_ok(x, what):
    if not (0.0 <= x <= 1.0):
        sys.exit('refused: %s = %.3f is not a fraction of a peak (a wrong byte count or a wrong time)' % (what, x))
    return x


if counters_only:
    N, Dt, Dj, K = 1048576, 61, 302, 100                    # B*: what tools/joinlb_time.py and tools/prof_knn.py run
    detail = {'filter_stage': {'rows_per_launch': 9600.0}}
else:
    line = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
    detail = json.load(open(os.path.join(src, 'bench_detail.json')))
    assert len(json.dumps(line)) < 6144
    json.dump(line, open(os.path.join(out, tag + '_bench.json'), 'w'), indent=1)
    json.dump(detail, open(os.path.join(out, tag + '_bench_detail.json'), 'w'), indent=1)
    shutil.copy(one('stats/**/*kernel_stats.csv'), os.path.join(out, tag + '_kernel_stats.csv'))
    stats = list(csv.DictReader(open(os.path.join(out, tag + '_kernel_stats.csv'))))
    sb = json.loads(open(os.path.join(src, 'stats_bench.json')).read().strip().splitlines()[-1])
    cfg = line['config']
    N, Dt, Dj, K = cfg['units'], cfg['target_dim'], cfg['join_dim'], cfg['n_candidates']


def traffic(fetch_sub, write_sub, needle):
    """HBM bytes per launch of a kernel from its FETCH_SIZE / WRITE_SIZE passes (KB as reported; FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950), with the workgroups per launch of the SAME dispatches."""
    f, fw, _ = counters(fetch_sub)
    w, ww, _ = counters(write_sub)
    cf, cw = pick(f, needle), pick(w, needle)
    if not cf or not cw:
        return None
    wg_f, wg_w = pick(fw, needle), pick(ww, needle)
    assert sorted(set(wg_f.values())) == sorted(set(wg_w.values())), (needle, set(wg_f.values()), set(wg_w.values()))
    fetch_kb, write_kb = avg(cf['FETCH_SIZE']), avg(cw['WRITE_SIZE'])
    return {'fetch_size_kb_reported': fetch_kb, 'write_size_kb_reported': write_kb, 'workgroups_per_launch': avg(wg_f),
            'workgroups_seen': sorted(set(wg_f.values())), 'launches_counted': len(cf['FETCH_SIZE']),
            'hbm_bytes_per_launch': fetch_kb * 1024 * 2 + write_kb * 1024, 'hbm_bytes_per_launch_fetch_not_doubled': fetch_kb * 1024 + write_kb * 1024}


# ---- the dominant whole-chip kernel: join_lb2 (alone: the Viterbi side of one group, tools/joinlb_time.py) ----
jt = traffic('jfetch', 'jwrite', 'join_lb2_kernel')
jm, jmw, jdur = counters('jmfma')
m = pick(jm, 'join_lb2_kernel')
d = pick(jdur, 'join_lb2_kernel')
pairs = jt['workgroups_per_launch']                       # one workgroup per row pair (quadrants off at K <= 128)
assert len(jt['workgroups_seen']) == 1, jt['workgroups_seen']
alg = pairs * (2 * K * Dj * 4 + K * K * 4)
gui = sum(m['GRBM_GUI_ACTIVE'].values())
simd_cycles = gui / 8 * 1024                              # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
n_mfma = sum(m['SQ_VALU_MFMA_BUSY_CYCLES'].values()) / 32.0     # a v_mfma_f32_32x32x16_bf16 keeps the pipe busy 32 cycles
n_valu = sum(m['SQ_INSTS_VALU'].values())
issue_frac = (4.0 * n_valu + 8.0 * n_mfma) / simd_cycles
alone_ms = avg(d) / 1e6
jt.update({'kernel': 'join_lb2_kernel (tools/joinlb_time.py 1: the Viterbi side of one group of 16 utterances, nothing beside it)',
           'rows_per_launch': float(pairs), 'rows_from_trace': True, 'algorithmic_bytes_per_launch': float(alg),
           'traffic_over_algorithmic': jt['hbm_bytes_per_launch'] / alg,
           'alone_avg_launch_ms': alone_ms, 'alone_frac_of_hbm_peak': frac_ok(alg / (alone_ms * 1e-3) / HBM_PEAK, 'join_lb2 alone'),
           'issue_frac': issue_frac, 'mfma_busy': sum(m['SQ_VALU_MFMA_BUSY_CYCLES'].values()) / simd_cycles,
           'valu_instructions_per_wavefront': n_valu / (pairs * len(m['SQ_INSTS_VALU']) * 4.0),
           'clock_ghz': gui / 8 / sum(d),
           'source_sha256': source_sha('snickery_amd/csrc/joinlb2_kernels.hip'),
           'note': 'separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_* + GRBM_GUI_ACTIVE) over tools/joinlb_time.py 1 --reps 2, averaged over '
                   'its launches; rows per launch = grid size / workgroup size of those very dispatches; FETCH_SIZE doubled as '
                   'MI355X_MICROARCH.md prescribes on gfx950 (the undoubled figure beside it); algorithmic = 2 K rows of Dj float32 '
                   'gathered + K^2 float32 bounds written per row pair (SURVEY 8d); issue_frac = (4 SQ_INSTS_VALU + 8 MFMA) / SIMD cycles '
                   '(cycle constants of MI355X_MICROARCH.md; SQ_INSTS_VALU may count the MFMAs too: then an upper figure by 7 %)'})
frac_ok(jt['issue_frac'], 'join_lb2 issue_frac')
if not stale:
    json.dump(jt, open(os.path.join(out, 'r06_traffic_joinlb2.json'), 'w'), indent=1)

# ---- the filter stage and the re-rank (two B* steps through the batch entry point, tools/prof_knn.py) ----
fetch, fwg, _ = counters('fetch')
write, _, _ = counters('write')
mfma, _, mdur = counters('mfma')
FILTER = ('knn_balls16b', 'knn_refine16b', 'knn_coarse16b')
per, tot = {}, 0.0
for k in FILTER:
    a, b = pick(fetch, k), pick(write, k)
    if not a:
        continue
    per[k] = {'fetch_bytes_corrected_x2': avg(a['FETCH_SIZE']) * 2048, 'write_bytes': avg(b['WRITE_SIZE']) * 1024}
    tot += per[k]['fetch_bytes_corrected_x2'] + per[k]['write_bytes']
rpl = detail['filter_stage']['rows_per_launch']
alg_f = N * 64 * 4 + rpl * Dt * 8 + rpl * K * 16
busy = {}
for k in FILTER + ('join_lb2_kernel',):
    c = pick(mfma, k) or pick(jm, k)
    if c:
        busy[k] = sum(c['SQ_VALU_MFMA_BUSY_CYCLES'].values()) / (sum(c['GRBM_GUI_ACTIVE'].values()) / 8 * 1024)
fc = {'rows_per_launch': rpl, 'per_kernel': per, 'hbm_bytes_per_launch': tot, 'algorithmic_bytes_per_launch': alg_f,
      'traffic_ratio': tot / alg_f, 'mfma_busy': busy, 'source_sha256': source_sha('snickery_amd/csrc/knn16_kernels.hip'),
      'note': 'separate --pmc passes over tools/prof_knn.py (two B* steps); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)'}
if not stale:
    json.dump(fc, open(os.path.join(out, 'r06_filter_counters.json'), 'w'), indent=1)

if counters_only:
    print('counter files written to', out)
    sys.exit(0)

# ---- the two kernels without a roofline until round 5: pass 3 of the Viterbi side, the re-rank ----
other = {}
orl = detail.get('other_rooflines', {})
for needle, fs, ws, alone in (('join_exact_sparse2_kernel', 'jfetch', 'jwrite', True), ('knn_finalize_kernel', 'fetch', 'write', False)):
    t = traffic(fs, ws, needle)
    if not t or needle not in orl:
        continue
    o = dict(orl[needle])
    st = [r for r in stats if needle in r['Name']]
    o.update({'traffic': t['hbm_bytes_per_launch'], 'traffic_fetch_not_doubled': t['hbm_bytes_per_launch_fetch_not_doubled'],
              'traffic_over_algorithmic': t['hbm_bytes_per_launch'] / o['algorithmic_bytes_per_launch'],
              'traffic_pass': ('tools/joinlb_time.py 1 (one group alone)' if alone else 'tools/prof_knn.py (two B* steps)') +
                              ': %d launches counted, %s workgroups each' % (t['launches_counted'], t['workgroups_seen']),
              'stats_pass': [{'name': r['Name'][:70], 'calls': int(r['Calls']), 'avg_ms': float(r['AverageNs']) / 1e6} for r in st]})
    frac_ok(o['frac'], needle)
    other[needle] = o
if not stale:
    json.dump(other, open(os.path.join(out, 'r06_other_rooflines.json'), 'w'), indent=1)

# ---- the summary ----
ro = line['roofline']
lines = ['# Round 6, profile %s%s' % (tag.split('_')[-1].upper(), '  (STALE: the tree has moved on since this pass)' if stale else ''), '',
         'Commands (MI355X, 1 GPU, B* workload, 32 utterances per step, %d steps in flight, query rows uploaded and paths returned inside '
         'every timed step; `tools/prof_round6.sh`; kernel sources as in `gpurun_out/<tag>/csrc.sha256`, checked against the tree by this script):' % line['config'].get('steps_in_flight', 2), '',
         '* `python bench.py --steps %d --warmup %d` -> %s_bench.json (the compact line the driver parses, %d bytes) + %s_bench_detail.json: '
         '**%.0f frames/s host -> host** (%.3f ms per step; rows resident in HBM: %s); `roofline` = %s: %.0f GB/s of algorithmic bytes = '
         '**%.3f** of 8 TB/s, %.3f ms per launch of %d rows by HIP events inside the pipeline' % (
             line['steps'], line['warmup'], tag, len(json.dumps(line)), tag, line['value'], line['ms_per_step'],
             line['summary'].get('resident_rows_frames_per_s'), ro['kernel'], ro['achieved'], ro['frac'], ro['avg_launch_ms'], ro['rows_per_launch']),
         '* `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-greedy --no-variants --no-shapes '
         '--steps 20 --warmup 5` -> %s_kernel_stats.csv (that run: %.0f frames/s; join_lb2 by its HIP events %.3f ms per launch)' % (
             tag, sb['value'], sb['roofline']['avg_launch_ms']),
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
                                                                                           sb['roofline']['avg_launch_ms'], ro['avg_launch_ms']))
lines.append('* alone (tools/joinlb_time.py, counter pass): %d row pairs per launch (grid size of the counted dispatches), average %.3f ms = %.0f GB/s of '
             'algorithmic bytes = **%.3f** of 8 TB/s; clock %.2f GHz; matrix pipe busy %.3f; %.0f vector instructions per wavefront; '
             '**issue_frac %.3f** = (4 SQ_INSTS_VALU + 8 MFMA) / SIMD cycles; wavefront cycles waiting for an issue slot %.2f, parked at a wait or barrier %.2f' % (
                 pairs, alone_ms, alg / (alone_ms * 1e-3) / 1e9, jt['alone_frac_of_hbm_peak'], jt['clock_ghz'], jt['mfma_busy'],
                 jt['valu_instructions_per_wavefront'], issue_frac,
                 sum(m['SQ_WAIT_INST_ANY'].values()) / sum(m['SQ_WAVE_CYCLES'].values()), sum(m['SQ_WAIT_ANY'].values()) / sum(m['SQ_WAVE_CYCLES'].values())))
lines.append('* HBM traffic per launch: FETCH_SIZE %.0f KB reported (x2 = %.1f MB), WRITE_SIZE %.1f MB => %.1f MB (fetch doubled; %.1f MB undoubled) '
             'against %.1f MB algorithmic = %.2f x' % (jt['fetch_size_kb_reported'], jt['fetch_size_kb_reported'] * 2048 / 1e6, jt['write_size_kb_reported'] * 1024 / 1e6,
                                                      jt['hbm_bytes_per_launch'] / 1e6, jt['hbm_bytes_per_launch_fetch_not_doubled'] / 1e6, alg / 1e6, jt['traffic_over_algorithmic']))
if os.path.isdir(os.path.join(src, 'sfetch')):
    sp = traffic('sfetch', 'swrite', 'join_lb2_kernel')
    if sp:
        salg = sp['workgroups_per_launch'] * (2 * K * Dj * 4 + K * K * 4)
        lines.append('* the same kernel on the speech-like voice (AR(1) join rows, held-out utterances; tools/joinlb_time.py 1 --speechlike): %d row pairs per launch, '
                     '%.1f MB of traffic (fetch doubled; %.1f MB undoubled) against %.1f MB algorithmic = %.2f x' % (
                         sp['workgroups_per_launch'], sp['hbm_bytes_per_launch'] / 1e6, sp['hbm_bytes_per_launch_fetch_not_doubled'] / 1e6, salg / 1e6,
                         sp['hbm_bytes_per_launch'] / salg))
lines += ['', '## The filter stage (`filter_stage` of the bench line)', '',
          '* HBM traffic per launch: ' + '; '.join('%s fetch x2 %.1f MB, write %.1f MB' % (k, v['fetch_bytes_corrected_x2'] / 1e6, v['write_bytes'] / 1e6) for k, v in per.items()) +
          ' => %.1f MB against %.1f MB algorithmic (%.2fx)' % (tot / 1e6, alg_f / 1e6, tot / alg_f),
          '* matrix pipe busy inside the batch step: ' + ', '.join('%s %.3f' % kv for kv in busy.items())]
lines += ['', '## Pass 3 and the re-rank (r06_other_rooflines.json)', '']
for k, o in other.items():
    lines.append('* `%s`: %.3f ms per launch by HIP events inside the pipeline (stats pass: %s), algorithmic %.1f MB per launch = **%.3f** of 8 TB/s; '
                 'counter traffic %.1f MB (fetch doubled; %.1f undoubled) = %.2f x algorithmic%s' % (
                     k, o['avg_launch_ms'], ', '.join('%.3f ms x %d' % (s['avg_ms'], s['calls']) for s in o['stats_pass']) or 'n/a',
                     o['algorithmic_bytes_per_launch'] / 1e6, o['frac'], o['traffic'] / 1e6, o['traffic_fetch_not_doubled'] / 1e6, o['traffic_over_algorithmic'],
                     '; float64 vector operations %.3f of the non-FMA rate' % o['f64_vector_frac'] if 'f64_vector_frac' in o else ''))
for name, title in (('joinlb_alone.log', 'Viterbi side of one group (16 utterances) alone, per form of the bounds pass (tools/joinlb_time.py)'),
                    ('joinlb_speechlike.log', 'The same on the speech-like voice (tools/joinlb_time.py 1 --speechlike)'),
                    ('knn_alone.log', 'K-NN of one group (9 600 rows) alone (tools/knn_time.py)')):
    if os.path.isfile(os.path.join(src, name)):
        lines += ['', '## ' + title, '', '```'] + [l[:900] for l in open(os.path.join(src, name), errors='replace').read().splitlines()
                                                    if l.startswith(('join_lb_variant', 'prefilter'))] + ['```']
if os.path.isfile(os.path.join(src, 'onepass.log')):
    shutil.copy(os.path.join(src, 'onepass.log'), os.path.join(out, 'r06_onepass.log'))
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
