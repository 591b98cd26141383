"""Timeline of the persistent greedy scan (SNK_G32_TRACE): where a step's time goes.

    python tools/g32_trace.py [units = 65536] [utterances per scan = 1]

Stamps per step and workgroup (100 MHz clock): 0 step started, 1 table in LDS, 2 scan done, 3 record published, 4 all records
gathered and classified; where a workgroup has to decide: 7 (a holder) its lanes' windows sent, 5 (decider) candidates collected,
8 candidate ids read, 9 terms of the candidates stored, 10 chains summed, 6 winner released.  In float16 scans the deciding workgroup
has usually decided BEFORE the gather (stamps 8 / 9 are then those of its speculation, 5 and 10 are missing)."""
import sys, os, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fn = '/tmp/g32_trace.bin'
os.environ['SNK_G32_TRACE'] = fn
import snickery_amd
from bench import synthetic_db, synthetic_targets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
NU = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Dt, Dj, T, me = 61, 151, 600, 6
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, False, 0)
Us = [synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(NU)]
for _ in range(3):
    eng.greedy(Us[0]) if NU == 1 else eng.greedy_batch(Us)
raw = open(fn, 'rb').read()
steps, nb = struct.unpack('qq', raw[:16])
t = np.frombuffer(raw[16:], dtype=np.uint64).reshape(steps, nb, 16).astype(np.float64) * 0.01      # us
rows = []
for s in range(8, 40):
    x = t[s]
    t0 = x[:, 0].min()
    dec = int(np.argmax(x[:, 6]))                     # the deciding workgroup (if any): the only one with stamp 6
    has = x[dec, 6] > 0
    f = lambda v: (v - t0) if (has and v > 0) else np.nan        # (a stamp that was not taken in this step -- e.g. a step decided before the gather -- is 0)
    rows.append([x[:, 0].max() - t0, np.median(x[:, 1]) - t0, np.median(x[:, 2]) - t0, x[:, 2].max() - t0, x[:, 3].max() - t0,
                 np.median(x[:, 4]) - t0, x[:, 4].max() - t0, f(x[:, 7].max()) if x[:, 7].max() > 0 else np.nan, f(x[dec, 5]), f(x[dec, 8]), f(x[dec, 9]), f(x[dec, 10]), f(x[dec, 6]),
                 np.median(t[s + 1][:, 0]) - t0, t[s + 1][:, 0].min() - t0])
r = np.array(rows)
names = ['last wg sees the step', 'table in LDS (median)', 'scan done (median)', 'scan done (last)', 'published (last)', 'gathered (median)', 'gathered (last)',
         'holders: lists sent (last)', 'decider: candidates collected', 'decider: candidate ids read', 'decider: terms stored', 'decider: chains summed', 'decider: released',
         'next step seen (median)', 'next step seen (first)']
print('%d utterance(s) per scan; ' % NU + 'N = %d, %d workgroups; microseconds from the first workgroup seeing the step (mean over %d steps; %d with a deciding workgroup)' % (N, nb, len(rows), int(np.sum(~np.isnan(r[:, 12])))))
for n, v in zip(names, np.nanmean(r, 0) if len(rows) else []):
    print('  %-32s %6.2f' % (n, v))
eng.close()
if os.environ.get('SNK_G32_DETAIL'):
    # where the scan ends late: per XCD (workgroup id mod 8) and the percentiles over workgroups, mean over the steps
    d = np.array([t[s][:, 2] - t[s][:, 0].min() for s in range(8, 40)])            # steps x workgroups
    m = d.mean(0)
    print('scan done, per workgroup (mean over steps): percentiles 0 10 50 90 100:', np.round(np.percentile(m, [0, 10, 50, 90, 100]), 2))
    print('  by workgroup id mod 8:', np.round([m[k::8].mean() for k in range(8)], 2))
    print('  by workgroup id // 32:', np.round([m[k * 32:(k + 1) * 32].mean() for k in range(8)], 2))
    print('  per step: spread (last - median):', np.round(np.mean(d.max(1) - np.median(d, 1)), 2), ' the same workgroup last every step?', np.bincount(d.argmax(1)).max(), 'of', d.shape[0])
    st = np.array([t[s][:, 0] - t[s][:, 0].min() for s in range(8, 40)])
    print('step start per workgroup: percentiles', np.round(np.percentile(st.mean(0), [0, 50, 100]), 2), ' table done:', np.round(np.percentile(np.array([t[s][:, 1] - t[s][:, 0].min() for s in range(8, 40)]).mean(0), [0, 50, 100]), 2))
