"""Timeline of the persistent greedy scan (SNK_G32_TRACE): where a step's time goes.

    python tools/g32_trace.py [units = 65536] [utterances per scan = 1]

Stamps per step and workgroup (100 MHz clock): 0 step started, 1 table in LDS, 2 scan done, 3 workgroup record
published, 4 (decider) last arrival, 5 (decider) first decision made, 6 (decider) winners released; second phase (every
lane offers its candidates; the usual case with float16 join tiles): 7 entered, 12 offers made, 13 (decider) last
arrival, 15 control words read, 8 candidate ids read, 9 terms of the candidates stored, 10 chains summed, 14 decided."""
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
    dec = int(np.argmax(x[:, 4]))                     # the deciding workgroup: the only one with stamps 4..6
    if x[dec, 4] == 0:
        continue
    t0 = x[:, 0].min()
    rows.append([x[:, 0].max() - t0, np.median(x[:, 1]) - t0, np.median(x[:, 2]) - t0, x[:, 2].max() - t0, x[:, 3].max() - t0,
                 x[dec, 4] - t0, x[dec, 5] - t0, np.median(x[:, 7]) - t0, x[:, 12].max() - t0, x[:, 13].max() - t0, x[:, 15].max() - t0, x[:, 8].max() - t0, x[:, 9].max() - t0, x[:, 10].max() - t0, x[:, 14].max() - t0, x[dec, 6] - t0, np.median(t[s + 1][:, 0]) - t0, t[s + 1][:, 0].min() - t0])
r = np.array(rows)
names = ['last wg sees table', 'table in LDS (median)', 'scan done (median)', 'scan done (last)', 'published (last)', 'decider: last arrival',
         'decider: decided', '2nd phase: entered (median)', '2nd phase: offers made (last)', '2nd phase: last arrival', '2nd: control words read', '2nd: candidate ids read', '2nd: terms of the candidates stored', '2nd: chains summed', '2nd phase: decided', 'decider: released', 'next step seen (median)', 'next step seen (first)']
print('%d utterance(s) per scan; ' % NU + 'N = %d, %d workgroups; microseconds from the first workgroup seeing the step (mean over %d steps)' % (N, nb, len(rows)))
for n, v in zip(names, r.mean(0)):
    print('  %-28s %6.2f' % (n, v))
eng.close()
