"""One bench leg (B* on the speech-like voice, or --compact: SURVEY 8d's walk) through the batch pipeline, host -> host, three steps in
flight, with the stage timers off / roofline stage only / all (option timers 0 / 2 / 1), in alternating order: does the rate depend
on the timestamp events a timed stage puts on its stream?     python tools/leg_timers_ab.py [--compact] [--opt name=value ...]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
snickery_amd.configure_runtime()
from bench import synthetic_db, synthetic_targets, speechlike_voice
N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 32
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
if '--compact' in sys.argv:
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    utts = [synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(U)]
else:
    F_unw, JC_unw, held_out = speechlike_voice(N, Dt, Dj, seed=0)
    utts = [held_out(T, u) * wt for u in range(U)]
eng = snickery_amd.HipSearchEngine(0)
args = sys.argv[1:]
for i, a in enumerate(args):
    if a == '--opt':
        n, v = args[i + 1].split('='); eng.set_option(n, float(v))
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
batch = snickery_amd.QueryBatch(utts).pin()
for _ in range(12):
    eng.knn_viterbi_batch_collect(eng.knn_viterbi_batch_submit(batch, K))


def run(steps=20, depth=3):
    pending = []
    t0 = time.perf_counter()
    for _ in range(steps):
        pending.append(eng.knn_viterbi_batch_submit(batch, K))
        if len(pending) >= depth:
            eng.knn_viterbi_batch_collect(pending.pop(0))
    while pending:
        eng.knn_viterbi_batch_collect(pending.pop(0))
    return T * U * steps / (time.perf_counter() - t0)
for mode in (2, 1, 0, 1, 2, 0, 2, 1):
    eng.set_option('timers', mode)
    eng.reset_timers()
    print('timers %d: %.0f frames/s' % (mode, run()), flush=True)
eng.close()
