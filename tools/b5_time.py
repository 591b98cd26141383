"""Scratch: BASELINE config 5 shape (halfphone: N = 1.3 M, Dt = 184, Dj = 151, T = 120, K = 100):
batch of 16 utterances, f32 prefilter (three 64-column chunks) against the f64 sweep; then the
monophone-restricted K-NN (45 classes, Zipf-distributed sizes)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, T, K, U = 1300000, 184, 151, 120, 100, 16
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.3); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
utts = [synthetic_targets(F_unw, T, seed=1 + s) * wt for s in range(U)]
ref = None
for prec in (1, 0):
    eng.set_option('precision', prec)
    eng.knn_viterbi_batch(utts, K)
    eng.reset_timers()
    t0 = time.time()
    for rep in range(3):
        paths, costs = eng.knn_viterbi_batch(utts, K)
    dt = (time.time() - t0) / 3
    if ref is None: ref = (paths, costs)
    same = all(np.array_equal(a, b) for a, b in zip(paths, ref[0])) and np.array_equal(costs, ref[1])
    tm = eng.timers()
    flops = 2.0 * U * T * N * Dt
    print('precision=%d: %.2f ms/step  %.0f units/s  same=%s fallbacks=%d  filter %.2f ms = %.1f TFLOP/s' % (
        prec, dt * 1e3, U * T / dt, same, eng.info('f16_fallbacks'), tm['knn_filter'][0] / 3,
        flops / (tm['knn_filter'][0] / 3 * 1e-3) / 1e12))
    print('    ' + '  '.join('%s %.3f/%d' % (k, ms, n) for k, (ms, n) in tm.items() if n))
# monophone-restricted K-NN
rng = np.random.RandomState(5)
p = 1.0 / np.arange(1, 46); p /= p.sum()
cls = rng.choice(45, size=N, p=p).astype(np.int32)
eng.set_unit_classes(cls)
allq = np.vstack(utts)
qc = cls[rng.randint(0, N, allq.shape[0])]
ref = None
for prec in (1, 0):
    eng.set_option('precision', prec)
    eng.knn_by_class(allq, K, qc)
    eng.reset_timers()
    before = eng.info('f16_fallbacks')
    t0 = time.time()
    for rep in range(3):
        c, d = eng.knn_by_class(allq, K, qc)
    dt = (time.time() - t0) / 3
    if ref is None: ref = (c, d)
    tm = eng.timers()
    print('by class, precision=%d: %.2f ms per %d rows  same=%s fallbacks=%d retries=%d filter %.2f ms' % (
        prec, dt * 1e3, allq.shape[0], bool(np.array_equal(c, ref[0]) and np.array_equal(d, ref[1])),
        eng.info('f16_fallbacks') - before, eng.info('last_knn_retries'), tm['knn_filter'][0] / max(tm['knn_filter'][1], 1)))
    print('    ' + '  '.join('%s %.3f/%d' % (k, ms, n) for k, (ms, n) in tm.items() if n), 'list mean/max', eng.info('last_list_mean'), eng.info('last_list_max'))
