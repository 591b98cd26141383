"""Stage times and filter rates of the batch pipeline over the BASELINE shapes (B2, B4, B5, B*): a quick
look for a variant that has fallen off its usual rate."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets

SHAPES = [('B2', 700000, 61, 302, 600, 50, 16), ('B*', 1048576, 61, 302, 600, 100, 16),
          ('B4', 1500000, 61, 302, 600, 200, 8), ('B5', 1300000, 184, 151, 120, 100, 64),
          ('tw', 1000000, 123, 151, 300, 50, 16)]
for name, N, Dt, Dj, T, K, U in SHAPES:
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
    batch = snickery_amd.QueryBatch([synthetic_targets(F_unw, T, seed=1 + u) * wt for u in range(U)])
    eng.knn_viterbi_batch(batch, K)
    eng.reset_timers()
    t0 = time.time()
    for _ in range(3): eng.knn_viterbi_batch(batch, K)
    dt = (time.time() - t0) / 3
    tm = eng.timers()
    filt = tm['knn_filter'][0] / 3
    tf = 2.0 * U * T * N * Dt / (filt * 1e-3) / 1e12
    st = ' '.join('%s %.2f' % (k.replace('knn_', ''), v[0] / 3) for k, v in tm.items() if v[1])
    print('%-3s N=%d Dt=%d K=%d %dx%d: %.2f ms/step = %.0f frames/s; filter %.1f TFLOP/s (fallbacks %d, redos %d)\n     %s' % (
        name, N, Dt, K, U, T, dt * 1e3, U * T / dt, tf, eng.info('f16_fallbacks'), eng.info('batch_redos'), st), flush=True)
    eng.close()
