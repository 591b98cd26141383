"""Single-utterance latency (snk_knn_viterbi, one call per sentence) at the BASELINE shapes."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets

for name, N, Dt, Dj, T, K in [('B*', 1048576, 61, 302, 600, 100), ('B*short', 1048576, 61, 302, 100, 100),
                              ('B5', 1300000, 184, 151, 120, 100), ('B4', 1500000, 61, 302, 600, 200)]:
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
    U = synthetic_targets(F_unw, T, seed=1) * wt
    eng.knn_viterbi(U, K)
    eng.reset_timers()
    t0 = time.time()
    for _ in range(5): eng.knn_viterbi(U, K)
    dt = (time.time() - t0) / 5
    tm = eng.timers()
    print('%-8s N=%d Dt=%d K=%d T=%d: %.2f ms per utterance (%.0f frames/s)  %s' % (
        name, N, Dt, K, T, dt * 1e3, T / dt, ' '.join('%s %.2f' % (k.replace('knn_', ''), v[0] / 5) for k, v in tm.items() if v[1])), flush=True)
    eng.close()
