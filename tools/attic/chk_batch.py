import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 32
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
for seed0 in (1, 1000):
    utts = [synthetic_targets(F_unw, T, seed=seed0 + s) * wt for s in range(U)]
    for rep in range(2):
        t0 = time.time(); eng.knn_viterbi_batch(utts, K); dt = time.time() - t0
        print(seed0, rep, round(dt * 1e3, 2), "list", eng.info("last_list_mean"), eng.info("last_list_max"), 'redos', eng.info('batch_redos'), 'fallbacks', eng.info('f16_fallbacks'), 'pool_overflows', eng.info('pool_overflows'), 'tie', eng.info('tie_overflow'))
