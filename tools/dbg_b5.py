import sys, os
import numpy as np
sys.path.insert(0, '/root/repo')
import snickery_amd
from bench import synthetic_db, synthetic_targets
N, Dt = 1300000, 184
F_unw, JC_unw = synthetic_db(N, Dt, 151, seed=0)
wt = np.full(Dt, 0.3); wj = np.full(151, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
utts = [synthetic_targets(F_unw, 120, seed=1 + s) * wt for s in range(16)]
for u, U in enumerate(utts):
    c, d = eng.knn(U, 100)
    print(u, 'status', eng.info('last_f16_status'), 'list mean/max', eng.info('last_list_mean'), eng.info('last_list_max'))
for n in (2, 4, 8, 16):
    c, d = eng.knn(np.vstack(utts[:n]), 100)
    print('rows', 120 * n, 'status', eng.info('last_f16_status'), 'fallbacks', eng.info('f16_fallbacks'), 'list mean/max',
          eng.info('last_list_mean'), eng.info('last_list_max'), 'pool', eng.info('pool_chunks_used'))
