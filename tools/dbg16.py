import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snickery_amd, snk_oracle as o
from bench import synthetic_db, synthetic_targets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
T, K, Dt, Dj = 600, 100, 61, 302
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
print('selftest', eng.selftest_mfma())
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
print('f16_ready', eng.info('f16_ready'))
U = synthetic_targets(F_unw, T, seed=1) * wt
for nt16 in (2, 4):
    eng.set_option('f32_tiles_per_wave', nt16); eng.set_weights(wt, wj); cap = nt16
    c, d = eng.knn(U, K)
    eng.reset_timers()
    c, d = eng.knn(U, K)
    print('cap', cap, 'status', eng.info('last_f16_status'), 'fallbacks', eng.info('f16_fallbacks'), 'listmean', eng.info('last_list_mean'), 'listmax', eng.info('last_list_max'), 'pool', eng.info('pool_chunks_used'))
    for k, (ms, n) in eng.timers().items():
        if n: print('   %-18s %8.3f ms avg over %d' % (k, ms / n, n))
eng.set_option('precision', 0)
c0, d0 = eng.knn(U, K)
print('same as f64 path:', np.array_equal(c, c0), np.array_equal(d, d0))
