"""snk_greedy_batch (up to three utterances per scan) against one call of snk_greedy per utterance."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets
for N in (65536, 1500000):
    Dt, Dj, T, me, U = 61, 151, 600, 6, 6
    F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
    wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, False, 0)
    utts = [synthetic_targets(F_unw, T - 6 * u, seed=1 + u) * wt for u in range(U)]
    for mode in (0, 1):
        eng.set_option('greedy_mode', mode)
        single = [eng.greedy(u, return_distances=True) for u in utts]
        t0 = time.time()
        for u in utts: eng.greedy(u)
        t_single = time.time() - t0
        got, gd = eng.greedy_batch(utts, return_distances=True)
        t0 = time.time()
        eng.greedy_batch(utts)
        t_batch = time.time() - t0
        ok = all(got[u] == single[u][0] and np.array_equal(gd[u], single[u][1]) for u in range(U))
        frames = sum(u.shape[0] for u in utts)
        print('mode %d ' % mode + 'N=%d: %d utterances one by one %.1f ms (%.0f frames/s), batched %.1f ms (%.0f frames/s), identical: %s' % (
            N, U, t_single * 1e3, frames / t_single, t_batch * 1e3, frames / t_batch, ok), flush=True)
    eng.close()
