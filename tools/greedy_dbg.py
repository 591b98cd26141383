"""Debug helper: greedy search vs the oracle at several sizes (prints the first mismatching step)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snickery_amd
import snk_oracle as o
for N, me, T in ((300, 2, 12), (300, 6, 12), (5000, 2, 12), (5000, 3, 12), (5000, 6, 12), (20000, 1, 12), (20000, 6, 12), (40000, 1, 12), (70000, 6, 12)):
    Dt, Dj = 61, 151
    F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=3)
    wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
    F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj); eng.set_greedy_layout(me, False, 0)
    U = o.synthetic_targets(F_unw, T, seed=1) * wt
    path, d = eng.greedy(U, return_distances=True)
    pr, cr, Fwin = o.greedy_layout(F, E, S, me)
    op, od = o.greedy_search(pr, cr, Fwin, o.greedy_queries(U, me))
    ok = list(path) == list(op)
    print('N=%d me=%d: %s' % (N, me, 'OK' if ok and np.array_equal(d, od) else 'MISMATCH'), flush=True)
    if not ok:
        print(' gpu', list(path)[:10]); print(' ref', list(op)[:10]); print(' d gpu', d[:4], 'ref', od[:4])
    eng.close()
