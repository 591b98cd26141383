"""Which first pass of the filter pays on a database that is less compact than SURVEY's generator makes it: the ball pass (its
list grows with the tiles' radii) or the coarse sweep (a fixed 1 ms per 9 600 rows + its own list).  K-NN of 9 600 rows against
the 'speechlike' variant of the B* database (bench.variant_database) for several values of coarse_gate_fraction.
    python tools/gate_time.py"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
from bench import synthetic_db, synthetic_targets, variant_database
N, Dt, Dj, K, rows = 1048576, 61, 302, 100, 9600
F0, JC0 = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
for kind in ('speechlike',):
    Fv, JCv = variant_database(kind, N, Dt, F0, JC0)
    U = np.vstack([synthetic_targets(Fv, 600, seed=1 + s) * wt for s in range(rows // 600)])
    ref = None
    for gate in (0.10, 0.20, 0.35, 0.60):
        eng = snickery_amd.HipSearchEngine(0)
        eng.set_option('coarse_gate_fraction', gate)
        eng.upload_db(Fv, JCv); eng.set_weights(wt, wj)
        for _ in range(3): cand, dist = eng.knn(U, K)            # (the voice settles on its filter)
        if ref is None: ref = (cand, dist)
        eng.reset_timers()
        for _ in range(3): eng.knn(U, K)
        st = {k: round(v[0] / 3, 3) for k, v in eng.timers().items() if v[1] and k.startswith('knn')}
        all_pairs = (rows / 32.0) * (N / 32.0)
        print('%s gate %.2f: filter_coarse %d onepass %d, pairs listed %.1f %%, same=%s, %s' % (kind, gate, eng.info('filter_coarse'), eng.info('filter_onepass'),
              100.0 * eng.info('coarse_pairs') / all_pairs, np.array_equal(ref[0], cand) and np.array_equal(ref[1], dist), st), flush=True)
        eng.close()
