import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snk_oracle as o, snk_oracle_c as oc, snickery_amd
N, Dt, Dj, T, K = 20000, 61, 40, 33000, 16
F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=61)
wt = np.full(Dt, 0.5); wj = np.full(Dj, 0.1)
e = snickery_amd.HipSearchEngine(0)
for k, v in [a.split('=') for a in sys.argv[1:]]:
    e.set_option(k, float(v))
e.upload_db(F_unw, JC_unw); e.set_weights(wt, wj)
F = o.weight(F_unw, wt)
rng = np.random.RandomState(62)
U = (F_unw[rng.randint(0, N, T)] + 0.3 * rng.randn(T, Dt)) * wt
cand, dist = e.knn(U, K)
oc_cand, oc_dist = oc.knn(F, U, K)
bad = np.nonzero((cand != oc_cand).any(1) | (dist != oc_dist).any(1))[0]
print('bad rows', len(bad), bad[:20], bad[-5:] if len(bad) else '')
if len(bad):
    r = bad[0]; print(cand[r], oc_cand[r]); print(dist[r], oc_dist[r])
print("fallbacks", e.info("f16_fallbacks"), "pairs", e.info("coarse_pairs"), e.info("coarse_pair_overflow"), "filter_coarse", e.info("filter_coarse"))
