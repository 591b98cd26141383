import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snickery_amd, snk_oracle as o
N, Dt, Dj, T, K = 6000, 61, 151, 48, 25
F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.7); wj = np.full(Dj, 0.1)
F, E, S = o.weighted_db(F_unw, JC_unw, wt, wj)
U = o.synthetic_targets(F_unw, T, seed=1) * wt
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
cand, dist = eng.knn(U, K)
oc, od = o.knn_bruteforce(F, U, K)
print('ids equal', np.array_equal(cand, oc))
ulp = np.abs(dist - od) / np.spacing(od)
print('dist ulp diff: max', ulp.max(), 'count nonzero', (ulp > 0).sum(), 'of', ulp.size)
# squared
d2o = np.empty_like(od)
for t in range(T):
    d2o[t] = o.sqdist_rows(F[oc[t]], U[t])
print('sqrt(d2o)==od', np.array_equal(np.sqrt(d2o), od))
# does gpu dist equal sqrt of any nearby d2?
for k in range(-2, 3):
    alt = np.sqrt(d2o + k * np.spacing(d2o))
    print(k, (alt == dist).mean())
# fma variant
d2f = np.zeros_like(od)
import math
for t in range(T):
    for j in range(K):
        acc = 0.0
        for c in range(Dt):
            d = U[t, c] - F[oc[t, j], c]
            acc = math.fma(d, d, acc) if hasattr(math, 'fma') else acc + d * d
        d2f[t, j] = acc
print('fma-form match', (np.sqrt(d2f) == dist).mean(), hasattr(math, 'fma'))
J = eng.join_costs(oc)
Jo = o.join_cost_dense(E, S, oc)
m = np.isfinite(Jo)
print('join equal', np.array_equal(J, Jo), 'max ulp', (np.abs(J[m] - Jo[m]) / np.spacing(np.maximum(Jo[m], 1e-300))).max())
