"""The all-pairs join K-NN of initialise_join_table_with_knn (script/active_learning_join.py:184-212) on the doubled join rows of
an epoch voice (2 x 151 = 302 columns, script/train_halfphone.py:263-266): N start vectors against N end vectors, K = 10, in
slices of 8192 query rows through the blocked bf16-split product (knn_wide16b) + exact re-rank; one slice through the
canonical-distance selection (precision 0) for comparison.   python tools/join_knn_time.py [N]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import snickery_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
K, Dj = 10, 302
rng = np.random.RandomState(0)
JC = np.cumsum(rng.randn(N + 1, 151), axis=0); JC = (JC / JC.std()).astype(np.float32)
J2 = np.hstack([JC[:-1], JC[1:]])                        # [j_t, j_t+1]
wj = np.full(Dj, 0.05)
e = snickery_amd.HipSearchEngine(0)
e.upload_target_only(J2[1:])                             # unit_end_data
e.set_weights(wj, None)
S = J2[:-1].astype(np.float64) * wj                      # unit_start_data, weighted
e.knn(S[:8192], K)
e.reset_timers()
t0 = time.time()
nat = 0
for r0 in range(0, S.shape[0], 8192):
    i, d = e.knn(S[r0:r0 + 8192], K)
    nat += int(np.sum(i[:, 0] == np.arange(r0, r0 + i.shape[0]) - 1))
dt = time.time() - t0
st = {k: round(v[0], 1) for k, v in e.timers().items() if v[1]}
print('N = %d rows of %d columns, K = %d: all-pairs table in %.2f s (%.0f query rows/s), wide launches %d, fallbacks %d, natural successor first in %d rows; stage ms %s'
      % (N, Dj, K, dt, S.shape[0] / dt, e.info('wide_launches'), e.info('f16_fallbacks'), nat, st), flush=True)
e.set_option('precision', 0)
t0 = time.time()
i0, d0 = e.knn(S[:8192], K)
dt0 = time.time() - t0
e.set_option('precision', 1)
i1, d1 = e.knn(S[:8192], K)
print('one slice of 8192 rows through the canonical-distance selection: %.2f s (x %d slices = %.0f s for the table); same result: %s'
      % (dt0, (N + 8191) // 8192, dt0 * ((N + 8191) // 8192), bool(np.array_equal(i0, i1) and np.array_equal(d0, d1))))
