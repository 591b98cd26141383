"""Scratch timing of the headline shape on the GPU box (not the bench contract)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import snickery_amd, snk_oracle as o

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
Dt, Dj = 61, 302
t0 = time.time()
F_unw, JC_unw = o.synthetic_db(N, Dt, Dj, seed=0)
print('gen %.1fs' % (time.time() - t0), flush=True)
wt = np.full(Dt, 0.7); wj = np.full(Dj, 0.1)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
utts = [o.synthetic_targets(F_unw, T, seed=s) * wt for s in range(1, 17)]
for nt in (8, 4):
    eng.set_option('db_tiles_per_wave', nt)
    eng.knn_viterbi(utts[0], K)
    eng.reset_timers()
    t0 = time.time()
    for U in utts[:4]:
        p, c = eng.knn_viterbi(U, K)
    dt = (time.time() - t0) / 4
    print('NT=%d single: %.3f ms/utt  %.0f frames/s  retries=%d listmean=%.0f listmax=%.0f' % (nt, dt * 1e3, T / dt, eng.info('last_knn_retries'), eng.info('last_list_mean'), eng.info('last_list_max')))
    for k, (ms, n) in eng.timers().items():
        if n: print('   %-18s %8.3f ms avg over %d' % (k, ms / n, n))
eng.set_option('db_tiles_per_wave', 4)
eng.knn_viterbi_batch(utts[:2], K)
eng.reset_timers()
t0 = time.time()
paths, costs = eng.knn_viterbi_batch(utts, K)
dt = (time.time() - t0) / len(utts)
print('batch: %.3f ms/utt  %.0f frames/s' % (dt * 1e3, T / dt))
for k, (ms, n) in eng.timers().items():
    if n: print('   %-18s %8.3f ms avg over %d' % (k, ms / n, n))
flops = 2.0 * T * N * Dt
ms = eng.timers()['knn_filter']; print('knn_filter TFLOP/s (f64): %.1f' % (flops / (ms[0] / ms[1] * 1e-3) / 1e12))
# correctness spot check at full size: 4 rows by numpy GEMM-form
U = utts[0]
p, c, cand, dist = eng.knn_viterbi(U, K, return_candidates=True)
F = o.weight(F_unw, wt)
for r in (0, 123, 599 if T > 599 else T - 1):
    d2 = ((F - U[r]) ** 2).sum(1)
    ref = np.lexsort((np.arange(N), d2))[:K]
    print('row', r, 'ids match:', np.array_equal(ref, cand[r]), 'maxrel', np.max(np.abs(np.sqrt(d2[ref]) - dist[r]) / dist[r]))
# greedy timing
eng.set_greedy_layout(6, False, 0) if Dj == 151 else None
