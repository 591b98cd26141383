"""Viterbi side of a B* batch step alone (no K-NN beside it): the 32 utterances' lists are computed once, then
merge (one list) + join bounds + recursions are timed through snk_merge_viterbi_batch_dev."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import snickery_amd
from bench import synthetic_db, synthetic_targets

N, Dt, Dj, T, K, U = 1048576, 61, 302, 600, 100, 32
FST32 = '--fst32' in sys.argv          # viterbi_weights 1: OpenFST's float32 chain on both exact paths
_pos = [a for a in sys.argv[1:] if not a.startswith('--')]
if _pos: K = int(_pos[0])
F_unw, JC_unw = synthetic_db(N, Dt, Dj, seed=0)
wt = np.full(Dt, 0.4); wj = np.full(Dj, 0.05)
eng = snickery_amd.HipSearchEngine(0)
eng.upload_db(F_unw, JC_unw); eng.set_weights(wt, wj)
eng.set_option('viterbi_latch', 0)
if FST32: eng.set_option('viterbi_weights', 1)
utts = [synthetic_targets(F_unw, T, seed=1 + s) * wt for s in range(U)]
dev = torch.device('cuda', 0)
d2 = torch.empty(U * T, K, dtype=torch.float64, device=dev)
ids = torch.empty(U * T, K, dtype=torch.int64, device=dev)
eng.knn_local_batch_dev(utts, K, d2.data_ptr(), ids.data_ptr())
torch.cuda.synchronize()
ref = eng.knn_viterbi_batch(utts, K)
for mode, variant in ((1, 1), (1, 0), (0, 1)):
    eng.set_option('viterbi_mode', mode)
    eng.set_option('join_lb_variant', variant)
    stats0 = [eng.info(k) for k in ('dense_cells', 'dense_steps', 'dense_exact_costs', 'set_overflows')]
    paths, costs = eng.merge_viterbi_batch_dev(d2.data_ptr(), ids.data_ptr(), 1, [T] * U, K)
    eng.reset_timers()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3):
        paths, costs = eng.merge_viterbi_batch_dev(d2.data_ptr(), ids.data_ptr(), 1, [T] * U, K)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    same = all(np.array_equal(a, b) for a, b in zip(paths, ref[0])) and np.array_equal(costs, ref[1])
    stats1 = [eng.info(k) for k in ('dense_cells', 'dense_steps', 'dense_exact_costs', 'set_overflows')]
    print('viterbi_mode %d join_lb_variant %d: %.2f ms per 32-utterance batch  same=%s  refined cells/steps/exact costs/overflows per batch %s'
          % (mode, variant, dt * 1e3, same, [round((b - a) / 4) for a, b in zip(stats0, stats1)]))
    print('   ', {k: (round(v[0] / 3, 3), v[1] // 3) for k, v in eng.timers().items() if v[1]})
