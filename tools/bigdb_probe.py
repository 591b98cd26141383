#!/usr/bin/env python3
"""Which stage of the K-NN fast path gives up as the database grows (tools/bigdb_time.py found every group of a batch redone
through the float64 sweep at N = 12 M): one K-NN call per database size and filter variant, status word and list statistics."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from bigdb_time import walk_matrix
import snickery_amd

sizes = [int(float(a)) for a in sys.argv[1:]] or [2000000, 4000000, 8000000, 12000000]
Dt, T, K = 61, 600, 100
wt = np.full(Dt, 0.4)
for N in sizes:
    F = walk_matrix(N, Dt, 0)
    JC = np.zeros((N + 1, 4), np.float32)
    U = (F[N // 3:N // 3 + T].astype(np.float64) + 0.3 * np.random.RandomState(1).randn(T, Dt)) * wt
    eng = snickery_amd.HipSearchEngine(0)
    eng.upload_db(F, JC)
    for opts in ({}, {'prefilter_balls': 0}, {'prefilter_two_pass': 0}, {'prefilter': 0}):
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.set_weights(wt, np.full(4, 0.1))
        fb = eng.info('f16_fallbacks')
        eng.knn(U, K)
        eng.reset_timers()
        eng.knn(U, K)
        tm = eng.timers()
        print('N=%d %s: status %d, fallbacks %+d, list mean %.0f max %.0f, pool chunks %.0f, pairs %.0f (overflow %.0f), coarse %d onepass %d, filter %.3f ms, minima %.3f thr %.3f fin %.3f' % (
            N, opts, eng.info('last_f16_status'), eng.info('f16_fallbacks') - fb, eng.info('last_list_mean'), eng.info('last_list_max'), eng.info('pool_chunks_used'),
            eng.info('coarse_pairs'), eng.info('coarse_pair_overflow'), eng.info('filter_coarse'), eng.info('filter_onepass'),
            tm['knn_filter'][0] / max(tm['knn_filter'][1], 1), tm['knn_minima'][0] / max(tm['knn_minima'][1], 1), tm['knn_threshold'][0] / max(tm['knn_threshold'][1], 1),
            tm['knn_finalize'][0] / max(tm['knn_finalize'][1], 1)), flush=True)
        for k in opts:
            eng.set_option(k, 1)
    eng.close()
    del F, JC
